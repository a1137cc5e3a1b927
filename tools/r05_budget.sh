#!/bin/bash
# GPU box: where the single-memcpy stand-ins (transport 2) should hand over to the host copy pool (transport 3): DRONE_HOST_BOUNCE_MAX_BYTES
# (default 1 MiB of unpinnable buffers = ~10 000 envs). ms per step, PCIe inclusive; (transport code) behind each figure.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; O=gpurun_out/${1:-r05_budget}; mkdir -p $O
ms() { "$@" 2>&1 | grep -v amdgpu.ids | grep -o '"transport": [0-9], "env_steps_per_s": [0-9.e+]*, "ms_per_step": [0-9.]*' | head -1 | sed 's/"transport": \([0-9]\).*"ms_per_step": \([0-9.]*\)/\2(\1)/'; }
echo "envs | zero-copy  default-budget  budget=100000(pool from ~1000 envs) | again" > $O/budget.txt
for n in 1024 2048 3072 4096 6144 8192 10240 12288; do
  line="$n"
  for rep in 1 2 3; do
    line="$line | $(ms host/drone_host --envs $n --steps 3000 --fill 0 --heap 0) $(ms host/drone_host --envs $n --steps 3000 --fill 0 --heap 1) $(DRONE_HOST_BOUNCE_MAX_BYTES=100000 ms host/drone_host --envs $n --steps 3000 --fill 0 --heap 1)"
  done
  echo "$line" >> $O/budget.txt
done
cat $O/budget.txt

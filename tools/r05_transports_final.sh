#!/bin/bash
# GPU box: the host transports' table on the round's final build (host/drone_host --steps 1500 --fill 0; ms per step, PCIe inclusive;
# (transport code) behind each figure): page-owning zero-copy | default for heap buffers | pool with 4 threads | no pool | one-thread stand-ins forced
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; O=gpurun_out/${1:-r05_transports_final}; mkdir -p $O
ms() { "$@" 2>&1 | grep -v amdgpu.ids | grep -o '"transport": [0-9], "env_steps_per_s": [0-9.e+]*, "ms_per_step": [0-9.]*' | head -1 | sed 's/"transport": \([0-9]\).*"ms_per_step": \([0-9.]*\)/\2(\1)/'; }
echo "envs | zero-copy  default  pool4  no-pool  standin1 | again" > $O/transports.txt
for n in 1024 4096 8192 16384 32768 65536 131072 262144; do
  line="$n"
  for rep in 1 2; do
    line="$line | $(ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 0) $(ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1) $(DRONE_HOST_COPY_THREADS=4 ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1) $(DRONE_HOST_COPY_THREADS=1 ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1) $(DRONE_HOST_COPY_THREADS=1 DRONE_HOST_BOUNCE_MAX_BYTES=1000000000 ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1)"
  done
  echo "$line" >> $O/transports.txt
done
cat $O/transports.txt

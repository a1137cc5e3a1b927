#!/bin/bash
# GPU box, experiment: output stores written through the L2 (sc0 sc1) + per-chunk words behind the drain alone. Parity, then timings.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; O=gpurun_out/${1:-r05_wt}; mkdir -p $O
ms() { "$@" 2>&1 | grep -v amdgpu.ids | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2; }
echo "envs | zero-copy pool(write-back) (shipped stores) | zero-copy pool(no write-back) pool(write-back) (write-through stores)" > $O/wt.txt
declare -A base
for n in 16384 32768 65536 131072; do base[$n]="$(ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 0) $(DRONE_HOST_WG_DONE_WRITEBACK=1 ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1)"; done
make -s -C drone_amd/csrc -B EXTRA=-DDRONE_EXP_OUT_WT=1 > $O/build.log 2>&1 || { cat $O/build.log; exit 1; }
timeout 600 python -m pytest tests/test_host_copy_pool_gpu.py -q -x -k "16384 or 32768" 2>&1 | tail -n 4
for n in 16384 32768 65536 131072; do
  echo "$n | ${base[$n]} | $(ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 0) $(ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1) $(DRONE_HOST_WG_DONE_WRITEBACK=1 ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1) | $(ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 0) $(ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1)" >> $O/wt.txt
done
cat $O/wt.txt
python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 | cut -c1-300

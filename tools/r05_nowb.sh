#!/bin/bash
# GPU box: what the L2 write-back ahead of the per-chunk completion words costs host transport 3 (experiment build without it, preloaded).
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; O=gpurun_out/${1:-r05_nowb}; mkdir -p $O /tmp/nowb
make -s -C drone_amd/csrc -B OUT=/tmp/nowb/libdrone_hip.so EXTRA=-DDRONE_EXP_WG_DONE_NO_WB=1 > $O/build.log 2>&1 || { cat $O/build.log; exit 1; }
ms() { "$@" 2>&1 | grep -v amdgpu.ids | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2; }
echo "envs | zero-copy  pool(shipped)  pool(no write-back) | again" > $O/nowb.txt
for n in 16384 32768 65536 131072; do
  line="$n"
  for rep in 1 2 3; do
    line="$line | $(ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 0) $(ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1) $(LD_PRELOAD=/tmp/nowb/libdrone_hip.so ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1)"
  done
  echo "$line" >> $O/nowb.txt
done
cat $O/nowb.txt

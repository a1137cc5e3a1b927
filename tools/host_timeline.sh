#!/bin/bash
# GPU box: where a step's microseconds go on the host for transport 3 (pinned stand-ins moved by the host copy pool), from a
# diagnostic build of the library (-DDRONE_HOST_STAMPS=1: CLOCK_MONOTONIC stamps at the stations of drone_vec_step, averaged and
# printed at exit) preloaded under the plain-C host. Beside it: the same host with page-owning buffers (zero-copy).
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; O=gpurun_out/${1:-host_timeline}; mkdir -p $O /tmp/hs
make -s -C drone_amd/csrc -B OUT=/tmp/hs/libdrone_hip.so EXTRA=-DDRONE_HOST_STAMPS=1 > $O/build.log 2>&1 || { cat $O/build.log; exit 1; }
for n in 16384 32768 65536 131072; do
  for rep in 1 2; do
    echo "== envs $n rep $rep" >> $O/timeline.txt
    host/drone_host --envs $n --steps 2000 --fill 0 --heap 0 2>&1 | grep -v amdgpu.ids | grep per-step | cut -c1-260 >> $O/timeline.txt
    LD_PRELOAD=/tmp/hs/libdrone_hip.so host/drone_host --envs $n --steps 2000 --fill 0 --heap 1 2>&1 | grep -v amdgpu.ids | grep -E 'per-step|host stamps' | cut -c1-400 >> $O/timeline.txt
  done
done
cat $O/timeline.txt

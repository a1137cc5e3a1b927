#!/bin/bash
# Round 5: per-chunk completion words without the L2 write-back where every output is a stand-in — parity first (pool tests, soak seeds
# with the pool transport in the draw), then the host transports' timings.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; O=gpurun_out/r05_run15; mkdir -p $O
timeout 900 python -m pytest tests/test_host_copy_pool_gpu.py tests/test_send_recv_gpu.py tests/test_robustness_gpu.py tests/test_c_host.py tests/test_multiprocess_shm_gpu.py -q -x 2>&1 | tail -n 4
for seed in 101 102; do timeout 260 python3 tests/soak_parity.py --minutes 4 --seed $seed > $O/soak_seed$seed.txt 2>&1; echo "rc=$?" >> $O/soak_seed$seed.txt; tail -n 2 $O/soak_seed$seed.txt | cut -c1-300; done
ms() { "$@" 2>&1 | grep -v amdgpu.ids | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2; }
echo "envs | zero-copy  pool  pool(DRONE_HOST_WG_DONE_WRITEBACK=1)  pool4  mirror(DRONE_HOST_COPY_THREADS=1) | again" > $O/transports.txt
for n in 4096 16384 32768 65536 131072 262144; do
  line="$n"
  for rep in 1 2 3; do
    line="$line | $(ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 0) $(ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1) $(DRONE_HOST_WG_DONE_WRITEBACK=1 ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1) $(DRONE_HOST_COPY_THREADS=4 ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1) $(DRONE_HOST_COPY_THREADS=1 ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1)"
  done
  echo "$line" >> $O/transports.txt
done
cat $O/transports.txt

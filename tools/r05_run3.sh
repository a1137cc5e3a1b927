#!/bin/bash
# Round 5, third GPU call: priority rotation A/B, the corrected pair micro-benchmark, the new host transport, every changed test.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/r05_run3"; mkdir -p "$OUT"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
# 1. priority rotation in the fused rollout: on (shipped) / off, and the period
for n in 65536 131072 262144 524288 1048576; do
  timeout 900 python3 tools/ab_step.py --mode rollout --envs $n --rounds 5 "rot8=" "off=-DDRONE_PRIO_ROTATE=0" "rot2=-DDRONE_PRIO_PERIOD_LOG2=1" "rot32=-DDRONE_PRIO_PERIOD_LOG2=5" > "$OUT/ab_prio_rollout_$n.txt" 2> "$OUT/ab_prio_rollout_$n.err"
done
for n in 131072 262144 1048576; do
  timeout 900 python3 tools/ab_step.py --mode many --k 32 --envs $n --rounds 5 "off=" "rot=-DDRONE_PRIO_ROTATE_MANY=1" > "$OUT/ab_prio_many32_$n.txt" 2> "$OUT/ab_prio_many32_$n.err"
  timeout 900 python3 tools/ab_step.py --mode many --k 8 --envs $n --rounds 5 "off=" "rot=-DDRONE_PRIO_ROTATE_MANY=1" > "$OUT/ab_prio_many8_$n.txt" 2> "$OUT/ab_prio_many8_$n.err"
done
timeout 900 python3 tools/wg_census.py --envs 131072 262144 1048576 --blocks 256 > "$OUT/wg_census_rotating.txt" 2> "$OUT/wg_census_rotating.err"
timeout 900 python3 tools/wg_census.py --envs 131072 262144 --blocks 256 --extra=-DDRONE_PRIO_ROTATE=0 > "$OUT/wg_census_oldest_first.txt" 2> "$OUT/wg_census_oldest_first.err"
for t in 1 2 3; do timeout 600 python3 tools/ab_step.py --mode rollout --task $( [ $t = 1 ] && echo waypoint || ( [ $t = 2 ] && echo swarm || echo race ) ) --envs 131072 --rounds 4 "rot8=" "off=-DDRONE_PRIO_ROTATE=0" > "$OUT/ab_prio_rollout_task${t}_131072.txt" 2>&1; done
# 2. the pair micro-benchmark, corrected
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/micro/valu_pairs.hip -o /tmp/valu_pairs 2> "$OUT/valu_pairs.build" && timeout 900 /tmp/valu_pairs > "$OUT/valu_pairs.txt" 2> "$OUT/valu_pairs.err"
# 3. host buffers the library may not pin, mid-size shards: the pool transport against the mirror and against page-owning zero-copy
for n in 4096 16384 32768 65536 131072 262144; do
  for mode in "zero-copy:--heap 0:" "pool4:--heap 1:" "pool2:--heap 1:DRONE_HOST_COPY_THREADS=2" "pool8:--heap 1:DRONE_HOST_COPY_THREADS=8" "mirror:--heap 1:DRONE_HOST_COPY_THREADS=1" "standin1:--heap 1:DRONE_HOST_BOUNCE_MAX_BYTES=1000000000"; do
    name="${mode%%:*}"; rest="${mode#*:}"; args="${rest%%:*}"; envs="${rest#*:}"
    echo "== $n $name" >> "$OUT/host_transports.txt"
    env $envs timeout 300 host/drone_host --envs $n --steps 1000 --fill 0 --rollout 8 $args 2>&1 | grep "per-step" >> "$OUT/host_transports.txt"
  done
done
# 4. tests touched this round
timeout 3000 python3 -m pytest tests/test_host_copy_pool_gpu.py tests/test_robustness_gpu.py tests/test_send_recv_gpu.py tests/test_gather_multirank_gpu.py tests/test_checkpoint_gpu.py tests/test_configs_gpu.py tests/test_parity_gpu.py tests/test_step_many_gpu.py -x -q -m gpu --durations=25 > "$OUT/pytest_a.log" 2>&1
echo "pytest a rc=$?" >> "$OUT/pytest_a.log"
timeout 1800 python3 -m pytest tests/test_bench_contract.py -x -q -m gpu --durations=10 -s > "$OUT/pytest_bench.log" 2>&1
echo "pytest bench rc=$?" >> "$OUT/pytest_bench.log"
tail -n 25 "$OUT/pytest_a.log"; tail -n 25 "$OUT/pytest_bench.log"
cat "$OUT"/ab_prio_*.txt "$OUT"/wg_census_*.txt "$OUT/host_transports.txt"
cat "$OUT/valu_pairs.txt" | cut -c1-260

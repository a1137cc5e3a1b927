#!/bin/bash
# GPU box: parity with the packed-RK4 form forced on, then A/B packed against scalar at equal placement (one library,
# the form chosen per handle: DRONE_PACKED_RK4=0/1).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; mkdir -p gpurun_out/r03_pk
DRONE_PACKED_RK4=1 python -m pytest tests/test_parity_gpu.py tests/test_step_many_gpu.py tests/test_golden_gpu.py tests/test_configs_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tee gpurun_out/r03_pk/parity_forced_on.log
DRONE_PACKED_RK4=0 python -m pytest tests/test_step_many_gpu.py tests/test_configs_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tee gpurun_out/r03_pk/parity_forced_off.log
for n in 65536 131072 262144 1048576; do
  python tools/ab_step.py --envs $n --mode many --k 32 --steps 2048 --rounds 6 "scalar=;DRONE_PACKED_RK4=0" "pk=;DRONE_PACKED_RK4=1" 2>&1 | grep variant | tee gpurun_out/r03_pk/ab_pk_many_$n.txt
  python tools/ab_step.py --envs $n --mode rollout --rounds 5 "scalar=;DRONE_PACKED_RK4=0" "pk=;DRONE_PACKED_RK4=1" 2>&1 | grep variant | tee gpurun_out/r03_pk/ab_pk_rollout_$n.txt
done
for t in waypoint race swarm; do python tools/ab_step.py --task $t --envs 65536 --mode many --k 32 --steps 2048 --rounds 5 "scalar=;DRONE_PACKED_RK4=0" "pk=;DRONE_PACKED_RK4=1" 2>&1 | grep variant | tee gpurun_out/r03_pk/ab_pk_many_${t}_65536.txt; done

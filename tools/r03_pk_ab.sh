#!/bin/bash
# GPU box: parity of the packed-RK4 build, then A/B against the scalar build at equal placement.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; mkdir -p gpurun_out/r03_pk
make -s -C drone_amd/csrc -B OUT=/tmp/libdrone_pk.so EXTRA=-DDRONE_PK_RK4=1 2>&1 | grep -E "error" 
DRONE_HIP_LIB=/tmp/libdrone_pk.so python -m pytest tests/test_parity_gpu.py tests/test_step_many_gpu.py tests/test_golden_gpu.py tests/test_configs_gpu.py -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/r03_pk/parity.log
for n in 65536 131072 1048576; do
  python tools/ab_step.py --envs $n --steps 400 scalar= "pk=-DDRONE_PK_RK4=1" 2>&1 | grep variant | tee gpurun_out/r03_pk/ab_pk_step_$n.txt
  python tools/ab_step.py --envs $n --mode many --k 32 --steps 2048 --rounds 6 scalar= "pk=-DDRONE_PK_RK4=1" 2>&1 | grep variant | tee gpurun_out/r03_pk/ab_pk_many_$n.txt
done
python tools/ab_step.py --envs 1048576 --mode rollout --rounds 5 scalar= "pk=-DDRONE_PK_RK4=1" 2>&1 | grep variant | tee gpurun_out/r03_pk/ab_pk_rollout_1048576.txt
python tools/ab_step.py --envs 65536 --mode rollout --rounds 5 scalar= "pk=-DDRONE_PK_RK4=1" 2>&1 | grep variant | tee gpurun_out/r03_pk/ab_pk_rollout_65536.txt
for t in waypoint race; do python tools/ab_step.py --task $t --envs 262144 --steps 400 scalar= "pk=-DDRONE_PK_RK4=1" 2>&1 | grep variant | tee gpurun_out/r03_pk/ab_pk_step_${t}_262144.txt; done

#!/bin/bash
# GPU box: the round-3 evidence set (copy what should be judged from gpurun_out/r03_prof/ into profiles/r03_*).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
P=r03_prof
bash tools/profile_gpu.sh $P/step_hover > /dev/null 2>&1
bash tools/profile_gpu.sh $P/step_hover_65536 --envs-per-gpu 65536 > /dev/null 2>&1
bash tools/profile_gpu.sh $P/step_hover_131072 --envs-per-gpu 131072 > /dev/null 2>&1
bash tools/profile_gpu.sh $P/step_waypoint_262144 --task waypoint --envs-per-gpu 262144 > /dev/null 2>&1
STEPS_ARGS="--steps 300 --warmup 30" bash tools/profile_gpu.sh $P/step_many_65536 --mode many --k 32 --envs-per-gpu 65536 > /dev/null 2>&1
STEPS_ARGS="--steps 300 --warmup 30" bash tools/profile_gpu.sh $P/step_many_131072 --mode many --k 32 --envs-per-gpu 131072 > /dev/null 2>&1
STEPS_ARGS="--steps 100 --warmup 10" bash tools/profile_gpu.sh $P/step_many_1048576 --mode many --k 8 > /dev/null 2>&1
STEPS_ARGS="--steps 20 --warmup 30" bash tools/profile_gpu.sh $P/rollout_hover --mode rollout > /dev/null 2>&1
bash tools/pmc_pass.sh $P/sq_rollout_hover "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" --mode rollout --steps 10 --warmup 2 > /dev/null 2>&1
bash tools/pmc_pass.sh $P/sq_step_many_65536 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU" --mode many --k 32 --steps 50 --warmup 5 --envs-per-gpu 65536 > /dev/null 2>&1
bash tools/pmc_pass.sh $P/sq2_step_many_65536 "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" --mode many --k 32 --steps 50 --warmup 5 --envs-per-gpu 65536 > /dev/null 2>&1
bash tools/pmc_pass.sh $P/sq_step_65536 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU" --steps 200 --warmup 20 --envs-per-gpu 65536 > /dev/null 2>&1
python bench.py > gpurun_out/$P/bench_default.json 2> gpurun_out/$P/bench_default.err
for d in step_hover step_hover_65536 step_hover_131072 step_waypoint_262144 step_many_65536 step_many_131072 step_many_1048576 rollout_hover; do echo "== $d"; python - "$d" <<'PY'
import json,sys
s=json.load(open(f"gpurun_out/r03_prof/{sys.argv[1]}/summary.json"))
for k,v in s["kernel_trace_avg_us"].items():
    if "step_kernel" in k or "rollout" in k or "many" in k: print(k[:70], v)
for k,v in s["traffic"].items():
    if "step_kernel" in k or "rollout" in k or "many" in k: print("traffic", v["hbm_bytes_per_launch"], v["read_bytes_corrected"], v["write_bytes"])
PY
done
for d in sq_rollout_hover sq_step_many_65536 sq2_step_many_65536 sq_step_65536; do echo "== $d"; cat gpurun_out/$P/$d/pmc_avg.json; done
tail -1 gpurun_out/$P/bench_default.json | cut -c1-600

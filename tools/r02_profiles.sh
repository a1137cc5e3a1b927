#!/bin/bash
# GPU box: the round-2 evidence set (copy what should be judged from gpurun_out/r02_prof/ into profiles/).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
bash tools/profile_gpu.sh r02_prof/step_hover > /dev/null 2>&1
bash tools/profile_gpu.sh r02_prof/step_hover_65536 --envs-per-gpu 65536 > /dev/null 2>&1
bash tools/profile_gpu.sh r02_prof/step_hover_131072 --envs-per-gpu 131072 > /dev/null 2>&1
bash tools/profile_gpu.sh r02_prof/step_hover_4194304 --envs-per-gpu 4194304 > /dev/null 2>&1
bash tools/profile_gpu.sh r02_prof/step_race --task race > /dev/null 2>&1
bash tools/profile_gpu.sh r02_prof/step_waypoint_262144 --task waypoint --envs-per-gpu 262144 > /dev/null 2>&1
bash tools/profile_gpu.sh r02_prof/rollout_hover --mode rollout --steps 20 --warmup 2 > /dev/null 2>&1
bash tools/pmc_pass.sh r02_prof/sq_step_65536 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU" --steps 200 --warmup 20 --envs-per-gpu 65536 > /dev/null 2>&1
bash tools/pmc_pass.sh r02_prof/sq_step_hover "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU" --steps 200 --warmup 20 > /dev/null 2>&1
bash tools/pmc_pass.sh r02_prof/sq_rollout_hover "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" --mode rollout --steps 10 --warmup 2 > /dev/null 2>&1
bash tools/pmc_pass.sh r02_prof/lds_step_hover "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" --steps 100 --warmup 10 > /dev/null 2>&1
SWEEP_ARGS="" bash tools/r02_sweep.sh r02_prof/sweep > /dev/null 2>&1
python bench.py > gpurun_out/r02_prof/bench_default.json 2> gpurun_out/r02_prof/bench_default.err
for d in step_hover step_hover_65536 step_hover_131072 step_hover_4194304 step_race step_waypoint_262144 rollout_hover; do echo "== $d"; python - "$d" <<'PY'
import json,sys
s=json.load(open(f"gpurun_out/r02_prof/{sys.argv[1]}/summary.json"))
for k,v in s["kernel_trace_avg_us"].items():
    if "step_kernel" in k or "rollout" in k: print(k[:60], v)
for k,v in s["traffic"].items():
    if "step_kernel" in k or "rollout" in k: print("traffic", v["hbm_bytes_per_launch"], v["read_bytes_corrected"], v["write_bytes"])
PY
done
cat gpurun_out/r02_prof/sweep/sweep.txt
tail -1 gpurun_out/r02_prof/bench_default.json | cut -c1-1500

#!/bin/bash
# GPU box: the round-4 evidence set, ALL from the one libdrone_hip.so that travelled with this snapshot (its sha256 and
# the git revision it was built from are recorded in gpurun_out/r04_prof/build.json and copied into every summary by
# tools/collect_r04.py, which turns gpurun_out/r04_prof/ into profiles/r04_*).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
P=r04_prof
mkdir -p gpurun_out/$P
python3 - <<'PY' > gpurun_out/$P/build.json
import hashlib, json, os
info = {}
try:
    info = json.load(open("drone_amd/BUILD_INFO.json"))
except (OSError, ValueError):
    pass
info["so_sha256_on_the_gpu_box"] = hashlib.sha256(open("drone_amd/libdrone_hip.so", "rb").read()).hexdigest()
print(json.dumps(info, indent=1))
PY
# the roofline kernel: per-step, beyond the Infinity Cache (what bench.py's roofline.frac describes) ...
STEPS_ARGS="--steps 400 --warmup 150" bash tools/profile_gpu.sh $P/step_hover_4194304 --envs-per-gpu 4194304 > /dev/null 2>&1
# ... and at the metric's size and the other single-GPU BASELINE sizes
bash tools/profile_gpu.sh $P/step_hover > /dev/null 2>&1
bash tools/profile_gpu.sh $P/step_hover_65536 --envs-per-gpu 65536 > /dev/null 2>&1
bash tools/profile_gpu.sh $P/step_hover_131072 --envs-per-gpu 131072 > /dev/null 2>&1
bash tools/profile_gpu.sh $P/step_waypoint_262144 --task waypoint --envs-per-gpu 262144 > /dev/null 2>&1
STEPS_ARGS="--steps 300 --warmup 30" bash tools/profile_gpu.sh $P/step_many_65536 --mode many --k 32 --envs-per-gpu 65536 > /dev/null 2>&1
STEPS_ARGS="--steps 20 --warmup 30" bash tools/profile_gpu.sh $P/rollout_hover --mode rollout > /dev/null 2>&1
bash tools/pmc_pass.sh $P/sq_rollout_hover "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" --mode rollout --steps 10 --warmup 2 > /dev/null 2>&1
bash tools/pmc_pass.sh $P/sq_step_65536 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU" --steps 200 --warmup 20 --envs-per-gpu 65536 > /dev/null 2>&1
# collect on the box too, so that the bench lines below read THIS build's traffic_latest.json / rollout_valu.json
# (the same collector runs again on the merged-back raw files at home and must produce the same profiles/)
python3 tools/collect_r04.py > gpurun_out/$P/collect_on_box.log 2>&1
python bench.py > gpurun_out/$P/bench_default.json 2> gpurun_out/$P/bench_default.err
python bench.py --force-dist --steps 200 --warmup 20 > gpurun_out/$P/bench_force_dist_one_rank.json 2> gpurun_out/$P/bench_force_dist.err
for d in step_hover_4194304 step_hover step_hover_65536 step_hover_131072 step_waypoint_262144 step_many_65536 rollout_hover; do echo "== $d"; python3 - "$d" <<'PY'
import json,sys
s=json.load(open(f"gpurun_out/r04_prof/{sys.argv[1]}/summary.json"))
for k,v in s["kernel_trace_avg_us"].items():
    if "step_kernel" in k or "rollout" in k or "many" in k: print(k[:90], v)
for k,v in s["traffic"].items():
    if "step_kernel" in k or "rollout" in k or "many" in k: print("traffic", v["hbm_bytes_per_launch"], v["read_bytes_corrected"], v["write_bytes"])
PY
done
cat gpurun_out/$P/sq_rollout_hover/pmc_avg.json | head -30
tail -1 gpurun_out/$P/bench_default.json | cut -c1-400
tail -1 gpurun_out/$P/bench_force_dist_one_rank.json | cut -c1-400

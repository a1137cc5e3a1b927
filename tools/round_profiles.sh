#!/bin/bash
# GPU box: a round's evidence set, ALL from the one libdrone_hip.so that travelled with this snapshot (its sha256 and
# the git revision it was built from are recorded in gpurun_out/<round>_prof/build.json and copied into every summary by
# tools/collect_round.py <round>, which turns gpurun_out/<round>_prof/ into profiles/<round>_*).
#   usage: tools/round_profiles.sh r06
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
ROUND="${1:?round tag, e.g. r06}"
P=${ROUND}_prof
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
# ... at the metric's size and the other single-GPU BASELINE sizes
bash tools/profile_gpu.sh $P/step_hover > /dev/null 2>&1
bash tools/profile_gpu.sh $P/step_hover_65536 --envs-per-gpu 65536 > /dev/null 2>&1
bash tools/profile_gpu.sh $P/step_hover_131072 --envs-per-gpu 131072 > /dev/null 2>&1
bash tools/profile_gpu.sh $P/step_waypoint_262144 --task waypoint --envs-per-gpu 262144 > /dev/null 2>&1
STEPS_ARGS="--steps 300 --warmup 30" bash tools/profile_gpu.sh $P/step_many_65536 --mode many --k 32 --envs-per-gpu 65536 > /dev/null 2>&1
# the fused rollout: the metric's size and the per-rank shards of configs[4] at N = 8 and N = 4
STEPS_ARGS="--steps 20 --warmup 30" bash tools/profile_gpu.sh $P/rollout_hover --mode rollout > /dev/null 2>&1
STEPS_ARGS="--steps 40 --warmup 30" bash tools/profile_gpu.sh $P/rollout_hover_131072 --mode rollout --envs-per-gpu 131072 > /dev/null 2>&1
STEPS_ARGS="--steps 40 --warmup 30" bash tools/profile_gpu.sh $P/rollout_hover_262144 --mode rollout --envs-per-gpu 262144 > /dev/null 2>&1
for n in 1048576 131072 262144; do
  tag=rollout_hover; [ $n != 1048576 ] && tag=rollout_hover_$n
  bash tools/pmc_pass.sh $P/sq_${tag}_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" --mode rollout --envs-per-gpu $n --steps 40 --warmup 30 > /dev/null 2>&1
  bash tools/pmc_pass.sh $P/sq_${tag}_b "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --mode rollout --envs-per-gpu $n --steps 40 --warmup 30 > /dev/null 2>&1
done
bash tools/pmc_pass.sh $P/sq_step_65536 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU" --steps 200 --warmup 20 --envs-per-gpu 65536 > /dev/null 2>&1
# where the rollout's waves run and how they share their SIMD; the shader clock held (diagnostic builds of the same sources)
python3 tools/wg_census.py --envs 65536 131072 262144 1048576 --blocks 256 > gpurun_out/$P/wg_census.txt 2> gpurun_out/$P/wg_census.err
python3 tools/rollout_clock.py --envs 65536 131072 262144 1048576 --valu-per-wave-step 414 > gpurun_out/$P/rollout_clock.txt 2> gpurun_out/$P/rollout_clock.err
# collect on the box too, so that the bench lines below read THIS build's traffic_latest.json / rollout_valu.json
# (the same collector runs again on the merged-back raw files at home and must produce the same profiles/)
python3 tools/collect_round.py $ROUND > gpurun_out/$P/collect_on_box.log 2>&1
python bench.py > gpurun_out/$P/bench_default.json 2> gpurun_out/$P/bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/$P/bench_driver_window.json 2> gpurun_out/$P/bench_driver_window.err
python bench.py --force-dist --steps 200 --warmup 20 > gpurun_out/$P/bench_force_dist_one_rank.json 2> gpurun_out/$P/bench_force_dist.err
for d in step_hover_4194304 step_hover step_hover_65536 step_hover_131072 step_waypoint_262144 step_many_65536 rollout_hover rollout_hover_131072 rollout_hover_262144; do echo "== $d"; python3 - "$d" "$P" <<'PY'
import json,sys
s=json.load(open(f"gpurun_out/{sys.argv[2]}/{sys.argv[1]}/summary.json"))
for k,v in s["kernel_trace_avg_us"].items():
    if "step_kernel" in k or "rollout" in k or "many" in k: print(k[:90], v)
for k,v in s["traffic"].items():
    if "step_kernel" in k or "rollout" in k or "many" in k: print("traffic", v["hbm_bytes_per_launch"], v["read_bytes_corrected"], v["write_bytes"])
PY
done
cat gpurun_out/$P/wg_census.txt gpurun_out/$P/rollout_clock.txt
tail -1 gpurun_out/$P/bench_default.json | cut -c1-600
tail -1 gpurun_out/$P/bench_driver_window.json | cut -c1-300
tail -1 gpurun_out/$P/bench_force_dist_one_rank.json | cut -c1-400

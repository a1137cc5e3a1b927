#!/bin/bash
# Round 5: the evidence set's three bench lines with the final bench.py (same library), the online-autotune check on a third box, more soak.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
P=r05_prof
mkdir -p gpurun_out/$P gpurun_out/r05_run11
python3 -c "import hashlib; print(hashlib.sha256(open('drone_amd/libdrone_hip.so','rb').read()).hexdigest())" > gpurun_out/r05_run11/so_sha256.txt
python bench.py > gpurun_out/$P/bench_default.json 2> gpurun_out/$P/bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/$P/bench_driver_window.json 2> gpurun_out/$P/bench_driver_window.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_run11/bench_driver_window_again.json 2> /dev/null
python bench.py --force-dist --steps 200 --warmup 20 > gpurun_out/$P/bench_force_dist_one_rank.json 2> gpurun_out/$P/bench_force_dist.err
bash tools/r05_autotune.sh online_box3 > gpurun_out/r05_run11/autotune.log 2>&1
for seed in 94 95 96; do timeout 400 python3 tests/soak_parity.py --minutes 6 --seed $seed > gpurun_out/r05_run11/soak_seed$seed.txt 2>&1; echo "rc=$?" >> gpurun_out/r05_run11/soak_seed$seed.txt; done
for f in gpurun_out/$P/bench_default.json gpurun_out/$P/bench_driver_window.json gpurun_out/r05_run11/bench_driver_window_again.json; do python3 - $f <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
am=d["roofline"].get("at_metric_size", d["roofline"])
print(sys.argv[1][-32:], "value", d["value"], "ms_per_step", d["ms_per_step"], "events us", am["launch_us"], "ends", d.get("episode_ends_per_env_step"), "frac", d["roofline"]["frac"])
PY
done
tail -n 2 gpurun_out/r05_run11/soak_seed9*.txt; cat gpurun_out/r05_run11/autotune.log | cut -c1-300 | head -24

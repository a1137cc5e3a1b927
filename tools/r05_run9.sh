#!/bin/bash
# Round 5: the three bench lines of the evidence set again (bench.py changed after the profile passes: longer pre-roll, the clock
# stops at the first synchronize; the LIBRARY is the one of the evidence set), an online-autotune check on this box, soak.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
P=r05_prof
mkdir -p gpurun_out/$P gpurun_out/r05_run9
python3 -c "import hashlib; print(hashlib.sha256(open('drone_amd/libdrone_hip.so','rb').read()).hexdigest())" > gpurun_out/r05_run9/so_sha256.txt
python bench.py > gpurun_out/$P/bench_default.json 2> gpurun_out/$P/bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/$P/bench_driver_window.json 2> gpurun_out/$P/bench_driver_window.err
python bench.py --force-dist --steps 200 --warmup 20 > gpurun_out/$P/bench_force_dist_one_rank.json 2> gpurun_out/$P/bench_force_dist.err
bash tools/r05_autotune.sh online_${1:-box2} > gpurun_out/r05_run9/autotune.log 2>&1
timeout 520 python3 tests/soak_parity.py --minutes 8 --seed 91 > gpurun_out/r05_run9/soak_seed91.txt 2>&1
timeout 520 python3 tests/soak_parity.py --minutes 8 --seed 92 > gpurun_out/r05_run9/soak_seed92.txt 2>&1
timeout 400 python3 tests/soak_parity.py --minutes 5 --seed 93 --big > gpurun_out/r05_run9/soak_seed93_big.txt 2>&1
for f in bench_default bench_driver_window; do python3 - $f <<'PY'
import json,sys
d=json.loads([l for l in open(f"gpurun_out/r05_prof/{sys.argv[1]}.json") if l.startswith("{")][-1])
am=d["roofline"].get("at_metric_size", d["roofline"])
print(sys.argv[1], "value", d["value"], "ms_per_step", d["ms_per_step"], "events us", am["launch_us"], "ends", d.get("episode_ends_per_env_step"), "frac", d["roofline"]["frac"], d["variants"].get("hover:4194304","")[-60:])
PY
done
tail -n 2 gpurun_out/r05_run9/soak_seed9*.txt; cat gpurun_out/r05_run9/autotune.log | cut -c1-400 | head -30; cat gpurun_out/r05_run9/so_sha256.txt

#!/bin/bash
# GPU box: (1) compute-free twin of the step kernel with dummy VALU work and occupancy caps (what separates the real
# kernel from its streaming floor?), (2) occupancy knobs on the real kernel, (3) race / waypoint / swarm at 2^20,
# (4) SQ_INSTS_VALU of the fused rollout.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O="$R/gpurun_out/r02_exp2"; mkdir -p "$O"
cd "$R"
hipcc --offload-arch=gfx950 -O3 -w tools/micro/stream_mix.hip -o /tmp/stream_mix && /tmp/stream_mix 4194304 1048576 > "$O/stream_mix_work.txt" 2>&1
grep -E "^----|fma|round-1 final" "$O/stream_mix_work.txt"
for n in 1048576 4194304; do
  python tools/ab_step.py --envs $n --rounds 5 --steps 300 "base=" "max4=-DDRONE_STEP_MAX_WAVES=4" "max3=-DDRONE_STEP_MAX_WAVES=3" "min6=-DDRONE_STEP_MIN_WAVES=6" "wg128=-DDRONE_BLOCK=128" "wg512=-DDRONE_BLOCK=512" > "$O/ab_occ_$n.txt" 2>&1
  echo "== $n"; grep variant "$O/ab_occ_$n.txt" | cut -c1-160
done
for t in race waypoint swarm; do
  python bench.py --task $t --steps 1000 --warmup 100 --cpu-seconds 0 --no-extras 2>/dev/null | tail -1 > "$O/bench_$t.json"
  python -c "import json;d=json.load(open('$O/bench_$t.json'));print('$t',d['value'],d['roofline']['launch_us'],d['roofline']['frac'])"
done
bash tools/pmc_pass.sh r02_exp2/sq_rollout "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" --mode rollout --steps 10 --warmup 2 | grep -A6 rollout

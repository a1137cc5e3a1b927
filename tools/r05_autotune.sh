#!/bin/bash
# Round 5: is the sweep order a handle measures on its own steps 161-256 the best one on THIS box? Each size:
# the autotuned handle against the three forced orders and the footprint table's pick, equal placement, and what each autotuned
# handle said it tried.   bash tools/r05_autotune.sh <tag>   -> gpurun_out/r05_autotune_<tag>/
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/r05_autotune_${1:-box}"; mkdir -p "$OUT"
cd "$R"
python3 -c "import torch; p=torch.cuda.get_device_properties(0); print('gpu uuid', getattr(p,'uuid',''), p.name)" > "$OUT/box.txt" 2>&1
for n in 2097152 4194304 8388608; do
  timeout 1200 python3 tools/ab_step.py --envs $n --rounds 4 --warm 300 --steps 150 "auto=" "o0=;DRONE_SWEEP_ORDER=0" "o6=;DRONE_SWEEP_ORDER=6" "o8=;DRONE_SWEEP_ORDER=8" "table=;DRONE_AUTOTUNE=0" > "$OUT/autotune_hover_$n.txt" 2> "$OUT/autotune_hover_$n.err"
done
timeout 1200 python3 tools/ab_step.py --task waypoint --envs 3145728 --rounds 4 --warm 300 --steps 150 "auto=" "o0=;DRONE_SWEEP_ORDER=0" "o6=;DRONE_SWEEP_ORDER=6" "o8=;DRONE_SWEEP_ORDER=8" "table=;DRONE_AUTOTUNE=0" > "$OUT/autotune_waypoint_3145728.txt" 2> "$OUT/autotune_waypoint_3145728.err"
cat "$OUT"/box.txt "$OUT"/autotune_*.txt | cut -c1-700

#!/bin/bash
# GPU box: parity of the tiled-state build, then tiled vs plane layout and line-complete on/off at equal placement.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O="$R/gpurun_out/r02_exp4"; mkdir -p "$O"
cd "$R"
timeout 1200 python -m pytest tests -x -q -m gpu > "$O/pytest.txt" 2>&1; grep -E "passed|failed|error" "$O/pytest.txt" | tail -3
for n in 4194304 1048576 131072 65536; do
  python tools/ab_step.py --envs $n --rounds 4 --steps 300 "tiled=" "planes=-DDRONE_TILED_STATE=0" "tiled_lines1=;DRONE_LINE_COMPLETE=1" "tiled_lines0=;DRONE_LINE_COMPLETE=0" "planes_lines1=-DDRONE_TILED_STATE=0;DRONE_LINE_COMPLETE=1" "tiled_wg128=-DDRONE_BLOCK=128" "tiled2=" > "$O/ab_layout_$n.txt" 2>&1
  echo "== $n"; grep variant "$O/ab_layout_$n.txt" | cut -c1-150
done
for t in waypoint race; do
python tools/ab_step.py --task $t --envs 1048576 --rounds 4 --steps 300 "tiled=" "planes=-DDRONE_TILED_STATE=0" > "$O/ab_layout_$t.txt" 2>&1; echo "== $t"; grep variant "$O/ab_layout_$t.txt" | cut -c1-150
done
python tools/ab_step.py --mode rollout --rounds 4 "tiled=" "planes=-DDRONE_TILED_STATE=0" > "$O/ab_rollout.txt" 2>&1; grep variant "$O/ab_rollout.txt" | cut -c1-150

#!/bin/bash
# GPU box: sweep order of the step kernel (DRONE_SWEEP_ORDER 0..3) across footprints, at equal placement.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O="$R/gpurun_out/r02_exp5"; mkdir -p "$O"
cd "$R"
for n in 524288 1048576 1572864 2097152 3145728 4194304 8388608; do
  python tools/ab_step.py --envs $n --rounds 3 --steps 200 "rr=;DRONE_SWEEP_ORDER=0" "xcd=;DRONE_SWEEP_ORDER=1" "rr_zz=;DRONE_SWEEP_ORDER=2" "xcd_zz=;DRONE_SWEEP_ORDER=3" "auto=" > "$O/ab_order_$n.txt" 2>&1
  echo "== $n"; grep variant "$O/ab_order_$n.txt" | cut -c1-120
done
for t in waypoint race; do
  python tools/ab_step.py --task $t --envs 4194304 --rounds 3 --steps 200 "rr=;DRONE_SWEEP_ORDER=0" "xcd=;DRONE_SWEEP_ORDER=1" "rr_zz=;DRONE_SWEEP_ORDER=2" "auto=" > "$O/ab_order_${t}_4194304.txt" 2>&1
  echo "== $t 4194304"; grep variant "$O/ab_order_${t}_4194304.txt" | cut -c1-120
done

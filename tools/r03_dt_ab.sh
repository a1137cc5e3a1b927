#!/bin/bash
# GPU box: derived-target layout on / off at equal placement, several sizes (one library; the layout is a per-handle choice)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; mkdir -p gpurun_out/r03_dt
for n in 65536 131072 262144 524288 1048576 2097152 4194304; do
  python tools/ab_step.py --envs $n --steps 300 "six_planes=;DRONE_DERIVED_TARGET=0" "derived=;DRONE_DERIVED_TARGET=1" 2>&1 | grep variant | tee gpurun_out/r03_dt/ab_dt_hover_$n.txt
done
for n in 262144 1048576 4194304; do python tools/ab_step.py --task swarm --envs $n --steps 300 "six_planes=;DRONE_DERIVED_TARGET=0" "derived=;DRONE_DERIVED_TARGET=1" 2>&1 | grep variant | tee gpurun_out/r03_dt/ab_dt_swarm_$n.txt; done

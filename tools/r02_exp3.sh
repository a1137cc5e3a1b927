#!/bin/bash
# GPU box: price the episode-end path of the step kernel at equal placement (tools/ab_step.py, one process).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O="$R/gpurun_out/r02_exp3"; mkdir -p "$O"
cd "$R"
for n in 4194304 1048576 131072; do
  python tools/ab_step.py --envs $n --rounds 4 --steps 300 "lines=" "scatter=-DDRONE_LINE_COMPLETE=0" "lines_early=-DDRONE_LOG_FOLD_LATE=0" "nolog_nopt=-DDRONE_EXP_NO_LOG=1 -DDRONE_EXP_NO_PT=1" "noends=;CFG_bound=1e6,CFG_horizon=1000000000" "lines2=" > "$O/ab_ends_$n.txt" 2>&1
  echo "== $n"; grep variant "$O/ab_ends_$n.txt" | cut -c1-170
done

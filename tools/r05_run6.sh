#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/r05_run6"; mkdir -p "$OUT"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/r05_autotune.sh box2 > "$OUT/autotune.log" 2>&1
timeout 3400 python3 -m pytest tests -x -q -m gpu --durations=15 > "$OUT/pytest_all.log" 2>&1
echo "pytest all rc=$?" >> "$OUT/pytest_all.log"
tail -n 30 "$OUT/pytest_all.log"; cat "$OUT/autotune.log" | cut -c1-900

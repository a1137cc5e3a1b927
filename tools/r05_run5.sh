#!/bin/bash
# Round 5, fifth GPU call: autotune check on this box, the whole GPU suite, a short soak with the pool transport in the draw.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/r05_run5"; mkdir -p "$OUT"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/r05_autotune.sh box1 > "$OUT/autotune.log" 2>&1
timeout 3400 python3 -m pytest tests -x -q -m gpu --durations=30 > "$OUT/pytest_all.log" 2>&1
echo "pytest all rc=$?" >> "$OUT/pytest_all.log"
timeout 400 python3 tests/soak_parity.py --minutes 4 --seed 81 > "$OUT/soak_seed81.txt" 2>&1
timeout 400 python3 tests/soak_parity.py --minutes 3 --seed 82 --big > "$OUT/soak_seed82_big.txt" 2>&1
tail -n 45 "$OUT/pytest_all.log"; tail -3 "$OUT/soak_seed81.txt" "$OUT/soak_seed82_big.txt"; cat "$OUT/autotune.log" | cut -c1-600

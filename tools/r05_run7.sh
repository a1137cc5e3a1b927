#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/r05_run7"; mkdir -p "$OUT"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/r05_autotune.sh online_box1 > "$OUT/autotune.log" 2>&1
timeout 1500 python3 -m pytest tests/test_robustness_gpu.py tests/test_configs_gpu.py tests/test_bench_contract.py tests/test_peer_store_gpu.py -x -q -m gpu --durations=12 > "$OUT/pytest_some.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest_some.log"
tail -n 25 "$OUT/pytest_some.log"; cat "$OUT/autotune.log" | cut -c1-1200

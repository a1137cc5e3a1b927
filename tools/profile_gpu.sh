#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats plus the two HBM
# PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950 —
# MI355X_MICROARCH.md "rocprofv3 PMC slots"), each in its own run, then summarise
# into gpurun_out/<tag>/summary.json. Copy what should be judged into profiles/.
#   usage: tools/profile_gpu.sh <tag> [bench.py args...]
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-prof}"; shift || true
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--cpu-seconds 0 --no-extras $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" ${STEPS_ARGS:---steps 2000 --warmup 200} $ARGS > "$OUT/stats.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" --steps 100 --warmup 10 $ARGS > "$OUT/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" --steps 100 --warmup 10 $ARGS > "$OUT/pmc_write.log" 2>&1
grep "^{" "$OUT/stats.log" | tail -1 > "$OUT/bench_line_under_rocprof.json"   # bench.py's own line while profiled
python3 "$R/tools/parse_rocprof.py" "$OUT" > "$OUT/summary.json" 2> "$OUT/parse.log"
cat "$OUT/summary.json"
# keep the merged-back payload small: drop the raw per-dispatch traces, keep stats + summaries
find "$OUT" -name '*kernel_trace.csv' -size +2M -delete
find "$OUT" -name '*counter_collection.csv' -size +2M -delete

#!/bin/bash
# Run ON THE GPU BOX: round-2 baseline evidence.
#  1. compute-free streaming floor (tools/micro/copy_bw.hip) at the shard sizes of configs[1], [2] and beyond the Infinity Cache
#  2. size sweep of the per-step kernel (rocprofv3 kernel-trace averages) 2^16 ... 2^23 envs
#  3. SQ counter pass at 65 536 envs
#   usage: tools/r02_sweep.sh <tag>
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-r02_sweep}"
OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w "$R/tools/micro/copy_bw.hip" -o /tmp/copy_bw 2>/dev/null && /tmp/copy_bw 65536 131072 262144 1048576 4194304 > "$OUT/copy_bw.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
for n in 65536 131072 262144 1048576 2097152 4194304 8388608; do
  d="$OUT/sweep_$n"; mkdir -p "$d"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$d/stats" -- python3 "$R/bench.py" --steps 1000 --warmup 100 --cpu-seconds 0 --no-extras --envs-per-gpu $n ${SWEEP_ARGS:-} > "$d/stats.log" 2>&1
  grep "^{" "$d/stats.log" | tail -1 > "$d/bench_line.json"
  python3 "$R/tools/parse_rocprof.py" "$d" > "$d/summary.json" 2>/dev/null
  find "$d" -name '*kernel_trace.csv' -delete
  python3 - "$d" $n <<'PY' >> "$OUT/sweep.txt"
import json, sys
d, n = sys.argv[1], int(sys.argv[2])
s = json.load(open(d + "/summary.json"))
for k, v in s["kernel_trace_avg_us"].items():
    if "step_kernel" in k:
        b = 278 * n
        print(json.dumps({"envs": n, "kernel": k[:40], "calls": v["calls"], "avg_us": round(v["avg_us"], 3), "min_us": v["min_us"], "max_us": v["max_us"],
                          "alg_TBps": round(b / v["avg_us"] / 1e6, 3), "frac_8TB": round(b / v["avg_us"] / 1e6 / 8.0, 4)}))
PY
done
cd "$R" && bash tools/pmc_pass.sh "$TAG/sq_65536" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU" --steps 200 --warmup 20 --envs-per-gpu 65536 > /dev/null 2>&1
cd "$R" && bash tools/pmc_pass.sh "$TAG/sq2_65536" "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" --steps 200 --warmup 20 --envs-per-gpu 65536 > /dev/null 2>&1
cat "$OUT/copy_bw.txt" "$OUT/sweep.txt"
cat "$OUT/sq_65536/pmc_avg.json" "$OUT/sq2_65536/pmc_avg.json"

#!/bin/bash
# Run ON THE GPU BOX: build tuning variants of the kernels (-D knobs of
# drone_kernels.hip), check each against the oracle (smoke) and time the
# per-step kernel with bench.py. One line per variant in gpurun_out/<tag>.txt.
#   usage: tools/exp_variants.sh <tag> "<name>|<-D flags>" ...
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="$1"; shift
mkdir -p "$R/gpurun_out"
OUT="$R/gpurun_out/$TAG.txt"
: > "$OUT"
BENCH_ARGS="${BENCH_ARGS:---steps 1000 --warmup 100 --cpu-seconds 0}"
for spec in "$@"; do
  name="${spec%%|*}"; flags="${spec#*|}"
  lib="/tmp/libdrone_$name.so"
  if ! make -s -C "$R/drone_amd/csrc" -B OUT="$lib" EXTRA="$flags" > "/tmp/build_$name.log" 2>&1; then
    echo "$name BUILD FAILED" >> "$OUT"; tail -5 "/tmp/build_$name.log" >> "$OUT"; continue
  fi
  res=$(make -s -C "$R/drone_amd/csrc" resources EXTRA="$flags" 2>&1 | grep -A6 "step_kernelILi0ELb0" | grep -E "VGPRs:|Occupancy" | sed 's/.*remark: *//' | tr '\n' ' ')
  ok=$(cd "$R" && DRONE_HIP_LIB="$lib" timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1)
  for rep in 1 2; do
    line=$(cd "$R" && DRONE_HIP_LIB="$lib" timeout 600 python3 bench.py $BENCH_ARGS 2>/dev/null | tail -1)
    echo "$name [$flags] $res | $ok | $(echo "$line" | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; f=d.get('fused_rollout') or {}
    print('launch_us=%.2f achieved=%.0fGB/s frac=%.3f value=%.3e fused=%.3e' % (r['launch_us'], r.get('achieved') or 0, r.get('frac') or 0, d['value'], f.get('env_steps_per_s',0)))
except Exception as e:
    print('PARSE FAIL', e)
")" >> "$OUT"
  done
done
cat "$OUT"

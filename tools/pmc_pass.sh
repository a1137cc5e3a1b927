#!/bin/bash
# One rocprofv3 --pmc pass over bench.py (GPU box). usage: tools/pmc_pass.sh <tag> "<CTR1 CTR2 ...>" [bench args]
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="$1"; CTRS="$2"; shift 2
OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/pmc" -- python3 "$R/bench.py" --cpu-seconds 0 --no-extras "$@" > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, json
from collections import defaultdict
root = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "drone" in row["Kernel_Name"]:
            vals[row["Kernel_Name"][:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"dispatches": len(next(iter(d.values())))} for k, d in vals.items()}
json.dump(out, open(os.path.join(root, "pmc_avg.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find "$OUT" -name '*.csv' -size +1M -delete

#!/usr/bin/env python3
"""profiles/traffic_latest.json from the summary.json files of tools/profile_gpu.sh runs:
   tools/make_traffic.py hover:1048576=profiles/r02_step_hover race:1048576=profiles/r02_step_race ..."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = os.path.join(ROOT, "profiles", "traffic_latest.json")
out = {}
for spec in sys.argv[1:]:
    key, d = spec.split("=")
    s = json.load(open(os.path.join(ROOT, d, "summary.json")))
    name = next(k for k in s["traffic"] if "step_kernel" in k)
    t = s["traffic"][name]
    line = {}
    try:
        line = json.load(open(os.path.join(ROOT, d, "bench_line_under_rocprof.json")))
    except (OSError, ValueError):
        pass
    out[key] = {
        "hbm_bytes_per_launch": t["hbm_bytes_per_launch"], "read_bytes": t["read_bytes_corrected"], "write_bytes": t["write_bytes"],
        "rocprof_kernel_avg_us": s["kernel_trace_avg_us"][name]["avg_us"],
        "bench_events_us_in_the_profiled_run": (line.get("roofline") or {}).get("launch_us"),
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, KiB units, FETCH_SIZE doubled (gfx950 wide coalesced reads count at half), averaged over dispatches",
        "source": d + "/summary.json",
    }
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1))

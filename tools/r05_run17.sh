#!/bin/bash
# Round 5: the stop word (launches queued behind a failed wait store nothing) — the peer tests, then the per-step kernel and the rollout
# against the previous build at equal placement (the kernels gained one launch-uniform test).
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; O=gpurun_out/r05_run17; mkdir -p $O
timeout 900 python -m pytest tests/test_peer_store_gpu.py tests/test_host_copy_pool_gpu.py -q -x 2>&1 | tail -n 4
for n in 65536 131072 1048576 4194304; do python3 tools/ab_step.py --envs $n --rounds 6 --steps 300 "prev=@tools/ab_libs/libdrone_hip_ca620ab.so" "stop=@drone_amd/libdrone_hip.so" > $O/ab_stop_step_$n.txt 2>&1; grep -h median_us $O/ab_stop_step_$n.txt | cut -c1-200; done
for n in 131072 1048576; do python3 tools/ab_step.py --mode rollout --envs $n --rounds 6 "prev=@tools/ab_libs/libdrone_hip_ca620ab.so" "stop=@drone_amd/libdrone_hip.so" > $O/ab_stop_rollout_$n.txt 2>&1; grep -h median_us $O/ab_stop_rollout_$n.txt | cut -c1-200; done
python bench.py --force-dist --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/bench_force_dist.json; python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_run17/bench_force_dist.json").read())
sv=d.get("secondary_values",{})
print({k:(round(v.get("ms_per_step",0)*1e3,2) if isinstance(v,dict) else v) for k,v in sv.items()})
PY

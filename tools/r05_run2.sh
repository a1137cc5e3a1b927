#!/bin/bash
# Round 5, second GPU call: the world-8 rehearsal tests, the in-kernel peer-store handshake, the pair micro-benchmark.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/r05_run2"; mkdir -p "$OUT"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
# 1. the pair micro-benchmark (what can two waves of a SIMD issue together?)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/micro/valu_pairs.hip -o /tmp/valu_pairs 2> "$OUT/valu_pairs.build" && timeout 600 /tmp/valu_pairs > "$OUT/valu_pairs.txt" 2> "$OUT/valu_pairs.err"
# 2. per-SIMD timelines of the rollout's waves
timeout 900 python3 tools/wg_census.py --envs 65536 131072 262144 --blocks 256 > "$OUT/wg_census_timeline.txt" 2> "$OUT/wg_census_timeline.err"
# 3. new tests, with durations
timeout 2400 python3 -m pytest tests/test_peer_store_gpu.py tests/test_gather_multirank_gpu.py tests/test_checkpoint_gpu.py -x -q -m gpu --durations=40 > "$OUT/pytest_exchange.log" 2>&1
echo "pytest exchange rc=$?" >> "$OUT/pytest_exchange.log"
timeout 1500 python3 -m pytest tests/test_bench_contract.py -x -q -m gpu --durations=10 -s > "$OUT/pytest_bench.log" 2>&1
echo "pytest bench rc=$?" >> "$OUT/pytest_bench.log"
# 4. does the wider kernarg (PeerSig) cost the per-step kernel anything? r04 library against this one, equal placement
for n in 65536 131072 1048576 4194304; do
  timeout 600 python3 tools/ab_step.py --envs $n --rounds 6 "r04=@tools/ab_libs/libdrone_hip_r04.so" "r05=" > "$OUT/ab_sig_step_$n.txt" 2> "$OUT/ab_sig_step_$n.err"
done
timeout 600 python3 tools/ab_step.py --mode rollout --envs 131072 --rounds 6 "r04=@tools/ab_libs/libdrone_hip_r04.so" "r05=" > "$OUT/ab_sig_rollout_131072.txt" 2> "$OUT/ab_sig_rollout_131072.err"
# 5. the handshake's cost with one rank: gather_peer_store against no_gather
timeout 900 python3 bench.py --force-dist --steps 2000 --warmup 200 --cpu-seconds 0 > "$OUT/bench_force_dist.log" 2> "$OUT/bench_force_dist.err"
DRONE_PEER_INKERNEL=0 timeout 900 python3 bench.py --force-dist --steps 2000 --warmup 200 --cpu-seconds 0 --phase optional > "$OUT/bench_force_dist_separate_launches.log" 2> "$OUT/bench_force_dist_separate_launches.err"
tail -n 30 "$OUT/pytest_exchange.log" "$OUT/pytest_bench.log"
cat "$OUT/valu_pairs.txt" "$OUT/wg_census_timeline.txt" "$OUT"/ab_sig_*.txt
python3 - "$OUT" <<'PY'
import json, sys, os
for f in ("bench_force_dist.log", "bench_force_dist_separate_launches.log"):
    try:
        d = json.loads([l for l in open(os.path.join(sys.argv[1], f)) if l.startswith("{")][-1])
        print(f, {k: round(v["ms_per_step"] * 1e3, 2) for k, v in d["records"].items() if isinstance(v, dict) and "ms_per_step" in v}, d.get("optional"))
    except Exception as e:
        print(f, "unreadable", e)
PY

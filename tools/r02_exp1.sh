#!/bin/bash
# GPU box: SPEC v4 parity, then the software-pipelined multi-chunk step kernel (DRONE_STEP_TILES) vs one chunk per
# workgroup at four shard sizes, the two-stream half-batch experiment, and the fused rollout.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O="$R/gpurun_out/r02_exp1"; mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > "$O/pytest.txt"; cat "$O/pytest.txt"
V='t1=-DDRONE_STEP_TILES=1 t2=-DDRONE_STEP_TILES=2 t4=-DDRONE_STEP_TILES=4 t8=-DDRONE_STEP_TILES=8'
for n in 1048576 4194304 131072 65536; do
  python tools/ab_step.py --envs $n --rounds 5 --steps 300 $V > "$O/ab_tiles_$n.txt" 2>&1
  echo "== $n"; grep variant "$O/ab_tiles_$n.txt" | cut -c1-200
done
python tools/ab_step.py --mode rollout --rounds 5 "base=" > "$O/ab_rollout.txt" 2>&1; grep variant "$O/ab_rollout.txt" | cut -c1-200
python tools/two_stream.py --envs 65536 131072 262144 --halves 1 2 4 > "$O/two_stream.txt" 2>&1; cat "$O/two_stream.txt"

#!/bin/bash
# Round 5, fourth GPU call: the RK4 constants in vector registers (A/B), operand-position variants of the pair micro-benchmark,
# the whole GPU suite with durations.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/r05_run4"; mkdir -p "$OUT"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/micro/valu_pairs.hip -o /tmp/valu_pairs 2> "$OUT/valu_pairs.build" && timeout 900 /tmp/valu_pairs > "$OUT/valu_pairs.txt" 2> "$OUT/valu_pairs.err"
for n in 65536 131072 262144 524288 1048576; do
  timeout 900 python3 tools/ab_step.py --mode rollout --envs $n --rounds 5 "vgpr=" "sgpr=-DDRONE_RK4_VGPR_CONSTS=0" > "$OUT/ab_vconst_rollout_hover_$n.txt" 2> "$OUT/ab_vconst_rollout_hover_$n.err"
done
for task in waypoint swarm race; do for n in 262144 1048576; do
  timeout 900 python3 tools/ab_step.py --mode rollout --task $task --envs $n --rounds 4 "vgpr=" "sgpr=-DDRONE_RK4_VGPR_CONSTS=0" > "$OUT/ab_vconst_rollout_${task}_$n.txt" 2> "$OUT/ab_vconst_rollout_${task}_$n.err"
done; done
for n in 131072 262144 1048576; do
  timeout 900 python3 tools/ab_step.py --mode many --k 32 --envs $n --rounds 5 "vgpr=" "sgpr=-DDRONE_RK4_VGPR_CONSTS=0" > "$OUT/ab_vconst_many32_hover_$n.txt" 2> "$OUT/ab_vconst_many32_hover_$n.err"
done
timeout 900 python3 tools/ab_step.py --mode many --k 32 --task waypoint --envs 262144 --rounds 5 "vgpr=" "sgpr=-DDRONE_RK4_VGPR_CONSTS=0" > "$OUT/ab_vconst_many32_waypoint_262144.txt" 2> "$OUT/ab_vconst_many32_waypoint_262144.err"
timeout 900 python3 tools/wg_census.py --envs 131072 262144 1048576 --blocks 256 > "$OUT/wg_census_vconst.txt" 2> "$OUT/wg_census_vconst.err"
for n in 131072 1048576; do
  bash tools/pmc_pass.sh r05_run4/sq_a_$n "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" --mode rollout --envs-per-gpu $n --steps 40 --warmup 30 > /dev/null
  bash tools/pmc_pass.sh r05_run4/sq_b_$n "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --mode rollout --envs-per-gpu $n --steps 40 --warmup 30 > /dev/null
done
# the whole GPU suite, as the driver runs it, with durations
timeout 3400 python3 -m pytest tests -x -q -m gpu --durations=40 > "$OUT/pytest_all.log" 2>&1
echo "pytest all rc=$?" >> "$OUT/pytest_all.log"
python3 -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" >> "$OUT/smoke.log"
tail -n 60 "$OUT/pytest_all.log"; tail -3 "$OUT/smoke.log"
cat "$OUT"/ab_vconst_*.txt "$OUT/wg_census_vconst.txt"
cat "$OUT"/sq_*/pmc_avg.json | grep -A12 rollout | head -80
grep -E "^solo|^same" "$OUT/valu_pairs.txt" | cut -c1-240 | tail -50

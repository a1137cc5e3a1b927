#!/bin/bash
# Round 5, VERDICT r4 item 2: what limits the fused rollout at the per-rank shard sizes (131 072 / 262 144 envs)?
# (GPU box) SQ counter passes at four sizes, the shader clock held, where the workgroups land, workgroup-size A/B.
#   bash tools/r05_rollout_counters.sh   -> gpurun_out/r05_rollout/
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/r05_rollout"; mkdir -p "$OUT"
cd "$R"
python3 -c "import json,hashlib,subprocess; print(json.dumps({'lib_sha256': hashlib.sha256(open('drone_amd/libdrone_hip.so','rb').read()).hexdigest()}))" > "$OUT/build.json"
for n in 65536 131072 262144 1048576; do
  bash tools/pmc_pass.sh r05_rollout/sq_a_$n "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" --mode rollout --envs-per-gpu $n --steps 40 --warmup 30 > /dev/null
  bash tools/pmc_pass.sh r05_rollout/sq_b_$n "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VALU" --mode rollout --envs-per-gpu $n --steps 40 --warmup 30 > /dev/null
  bash tools/pmc_pass.sh r05_rollout/sq_c_$n "SQ_IFETCH SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU" --mode rollout --envs-per-gpu $n --steps 40 --warmup 30 > /dev/null
done
python3 tools/wg_census.py --envs 65536 131072 262144 524288 1048576 --blocks 256 512 128 > "$OUT/wg_census.txt" 2> "$OUT/wg_census.err"
python3 tools/rollout_clock.py --envs 65536 131072 262144 1048576 --valu-per-wave-step 412.5 > "$OUT/rollout_clock.txt" 2> "$OUT/rollout_clock.err"
for n in 65536 131072 262144 524288 1048576; do
  python3 tools/ab_step.py --mode rollout --envs $n --rounds 5 "b256=" "b512=-DDRONE_BLOCK=512" "b128=-DDRONE_BLOCK=128" "b64=-DDRONE_BLOCK=64" > "$OUT/ab_block_rollout_$n.txt" 2> "$OUT/ab_block_rollout_$n.err"
done
for n in 65536 131072 262144; do
  python3 tools/ab_step.py --mode many --k 32 --envs $n --rounds 5 "b256=" "b512=-DDRONE_BLOCK=512" "b128=-DDRONE_BLOCK=128" > "$OUT/ab_block_many_$n.txt" 2> "$OUT/ab_block_many_$n.err"
done
ls -la "$OUT"
tail -n +1 "$OUT"/wg_census.txt "$OUT"/rollout_clock.txt "$OUT"/ab_block_rollout_*.txt "$OUT"/ab_block_many_*.txt

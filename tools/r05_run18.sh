#!/bin/bash
# Round 5: the stop word with the rollout kernel's check behind its loop — peer tests, rollout / per-step A/B against the previous build.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; O=gpurun_out/r05_run18; mkdir -p $O
timeout 900 python -m pytest tests/test_peer_store_gpu.py -q -x 2>&1 | tail -n 3
for n in 65536 131072 262144 1048576; do python3 tools/ab_step.py --mode rollout --envs $n --rounds 6 "prev=@tools/ab_libs/libdrone_hip_ca620ab.so" "stop=@drone_amd/libdrone_hip.so" > $O/ab_stop_rollout_$n.txt 2>&1; grep -h median_us $O/ab_stop_rollout_$n.txt | cut -c1-200; done
for t in waypoint swarm race; do python3 tools/ab_step.py --mode rollout --task $t --envs 262144 --rounds 4 "prev=@tools/ab_libs/libdrone_hip_ca620ab.so" "stop=@drone_amd/libdrone_hip.so" > $O/ab_stop_rollout_${t}_262144.txt 2>&1; grep -h median_us $O/ab_stop_rollout_${t}_262144.txt | cut -c1-200; done
for n in 131072 1048576; do python3 tools/ab_step.py --mode many --k 8 --envs $n --rounds 6 "prev=@tools/ab_libs/libdrone_hip_ca620ab.so" "stop=@drone_amd/libdrone_hip.so" > $O/ab_stop_many8_$n.txt 2>&1; grep -h median_us $O/ab_stop_many8_$n.txt | cut -c1-200; done

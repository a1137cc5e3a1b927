#!/bin/bash
# Round 5, after the registration rule changed (host buffers are pinned only on the caller's word): the whole GPU suite, then — only
# if it is green — the evidence set from this build, then the soak seeds that died under the old rule and fresh ones.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; O=gpurun_out/r05_run13; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?
tail -n 5 $O/pytest_gpu.txt
if [ $rc = 0 ]; then bash tools/r05_profiles.sh > $O/profiles.log 2>&1; tail -n 40 $O/profiles.log | cut -c1-400; fi
for seed in 92 94 97 98; do timeout 330 python3 tests/soak_parity.py --minutes 5 --seed $seed > $O/soak_seed$seed.txt 2>&1; echo "rc=$?" >> $O/soak_seed$seed.txt; tail -n 2 $O/soak_seed$seed.txt | cut -c1-300; done

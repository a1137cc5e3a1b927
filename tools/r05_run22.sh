#!/bin/bash
# Round 5 final: the whole GPU suite, the evidence set from this build, two soak seeds.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; O=gpurun_out/r05_run22; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -n 6 $O/pytest_gpu.txt | cut -c1-300
bash tools/r05_profiles.sh > $O/profiles.log 2>&1; tail -n 4 $O/profiles.log | cut -c1-600
for seed in 106; do timeout 330 python3 tests/soak_parity.py --minutes 5 --seed $seed > $O/soak_seed$seed.txt 2>&1; echo "rc=$?" >> $O/soak_seed$seed.txt; tail -n 2 $O/soak_seed$seed.txt | cut -c1-300; done

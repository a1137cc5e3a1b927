#!/bin/bash
# Round 5: the whole GPU suite, the evidence set from this build, and the host-side timeline of transport 3.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; O=gpurun_out/r05_run14; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -n 8 $O/pytest_gpu.txt | cut -c1-300
bash tools/r05_profiles.sh > $O/profiles.log 2>&1; tail -n 45 $O/profiles.log | cut -c1-400
bash tools/host_timeline.sh r05_run14 2>&1 | cut -c1-400

#!/bin/bash
# GPU box: host transport 3 with and without the per-chunk completion words (DRONE_HOST_CHUNK_WORDS_MIN_WG=100000: one wait for the
# kernel, then the whole copy-out by the pool's threads). ms per step, PCIe inclusive.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; O=gpurun_out/${1:-r05_words}; mkdir -p $O
ms() { "$@" 2>&1 | grep -v amdgpu.ids | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2; }
echo "envs | zero-copy  pool+words  pool, whole copy after the wait | again" > $O/words.txt
for n in 8192 16384 32768 65536 131072 262144; do
  line="$n"
  for rep in 1 2 3; do
    line="$line | $(ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 0) $(ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1) $(DRONE_HOST_CHUNK_WORDS_MIN_WG=100000 ms host/drone_host --envs $n --steps 1500 --fill 0 --heap 1)"
  done
  echo "$line" >> $O/words.txt
done
cat $O/words.txt

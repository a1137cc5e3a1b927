#!/bin/bash
# (Needs profiles/r05_ab/store_window_not_kept.patch applied: the knob is not in the shipped library.)
# GPU box: host transport 3 with the kernel's stores released in launch order, W workgroups at a time (LaunchSig::store_window;
# DRONE_HOST_STORE_WINDOW=W, 0 = all workgroups at once as before), against page-owning zero-copy buffers. ms per step, PCIe inclusive.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; O=gpurun_out/${1:-r05_window}; mkdir -p $O
timeout 600 python -m pytest tests/test_host_copy_pool_gpu.py tests/test_send_recv_gpu.py -q -x 2>&1 | tail -n 3
us() { "$@" 2>&1 | grep -v amdgpu.ids | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2; }
echo "envs zero-copy W=0 W=1 W=2 W=4 W=8 W=16 W=32" > $O/window.txt
for n in 16384 32768 65536 131072 262144; do
  line="$n"
  for rep in 1 2; do
    line="$line | $(us host/drone_host --envs $n --steps 1500 --fill 0 --heap 0)"
    for w in 0 1 2 4 8 16 32; do line="$line $(DRONE_HOST_STORE_WINDOW=$w us host/drone_host --envs $n --steps 1500 --fill 0 --heap 1)"; done
  done
  echo "$line" >> $O/window.txt
done
cat $O/window.txt
bash tools/host_timeline.sh ${1:-r05_window} 2>&1 | cut -c1-400

#!/usr/bin/env python3
"""Where do a small shard's microseconds go? Builds the DIAGNOSTIC variant of the library (-DDRONE_STAMPS=1: s_memtime
at the phase boundaries of the step kernel, one row per wave), runs steps at the given sizes and prints the median
per-phase durations over the waves of the last launch, plus the spread of wave start / end times (s_memrealtime,
100 MHz) — i.e. how long the launch takes to get all its waves going and how ragged the tail is. Read the SHARES, not
the absolute length: the stamps' fences forbid overlaps the real kernel has.
   python tools/stamps.py --envs 65536 131072 1048576"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = ["entry->loads issued", "loads issued->data arrived", "integrate+reward+reset (VALU)", "state stores issued", "observation math",
          "LDS transpose+barrier+obs stores", "log fold"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[65536, 131072, 1048576])
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    import torch

    from drone_amd import abi, binding

    lib = "/tmp/libdrone_stamps.so"
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "drone_amd", "csrc"), "-B", f"OUT={lib}", "EXTRA=-DDRONE_STAMPS=1"], check=True, capture_output=True)
    fns = binding.load_variant(lib)
    raw = C.CDLL(lib)
    raw.drone_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    for n in a.envs:
        v = binding.DroneVec(n, seed=0, task=abi.TASK_HOVER, device="cuda:0", fns=fns)
        v.reset(0)
        ring = [torch.empty_like(v.actions) for _ in range(4)]
        for g, r in enumerate(ring):
            v.fill_random_actions(gstep=g, out=r)
        for k in range(a.steps):
            v.bind_actions(ring[k & 3]); v.step()
        torch.cuda.synchronize()
        rows = (n + 63) // 64
        buf = np.zeros((rows, 10), dtype=np.uint64)
        got = raw.drone_debug_stamps(v._h, buf.ctypes.data, rows)
        assert got == rows
        t = buf[:, :8].astype(np.int64)
        d = np.diff(t, axis=1)
        rt0, rt1 = buf[:, 8].astype(np.int64), buf[:, 9].astype(np.int64)
        clock_mhz = float(np.median((t[:, 7] - t[:, 0]) / np.maximum(rt1 - rt0, 1))) * 100.0
        out = {"envs": n, "waves": rows, "shader_clock_MHz_est": round(clock_mhz),
               "phase_cycles_median": {p: int(np.median(d[:, k])) for k, p in enumerate(PHASES)},
               "phase_us_median": {p: round(float(np.median(d[:, k])) / clock_mhz, 3) for k, p in enumerate(PHASES)},
               "wave_lifetime_us_median": round(float(np.median(t[:, 7] - t[:, 0])) / clock_mhz, 3),
               "first_wave_start_to_last_wave_start_us": round((rt0.max() - rt0.min()) / 100.0, 2),
               "first_wave_start_to_last_wave_end_us": round((rt1.max() - rt0.min()) / 100.0, 2),
               "wave_end_spread_us(p5,p50,p95 after first start)": [round(float(x - rt0.min()) / 100.0, 2) for x in np.percentile(rt1, [5, 50, 95])]}
        print(json.dumps(out))
        v.close()


if __name__ == "__main__":
    main()

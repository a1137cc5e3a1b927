#!/usr/bin/env python3
"""What shader clock does the chip hold while the fused rollout runs? Diagnostic build (-DDRONE_STAMPS=1): every wave
stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) before and after its horizon loop; clock = delta cycles /
delta realtime x 100 MHz (MI355X_MICROARCH.md "DVFS give-back" item 6), median over the waves of a launch taken after
seconds of back-to-back launches. With the instruction count per wave-step (SQ_INSTS_VALU) this turns "fraction of the
157.3 TF datasheet peak" (which assumes 2.4 GHz) into cycles per VALU instruction per SIMD at the clock actually held.
   python tools/rollout_clock.py [--envs 1048576 65536] [--valu-per-wave-step 426]"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[1 << 20, 65536])
    ap.add_argument("--horizon", type=int, default=128)
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--valu-per-wave-step", type=float, default=426.0)
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
        t0 = time.time()
        launches = 0
        while time.time() - t0 < a.seconds:  # back to back: the clock settles ~30 ms after the GPU leaves idle
            for _ in range(20):
                v.rollout(a.horizon)
            torch.cuda.synchronize()
            launches += 20
        v.timer_start()
        for _ in range(10):
            v.rollout(a.horizon)
        ms = v.timer_stop() / 10
        rows = (n + 63) // 64
        buf = np.zeros((rows, 10), dtype=np.uint64)
        assert raw.drone_debug_stamps(v._h, buf.ctypes.data, rows) == rows
        cyc = (buf[:, 1] - buf[:, 0]).astype(np.float64)
        rt = (buf[:, 9] - buf[:, 8]).astype(np.float64)
        ghz = np.median(cyc / np.maximum(rt, 1.0)) * 0.1
        waves_per_simd = rows / 1024.0
        cyc_per_wave_step = float(np.median(cyc)) / a.horizon
        # one SIMD time-shares its resident waves: cycles per VALU per SIMD = wave cycles per step / instructions per step / waves sharing the SIMD
        resident = min(waves_per_simd, 4.0)
        out = {"envs": n, "ms_per_launch": round(ms, 4), "shader_clock_GHz_held": round(float(ghz), 3), "fraction_of_2.4_GHz": round(float(ghz) / 2.4, 3),
               "wave_cycles_per_step_median": round(cyc_per_wave_step), "waves_per_simd_total": waves_per_simd, "resident_waves_per_simd": resident,
               "cycles_per_valu_per_simd": round(cyc_per_wave_step / a.valu_per_wave_step / resident, 3),
               "f32_vector_peak_at_held_clock_TF": round(157.3 * float(ghz) / 2.4, 1), "launches_before": launches}
        print(json.dumps(out), flush=True)
        v.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Where do the fused rollout's workgroups land? (round 5)

A shard of 131 072 envs is 512 workgroups of four waves on 256 CUs: two per CU IF the dispatcher deals them evenly.
Nothing promises that (MI355X_MICROARCH.md, "HIP promises nothing about ... workgroup->XCD placement"): every
workgroup of such a launch fits at once (the kernel's registers allow four waves per SIMD), so a CU may take three or
four while its neighbour takes one, and the launch ends with its most loaded CU. Diagnostic build (-DDRONE_STAMPS=1):
every wave of the rollout kernel records HW_REG_HW_ID / HW_REG_XCC_ID and its own cycle / real-time stamps; this tool
prints, per size and workgroup size, the histogram of waves per SIMD and workgroups per CU, and the median wave lifetime
by how many waves shared the wave's SIMD.
   python tools/wg_census.py --envs 65536 131072 262144 [--blocks 256 512]"""
import argparse
import collections
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hw_fields(hw, xcc):
    """gfx9 HW_ID layout: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]; XCC_ID: xcc[3:0]."""
    return {"wave": hw & 0xF, "simd": (hw >> 4) & 3, "cu": (hw >> 8) & 0xF, "sh": (hw >> 12) & 1, "se": (hw >> 13) & 7, "xcc": xcc & 0xF}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[65536, 131072, 262144])
    ap.add_argument("--blocks", type=int, nargs="+", default=[256])
    ap.add_argument("--horizon", type=int, default=128)
    ap.add_argument("--launches", type=int, default=40)
    ap.add_argument("--extra", default="", help="more -D flags for the diagnostic build")
    a = ap.parse_args()
    import torch

    from drone_amd import abi, binding

    for block in a.blocks:
        lib = f"/tmp/libdrone_census_{block}_{abs(hash(a.extra)) % 10**6}.so"
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "drone_amd", "csrc"), "-B", f"OUT={lib}", f"EXTRA=-DDRONE_STAMPS=1 -DDRONE_BLOCK={block} {a.extra}"],
                       check=True, capture_output=True)
        fns = binding.load_variant(lib)
        raw = C.CDLL(lib)
        raw.drone_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        for n in a.envs:
            v = binding.DroneVec(n, seed=0, task=abi.TASK_HOVER, device="cuda:0", fns=fns)
            v.reset(0)
            for _ in range(a.launches):
                v.rollout(a.horizon)
            torch.cuda.synchronize()
            v.timer_start()
            for _ in range(10):
                v.rollout(a.horizon)
            ms = v.timer_stop() / 10
            rows = (n + 63) // 64
            buf = np.zeros((rows, 10), dtype=np.uint64)
            assert raw.drone_debug_stamps(v._h, buf.ctypes.data, rows) == rows
            hw, xcc = buf[:, 2].astype(np.int64), buf[:, 3].astype(np.int64)
            cyc = (buf[:, 1] - buf[:, 0]).astype(np.float64)
            rt0, rt1 = buf[:, 8].astype(np.int64), buf[:, 9].astype(np.int64)
            simd_of, cu_of = [], []
            for h, x in zip(hw, xcc):
                f = hw_fields(int(h), int(x))
                cu = (f["xcc"], f["se"], f["sh"], f["cu"])
                cu_of.append(cu)
                simd_of.append(cu + (f["simd"],))
            per_simd = collections.Counter(simd_of)
            per_cu = collections.Counter(cu_of)
            waves_per_wg = block // 64
            share = np.array([per_simd[s] for s in simd_of])
            life_by_share = {int(k): round(float(np.median(cyc[share == k])) / a.horizon) for k in sorted(set(share))}
            # per SIMD, its waves in start order: median start / end (us after the launch's first wave) and cycles per step of the k-th wave —
            # shows whether the waves of a SIMD share it evenly or the oldest runs ahead (VALU issue is arbitrated by age)
            by_simd = collections.defaultdict(list)
            for w, s in enumerate(simd_of):
                by_simd[s].append((int(rt0[w]), int(rt1[w]), float(cyc[w])))
            common = collections.Counter(len(v_) for v_ in by_simd.values()).most_common(1)[0][0]
            lanes = [sorted(v_) for v_ in by_simd.values() if len(v_) == common and common <= 4]
            timeline = [{"start_us": round(float(np.median([l[k][0] for l in lanes]) - rt0.min()) / 100.0, 1),
                         "end_us": round(float(np.median([l[k][1] for l in lanes]) - rt0.min()) / 100.0, 1),
                         "cycles_per_step": round(float(np.median([l[k][2] for l in lanes])) / a.horizon)} for k in range(common)] if lanes else None
            # the hardware wave slots (HW_ID[3:0]) of the waves that shared a SIMD: the kernel's priority rotation starts each wave from its slot
            slots_by_simd = collections.defaultdict(list)
            for h, s_ in zip(hw, simd_of):
                slots_by_simd[s_].append(int(h) & 0xF)
            slot_sets = collections.Counter(tuple(sorted(v_)) for v_ in slots_by_simd.values()).most_common(6)
            out = {"envs": n, "block": block, "waves": rows, "ms_per_launch": round(ms, 4), "waves_of_a_simd_in_start_order": timeline,
                   "wave_slots_on_a_simd_most_common": [[list(k), c] for k, c in slot_sets],
                   "cus_used": len(per_cu), "simds_used": len(per_simd),
                   "workgroups_per_cu_histogram": dict(sorted(collections.Counter(c // waves_per_wg for c in per_cu.values()).items())),
                   "waves_per_simd_histogram": dict(sorted(collections.Counter(per_simd.values()).items())),
                   "wave_cycles_per_step_median_by_waves_on_its_simd": life_by_share,
                   "first_start_to_last_end_us": round((rt1.max() - rt0.min()) / 100.0, 1),
                   "wave_end_after_first_start_us_p5_p50_p95_max": [round(float(x - rt0.min()) / 100.0, 1) for x in list(np.percentile(rt1, [5, 50, 95])) + [rt1.max()]],
                   "note": "waves that overlapped in time only if the launch fits at once (<= 4 waves per SIMD at 108 VGPRs); the last launch's placement"}
            print(json.dumps(out), flush=True)
            v.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A/B the per-step kernel across builds of the library in ONE process with
interleaved rounds (cdna_hip_programming.md §5.4 rule 24): separate processes
differ by more than the variants do. Run on the GPU box.

  python tools/ab_step.py [--envs N] [--task hover] [--rounds 12] [--steps 200] name=-Dflags ...
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1 << 20)
    ap.add_argument("--task", default="hover")
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--mode", default="step", choices=["step", "rollout"])
    ap.add_argument("--horizon", type=int, default=128)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    import torch

    from drone_amd import abi, binding

    task = abi.TASK_HOVER if a.task == "hover" else abi.TASK_WAYPOINT
    vecs = {}
    for spec in a.variants:
        name, _, flags = spec.partition("=")
        lib = f"/tmp/ab_{name}.so"
        r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "drone_amd", "csrc"), "-B", f"OUT={lib}", f"EXTRA={flags}"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            print(name, "BUILD FAILED", r.stderr[-400:])
            continue
        fns = binding.load_variant(lib)
        v = binding.DroneVec(a.envs, seed=0, task=task, device="cuda:0", fns=fns)
        v.reset(0)
        ring = [torch.empty_like(v.actions) for _ in range(4)]
        for k, r_ in enumerate(ring):
            v.fill_random_actions(gstep=k, out=r_)
        vecs[name] = (v, ring, flags)
    times = {k: [] for k in vecs}
    for rnd in range(a.rounds + 1):
        for name, (v, ring, _) in vecs.items():
            torch.cuda.synchronize()
            v.timer_start()
            if a.mode == "step":
                for k in range(a.steps):
                    v.bind_actions(ring[k % len(ring)])
                    v.step()
                per = v.timer_stop() * 1e3 / a.steps
            else:
                v.rollout(a.horizon)
                per = v.timer_stop() * 1e3
            if rnd > 0:
                times[name].append(per)
    base = None
    for name, ts in times.items():
        med, mn = statistics.median(ts), min(ts)
        base = base or med
        print(json.dumps({"variant": name, "flags": vecs[name][2], "median_us": round(med, 2), "min_us": round(mn, 2),
                          "max_us": round(max(ts), 2), "vs_first": round(med / base, 4)}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Small shards: one handle on one stream vs. the same envs split into H half-batches, each with its own handle
and HIP stream, launched round-robin (what an async double-buffered vec-env does: while the policy consumes half
A, half B steps). Launches on different streams carry no dependency, so the ~2.5 us dependent-launch boundary of
one half overlaps the execution of the other.
   python tools/two_stream.py --envs 65536 131072 --halves 1 2 4"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[65536, 131072])
    ap.add_argument("--halves", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    import torch

    from drone_amd import abi, binding

    dev = torch.device("cuda:0")
    for n in a.envs:
        for h in a.halves:
            part = n // h
            streams = [torch.cuda.Stream(device=dev) for _ in range(h)]
            vecs, rings = [], []
            for k, s in enumerate(streams):
                with torch.cuda.stream(s):
                    v = binding.DroneVec(part, seed=0, task=abi.TASK_HOVER, device=dev, env_offset=k * part)
                    v.reset(0)
                    ring = [torch.empty_like(v.actions) for _ in range(4)]
                    for g, r in enumerate(ring):
                        v.fill_random_actions(gstep=g, out=r)
                vecs.append(v)
                rings.append(ring)
            torch.cuda.synchronize()
            best = None
            for _ in range(a.rounds):
                for t in range(50):
                    for v, ring in zip(vecs, rings):
                        v.bind_actions(ring[t & 3]); v.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for t in range(a.steps):
                    for v, ring in zip(vecs, rings):
                        v.bind_actions(ring[t & 3]); v.step()
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                best = el if best is None else min(best, el)
            us = best * 1e6 / a.steps
            print(json.dumps({"envs": n, "handles_x_streams": h, "envs_per_handle": part, "us_per_full_step": round(us, 3),
                              "env_steps_per_s": round(n * a.steps / best, 1), "alg_TBps": round(278 * n / us / 1e6, 3), "frac_8TB": round(278 * n / us / 1e6 / 8, 4)}))
            for v in vecs:
                v.close()


if __name__ == "__main__":
    main()

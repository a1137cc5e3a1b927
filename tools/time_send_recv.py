#!/usr/bin/env python3
"""What drone_vec_step_send / step_recv buy a host-buffer consumer: N envs as ONE handle stepped synchronously, against
the same N envs as TWO handles of N/2 stepping out of phase (each sent before the other is received), per env step.

    python tools/time_send_recv.py [--envs 1024 4096 16384 65536] [--steps 2000]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from drone_amd import binding  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[1024, 4096, 16384, 65536])
    ap.add_argument("--steps", type=int, default=2000)
    a = ap.parse_args()
    for n in a.envs:
        one = binding.DroneVec(n, seed=0)
        one.reset(0)
        one.fill_random_actions()
        for _ in range(50):
            one.step()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            one.step()
        sync_us = (time.perf_counter() - t0) / a.steps * 1e6
        one.close()
        h = [binding.DroneVec(n // 2, seed=0, env_offset=k * (n // 2)) for k in range(2)]
        for x in h:
            x.reset(0)
            x.fill_random_actions()
        h[0].step_send()
        for _ in range(50):
            h[1].step_send(); h[0].step_recv(); h[0].step_send(); h[1].step_recv()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            h[1].step_send()
            h[0].step_recv()
            h[0].step_send()
            h[1].step_recv()
        pair_us = (time.perf_counter() - t0) / a.steps * 1e6
        h[0].step_recv()
        for x in h:
            x.close()
        print(f"{n} envs, host buffers ({binding.DroneVec.__name__}): one handle, step() {sync_us:.1f} us per step = {n / sync_us * 1e6:.3e} env-steps/s; "
              f"two handles of {n // 2} out of phase {pair_us:.1f} us per step of all {n} = {n / pair_us * 1e6:.3e} env-steps/s ({sync_us / pair_us:.2f}x)")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The vec-env deployment shape: W worker PROCESSES on one GPU, each stepping its own host-buffer shard of E envs through
the C-ABI (numpy buffers, one handle per process), all at once. Prints the aggregate env-steps/s by W.

    python tools/time_workers.py [--workers 1 2 4 8] [--envs 1024] [--steps 3000]

Workers are spawned (never forked: the parent does not touch HIP either).
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, envs, steps, barrier, out):
    sys.path.insert(0, ROOT)
    from drone_amd import binding

    v = binding.DroneVec(envs, seed=0, env_offset=rank * envs)
    v.reset(0)
    v.fill_random_actions()
    for _ in range(100):
        v.step()
    barrier.wait()
    t0 = time.perf_counter()
    for _ in range(steps):
        v.step()
    el = time.perf_counter() - t0
    barrier.wait()
    out.put((rank, el, v.host_transport))
    v.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--threads", action="store_true", help="the workers are THREADS of one process (one handle and stream each) instead of processes")
    a = ap.parse_args()
    if a.threads:
        import queue
        import threading

        for w in a.workers:
            barrier, out = threading.Barrier(w), queue.Queue()
            ts = [threading.Thread(target=worker, args=(r, a.envs, a.steps, barrier, out)) for r in range(w)]
            for t in ts:
                t.start()
            for t in ts:
                t.join(300)
            res = [out.get() for _ in ts]
            slowest = max(r[1] for r in res)
            print(f"{w} worker thread(s) of one process x {a.envs} envs, host buffers ({res[0][2]}): {slowest / a.steps * 1e6:.1f} us per step of every worker, "
                  f"{w * a.envs * a.steps / slowest:.3e} env-steps/s in total", flush=True)
        return
    ctx = mp.get_context("spawn")
    for w in a.workers:
        barrier, out = ctx.Barrier(w), ctx.Queue()
        ps = [ctx.Process(target=worker, args=(r, a.envs, a.steps, barrier, out)) for r in range(w)]
        for p in ps:
            p.start()
        res = [out.get(timeout=300) for _ in ps]
        for p in ps:
            p.join(60)
        slowest = max(r[1] for r in res)
        print(f"{w} worker process(es) x {a.envs} envs, host buffers ({res[0][2]}): {slowest / a.steps * 1e6:.1f} us per step of every worker, "
              f"{w * a.envs * a.steps / slowest:.3e} env-steps/s in total", flush=True)


if __name__ == "__main__":
    main()

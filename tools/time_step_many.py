#!/usr/bin/env python3
"""Times drone_vec_step_many against one launch per step at several shard sizes (HIP events on the launch stream).
usage: time_step_many.py [--task hover] [--envs 65536,131072] [--ks 1,8,16,32,64] [--policy 0|1]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="hover")
    ap.add_argument("--envs", default="65536,131072,1048576")
    ap.add_argument("--ks", default="1,4,8,16,32,64")
    ap.add_argument("--policy", type=int, default=0)
    ap.add_argument("--budget-steps", type=int, default=2048, help="env steps timed per point")
    a = ap.parse_args()
    import torch

    from drone_amd import abi, binding

    task = {"hover": abi.TASK_HOVER, "waypoint": abi.TASK_WAYPOINT, "swarm": abi.TASK_SWARM, "race": abi.TASK_RACE}[a.task]
    for n in [int(x) for x in a.envs.split(",")]:
        v = binding.DroneVec(n, seed=0, task=task, device="cuda:0")
        v.reset(0)
        ring = [torch.empty_like(v.actions) for _ in range(4)]
        for k, r in enumerate(ring):
            v.fill_random_actions(gstep=k, out=r)
        for k in range(300):
            v.bind_actions(ring[k % 4]); v.step()
        torch.cuda.synchronize()
        v.timer_start()
        for k in range(a.budget_steps):
            v.bind_actions(ring[k % 4]); v.step()
        base = v.timer_stop() * 1e3 / a.budget_steps
        print(json.dumps({"envs": n, "task": a.task, "form": "one launch per step", "us_per_env_step": round(base, 3)}), flush=True)
        for K in [int(x) for x in a.ks.split(",")]:
            if K * n * 86 > 6 << 30:
                continue
            bufs = v.alloc_step_many(K)
            for k in range(K):
                v.fill_random_actions(gstep=k, out=bufs.actions[k])
            reps = max(3, a.budget_steps // K)
            for _ in range(3):
                v.step_many(bufs, policy=bool(a.policy))
            torch.cuda.synchronize()
            v.timer_start()
            for _ in range(reps):
                v.step_many(bufs, policy=bool(a.policy))
            us = v.timer_stop() * 1e3 / (reps * K)
            print(json.dumps({"envs": n, "task": a.task, "form": f"step_many K={K}" + (" policy" if a.policy else ""), "us_per_env_step": round(us, 3),
                              "us_per_launch": round(us * K, 2), "vs_per_step": round(us / base, 3)}), flush=True)
            del bufs
        v.close()


if __name__ == "__main__":
    main()

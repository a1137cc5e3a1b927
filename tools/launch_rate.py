#!/usr/bin/env python3
"""Host-side cost of one step call (launch-rate bound regime): tiny shard, no syncs inside the loop."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from drone_amd import abi, binding
for n in (4096, 65536, 131072):
    v = binding.DroneVec(n, seed=0, task=abi.TASK_HOVER, device="cuda:0")
    v.reset(0)
    ring = [torch.empty_like(v.actions) for _ in range(4)]
    for g, r in enumerate(ring): v.fill_random_actions(gstep=g, out=r)
    K = 20000
    for mode in ("bind+step", "step only", "raw ctypes step"):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode == "bind+step":
            for k in range(K):
                v.bind_actions(ring[k & 3]); v.step()
        elif mode == "step only":
            for k in range(K): v.step()
        else:
            f, h = v._step, v._h
            for k in range(K): f(h)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(json.dumps({"envs": n, "mode": mode, "host_us_per_step_issue": round((t1 - t0) * 1e6 / K, 2), "us_per_step_total": round((t2 - t0) * 1e6 / K, 2)}))
    v.close()

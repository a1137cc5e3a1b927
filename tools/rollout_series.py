#!/usr/bin/env python3
"""Per-launch duration of N consecutive fused-rollout launches (HIP events per launch): is the launch-to-launch
spread a clock ramp, the episode phase of the envs, or noise?  python tools/rollout_series.py [--launches 120]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from drone_amd import abi, binding  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=120)
ap.add_argument("--envs", type=int, default=1 << 20)
a = ap.parse_args()
for label, over in (("default (episodes end)", {}), ("no episode ends", {"bound": 1e6, "horizon": 10**9})):
    v = binding.DroneVec(a.envs, seed=0, task=abi.TASK_HOVER, device="cuda:0", **over)
    v.reset(0)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.launches + 1)]
    ev[0].record()
    for k in range(a.launches):
        v.rollout(128)
        ev[k + 1].record()
    torch.cuda.synchronize()
    us = [round(ev[k].elapsed_time(ev[k + 1]) * 1e3) for k in range(a.launches)]
    print(json.dumps({"config": label, "envs": a.envs, "us_per_launch_in_order": us}))
    v.close()

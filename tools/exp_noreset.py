#!/usr/bin/env python3
"""How much of the step kernel's time is the episode-end path (divergent reset code on ~half the waves, scattered
log-plane RMW and target-plane writes on ~1 % of the lanes)? Same kernel, same sizes, with and without episode ends
(huge box and horizon). Also repeats each measurement on freshly allocated handles to show placement spread."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from drone_amd import abi, binding  # noqa: E402


def run(n, task, steps, **over):
    v = binding.DroneVec(n, seed=0, task=task, device="cuda:0", **over)
    v.reset(0)
    ring = [torch.empty_like(v.actions) for _ in range(4)]
    for g, r in enumerate(ring):
        v.fill_random_actions(gstep=g, out=r)
    for k in range(150):  # past the first episode ends
        v.bind_actions(ring[k & 3]); v.step()
    torch.cuda.synchronize()
    v.timer_start()
    for k in range(steps):
        v.bind_actions(ring[k & 3]); v.step()
    us = v.timer_stop() * 1e3 / steps
    n_ep = v.log()["n"]
    v.close()
    return us, n_ep


for n in (1 << 20, 1 << 22, 131072):
    for name, over in (("default", {}), ("no_episode_ends", {"bound": 1e6, "horizon": 10**9}), ("default_again", {})):
        res = [run(n, abi.TASK_HOVER, 300, **over) for _ in range(3)]
        print(json.dumps({"envs": n, "config": name, "us": [round(r[0], 2) for r in res], "episodes": res[0][1]}))

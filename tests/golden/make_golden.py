#!/usr/bin/env python3
"""Generate the committed golden vectors from the CPU oracle.

SELF-REFERENTIAL BY CONSTRUCTION: the reference snapshot has no simulator
source, tests or vectors (/root/reference/.gitmodules:1-3),
so these fixtures pin THIS REPO's oracle (and through it the HIP path) against
regressions — they say nothing about parity with upstream. Regenerate only when
SPEC.md changes:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import pyoracle  # noqa: E402

CHECKPOINTS = (1, 10, 100, 1000)
N, SEED = 64, 20251017


def run(task, horizon, env_offset, **extra):
    cfg = pyoracle.default_config(task, horizon=horizon, env_offset=env_offset, **extra)
    v = pyoracle.OracleVec(N, seed=SEED, cfg=cfg)
    v.reset(SEED)
    out = {"reset_state": v.get_state().copy(), "reset_obs": v.observations.copy()}
    rew_trace = np.zeros((1000, N), np.float32)
    term_trace = np.zeros((1000, N), np.uint8)
    trunc_trace = np.zeros((1000, N), np.uint8)
    for t in range(1, 1001):
        v.fill_random_actions()
        if t == 1:
            out["actions_step1"] = v.actions.copy()
        v.step()
        rew_trace[t - 1], term_trace[t - 1], trunc_trace[t - 1] = v.rewards, v.terminals, v.truncations
        if t in CHECKPOINTS:
            out[f"state_{t}"] = v.get_state().copy()
            out[f"obs_{t}"] = v.observations.copy()
    out["rewards"] = rew_trace
    out["terminals"] = np.packbits(term_trace, axis=1)
    out["truncations"] = np.packbits(trunc_trace, axis=1)
    log = v.log()
    out["log"] = np.array([log[k] for k in ("perf", "score", "episode_return", "episode_length", "oob", "n")], np.float32)
    return out


def main():
    for name, task, horizon, off, extra in (("hover", 0, 1024, 0, {}), ("waypoint", 1, 1024, 0, {}), ("hover_h100_off", 0, 100, 1 << 20, {}),
                                             ("swarm", 2, 300, 64, {"collision_radius": 0.5}), ("race", 3, 200, 0, {"gate_radius": 2.5})):
        np.savez_compressed(os.path.join(HERE, f"golden_{name}.npz"), **run(task, horizon, off, **extra))
        print("wrote", name)


if __name__ == "__main__":
    main()

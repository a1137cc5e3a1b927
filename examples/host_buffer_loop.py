#!/usr/bin/env python3
"""The drop-in shape for a CPU-side trainer: numpy buffers the CALLER owns (plain heap arrays, as a vec-env hands them to
its envs), the compiled binding's PufferLib-style vec_* calls, and the step in two halves — vec_send starts the env step,
the "policy" (a numpy matmul here) works on the previous observations meanwhile, vec_recv delivers the new ones.

    python examples/host_buffer_loop.py [--envs 4096] [--steps 1000] [--task 0]

Prints env-steps/s for the plain loop (policy, then vec_step) and for the overlapped one (vec_send, policy, vec_recv; the
policy then lags one step), and which host transport the library chose for these buffers (0 mirror, 1 zero-copy, 2
zero-copy through pinned stand-ins).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from drone_amd import drone_binding as binding  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--task", type=int, default=binding.TASK_HOVER)
    a = ap.parse_args()
    n, od = a.envs, binding.obs_dim(a.task)
    obs, act = np.zeros((n, od), np.float32), np.zeros((n, 4), np.float32)
    rew, term, trunc = np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    env = binding.vec_init(obs, act, rew, term, trunc, n, 0, task=a.task)
    binding.vec_reset(env, 0)
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((od, 4)) * 0.3).astype(np.float32)

    def policy(o, out):  # a linear policy squashed to [-1, 1]
        np.tanh(o @ w, out=out)

    for _ in range(20):
        policy(obs, act)
        binding.vec_step(env)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        policy(obs, act)
        binding.vec_step(env)
    plain = time.perf_counter() - t0

    # overlapped, with one step of policy lag: the action for step t+1 is computed from the observations of step t-1
    # (a private copy — the env owns `obs` between send and recv) while the GPU runs step t
    nxt, last = np.zeros_like(act), obs.copy()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        binding.vec_send(env)        # reads `act`, starts the step; obs / rew / term / trunc must not be touched ...
        policy(last, nxt)            # ... while this runs ...
        binding.vec_recv(env)        # ... until this returns
        last[:] = obs
        act[:] = nxt
    overlapped = time.perf_counter() - t0

    log = binding.vec_log(env)
    print(f"{n} envs, caller-owned numpy heap buffers (host transport {binding.vec_host_transport(env)}): policy then vec_step "
          f"{n * a.steps / plain:.3e} env-steps/s ({plain / a.steps * 1e6:.1f} us per step); vec_send, policy, vec_recv "
          f"{n * a.steps / overlapped:.3e} ({overlapped / a.steps * 1e6:.1f} us); episodes {log['n']:.0f}, mean return {log['episode_return']:.2f}")
    binding.vec_close(env)


if __name__ == "__main__":
    main()

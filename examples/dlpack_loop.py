#!/usr/bin/env python3
"""The env through the compiled binding alone, with buffers the LIBRARY owns in HBM, handed to a torch policy as DLPack
capsules. Nothing on the env side imports torch or allocates through it; any other DLPack consumer
(cupy, jax) would take the same capsules.

    python examples/dlpack_loop.py [--envs 65536] [--steps 300] [--task 0]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from drone_amd import dlpack  # noqa: E402
from drone_amd import drone_binding as binding  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--task", type=int, default=binding.TASK_HOVER)
    a = ap.parse_args()
    env = binding.vec_init(None, None, None, None, None, a.envs, 0, task=a.task)
    obs, act, rew, term, trunc = (torch.from_dlpack(b) for b in dlpack.buffers(env))  # protocol objects over binding.vec_dlpack capsules
    od = binding.obs_dim(a.task)
    policy = torch.nn.Sequential(torch.nn.Linear(od, 64), torch.nn.Tanh(), torch.nn.Linear(64, 4), torch.nn.Tanh()).to(obs.device)
    stream = torch.cuda.Stream()
    binding.vec_set_stream(env, stream.cuda_stream)  # env and policy on one stream: no syncs between them
    ret = torch.zeros(a.envs, device=obs.device)
    with torch.no_grad(), torch.cuda.stream(stream):
        binding.vec_reset(env, 0)
        for _ in range(10):
            act.copy_(policy(obs))
            binding.vec_step(env)
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            act.copy_(policy(obs))
            binding.vec_step(env)  # obs / rew / term / trunc updated in place
            ret += rew
        stream.synchronize()
        el = time.perf_counter() - t0
    log = binding.vec_log(env)
    print(f"{a.envs} envs x {a.steps} steps, library-owned HBM buffers through DLPack: {a.envs * a.steps / el:.3e} env-steps/s "
          f"including the policy; episodes {log['n']:.0f}, mean return {log['episode_return']:.3f}, dones now {int((term | trunc).sum())}")
    del obs, act, rew, term, trunc  # the views keep the env alive; vec_close insists they are gone
    binding.vec_close(env)


if __name__ == "__main__":
    main()

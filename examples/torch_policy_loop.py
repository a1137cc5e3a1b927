#!/usr/bin/env python3
"""Device-resident RL loop: a small torch-ROCm MLP policy reads the env's
observation tensor in HBM and writes the action tensor the env reads — no host
copies, one stream. Prints env-steps/s including the policy,
first launched eagerly (a dozen launches per step: host-bound at small shards),
then with "policy forward + env step" captured ONCE into a hipGraph
(torch.cuda.graph) and replayed: graph-safe stepping (drone_vec_enable_graph_capture)
keeps the step counter in HBM so the replays advance it. Last: the same graph
with a frame skip of 4 — one policy forward, then four env steps under that
action in ONE launch (drone_vec_step_repeat), the policy reading the last of the
four observation blocks.

    python examples/torch_policy_loop.py [--envs 65536] [--steps 500] [--task hover]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from drone_amd.env import Drone  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--task", default="hover")
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--skip", type=int, default=4, help="frame skip of the last loop (env steps per policy step)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    env = Drone(num_envs=a.envs, task=a.task, device=dev, seed=0, log_interval=0)
    policy = torch.nn.Sequential(torch.nn.Linear(20, a.hidden), torch.nn.Tanh(), torch.nn.Linear(a.hidden, a.hidden), torch.nn.Tanh(),
                                 torch.nn.Linear(a.hidden, 4), torch.nn.Tanh()).to(dev)
    obs, _ = env.reset(0)
    ret = torch.zeros(a.envs, device=dev)
    with torch.no_grad():
        for _ in range(20):
            env.actions.copy_(policy(obs))
            obs, rew, term, trunc, _ = env.step(env.actions)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            env.actions.copy_(policy(obs))  # the policy writes the env's own action tensor
            obs, rew, term, trunc, _ = env.step(env.actions)
            ret += rew
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    log = env.vec.log()
    print(f"{a.envs} envs x {a.steps} steps with a 20-{a.hidden}-{a.hidden}-4 tanh MLP policy on the same stream: "
          f"{a.envs * a.steps / el:.3e} env-steps/s ({el * 1e6 / a.steps:.1f} us per step); "
          f"episodes {log['n']:.0f}, mean return {log['episode_return']:.3f}, mean length {log['episode_length']:.1f}")

    # the same loop as ONE captured graph per step
    vec = env.vec
    vec.enable_graph_capture(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.no_grad():
        with torch.cuda.stream(side):
            vec.use_torch_stream()
            for _ in range(3):
                vec.actions.copy_(policy(vec.observations))
                vec.step()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            vec.use_torch_stream()
            vec.actions.copy_(policy(vec.observations))
            vec.step()
        g0 = vec.gstep
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            g.replay()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    assert vec.gstep == g0 + a.steps
    log = vec.log()
    print(f"captured as one hipGraph per step and replayed: {a.envs * a.steps / el:.3e} env-steps/s ({el * 1e6 / a.steps:.1f} us per step); "
          f"episodes {log['n']:.0f}, mean return {log['episode_return']:.3f}")

    # frame skip: policy forward + K env steps under that action in one launch, captured once
    K = a.skip
    bufs = vec.alloc_step_many(K)
    with torch.no_grad():
        with torch.cuda.stream(side):
            vec.use_torch_stream()
            bufs.observations[K - 1].copy_(vec.observations)
            for _ in range(3):
                vec.actions.copy_(policy(bufs.observations[K - 1]))
                vec.step_repeat(bufs)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            vec.use_torch_stream()
            vec.actions.copy_(policy(bufs.observations[K - 1]))  # the observation after the previous K steps
            vec.step_repeat(bufs)
        g0 = vec.gstep
        calls = max(1, a.steps // K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            g.replay()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    assert vec.gstep == g0 + calls * K
    print(f"frame skip {K} (one policy forward + drone_vec_step_repeat per replay): {a.envs * calls * K / el:.3e} env-steps/s "
          f"({el * 1e6 / (calls * K):.1f} us per env step, {el * 1e6 / calls:.1f} us per policy step)")
    env.close()


if __name__ == "__main__":
    main()

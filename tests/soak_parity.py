#!/usr/bin/env python3
"""Randomised call-sequence parity soak: the HIP path against the oracle, bit for bit.

Each case draws a task, a shard size (ragged sizes included), buffer kind, horizon, substeps, time step, scaled physics
constants, env offset, the per-handle kernel variants (derived-target layout, packed RK4, graph-safe counters) and then a
random SEQUENCE of path calls — step (random-policy, hostile or repeated actions, or the product's own fill_random_actions; as one call or as step_send + step_recv), rebinds of the action / output buffers, step_many (caller actions or the
in-kernel policy), step_repeat, fused rollout, log, state read-back, a checkpoint round trip through get_state /
set_state / set_gstep — mirrored call for call on the oracle. Every output of every call, every done-id list and the
final state must match bit for bit.

Test infrastructure (it drives the oracle): lives under tests/. `tests/test_soak_gpu.py` runs a short, fixed-seed slice
in the GPU suite; run it longer by hand:

    python tests/soak_parity.py --minutes 20 --seed 1 [--log profiles/r03_soak.txt]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from drone_amd import abi  # noqa: E402
from helpers import assert_bits_equal, assert_outputs_equal, assert_state_equal  # noqa: E402

SIZES = [1, 2, 63, 64, 65, 127, 255, 256, 257, 1000, 1024, 4097, 8192, 16384 + 17, 32768]
BIG_SIZES = [65536 + 1, 131072, 262144 + 65, 524288 - 63, 1048576, 1048576 + 3,  # --big: many workgroups per XCD, the size-dependent layout choices taken for real
             1835008 + 1, 2621440, 3145728 - 63]  # round 4: footprints of 450 ... 900 MiB per step, where the sweep-order / non-temporal-load bands switch
SCALED = ["mass", "arm", "ixx", "iyy", "izz", "k_thrust", "k_torque", "k_drag", "k_ang_damp", "motor_tau", "max_vel", "max_omega",
          "bound", "spawn_extent", "target_extent", "tilt_init", "hover_radius", "waypoint_radius", "wind_theta", "wind_sigma", "wind_max",
          "c_omega", "c_action", "crash_penalty", "progress_scale", "waypoint_bonus", "collision_radius", "proximity_radius", "c_proximity", "gate_radius"]


def draw_case(rng, big=False):
    task = int(rng.integers(0, 4))
    n = int(rng.choice(BIG_SIZES if big else SIZES))
    over = {"horizon": int(rng.choice([1, 2, 7, 33, 100, 300, 1000])), "substeps": int(rng.integers(1, 4)),
            "dt": float(rng.choice([0.005, 0.01, 0.02, 0.03])), "compact_done": int(rng.integers(0, 2))}
    agents = 1
    if task == abi.TASK_SWARM:
        agents = int(2 ** rng.integers(0, 7))
        n = max(agents, n // agents * agents)
        over["agents_per_env"] = agents
    over["env_offset"] = int(rng.integers(0, 1 << 20)) // agents * agents
    if rng.random() < 0.4:  # round 4: the state layout as a declared choice (DroneConfig.state_layout); derived target only where it can hold the handle
        over["state_layout"] = int(rng.choice([abi.LAYOUT_TARGET_PLANE, abi.LAYOUT_DERIVED_TARGET] if task in (abi.TASK_HOVER, abi.TASK_SWARM) else [abi.LAYOUT_TARGET_PLANE]))
    if rng.random() < 0.5:  # scale a few physics / task constants; every kernel reads them from the same KParams
        for name in rng.choice(SCALED, size=int(rng.integers(1, 6)), replace=False):
            over[str(name)] = ("scale", float(rng.uniform(0.5, 2.0)))
    env = {"DRONE_DERIVED_TARGET": rng.choice(["", "0", "1"]), "DRONE_PACKED_RK4": rng.choice(["", "0", "1"]),
           "DRONE_LINE_COMPLETE": rng.choice(["", "0", "1"]), "DRONE_SWEEP_ORDER": rng.choice(["", "0", "1", "6", "8", "15"]),
           "DRONE_HOST_ZEROCOPY": rng.choice(["", "0", "1"]),
           # round 5: a small single-memcpy budget sends heap-buffer handles from ~1000 envs on through the host copy pool (transport 3:
           # stand-ins delivered chunk by chunk while the kernel runs), not only the 16 401- and 32 768-env draws that exceed the default MiB
           "DRONE_HOST_BOUNCE_MAX_BYTES": rng.choice(["", "", "60000"])}
    return {"task": task, "n": n, "seed": int(rng.integers(0, 1 << 62)), "device": bool(rng.integers(0, 2)), "over": over, "env": env,
            "graph_safe": bool(rng.random() < 0.25), "ops": int(rng.integers(3, 7) if big else rng.integers(4, 14)), "max_k": 3 if big else 9,
            "heap_buffers": bool(rng.random() < 0.3)}  # host handles: plain numpy arrays (pages shared with the heap: never pinned) instead of page-owning ones


def make_cfg(mod, task, over):
    cfg = mod.default_config(task)
    for k, v in over.items():
        if isinstance(v, tuple):
            setattr(cfg, k, getattr(cfg, k) * v[1])
        else:
            setattr(cfg, k, v)
    return cfg


def hostile_actions(rng, n):
    a = rng.normal(0.0, 1.5, size=(n, 4)).astype(np.float32)  # mostly out of [-1, 1]: the clamp is exercised
    bad = rng.random((n, 4))
    a[bad < 0.01] = np.nan
    a[(bad >= 0.01) & (bad < 0.02)] = np.inf
    a[(bad >= 0.02) & (bad < 0.03)] = -np.inf
    return a


def put(dst, src):
    if type(dst).__module__.startswith("torch"):
        import torch

        dst.copy_(torch.from_numpy(np.ascontiguousarray(src)))
        torch.cuda.synchronize()
    else:
        dst[...] = src


def to_np_(x):
    return x.cpu().numpy() if type(x).__module__.startswith("torch") else np.asarray(x)


def sync(v):
    if v.torch_device is not None:
        v.sync()


def fresh_buffers(case):
    """A set of caller-owned buffers of the case's kind, NOT page-owning for host handles: (obs, act, rew, term, trunc)."""
    n, od = case["n"], abi.obs_dim(case["task"])
    if case["device"]:
        import torch

        return (torch.zeros(n, od, device="cuda:0"), torch.zeros(n, 4, device="cuda:0"), torch.zeros(n, device="cuda:0"),
                torch.zeros(n, dtype=torch.uint8, device="cuda:0"), torch.zeros(n, dtype=torch.uint8, device="cuda:0"))
    return (np.zeros((n, od), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros(n, np.uint8))


def make_vec(binding, case):
    """The product handle of a case; the per-handle kernel variants are chosen from the environment at init."""
    saved = {k: os.environ.get(k) for k in case["env"]}
    for k, val in case["env"].items():
        if val:
            os.environ[k] = str(val)
        else:
            os.environ.pop(k, None)
    try:
        v = binding.DroneVec(case["n"], seed=case["seed"], cfg=make_cfg(binding, case["task"], case["over"]), device="cuda:0" if case["device"] else None,
                             buffers=fresh_buffers(case) if (case.get("heap_buffers") and not case["device"]) else None)
    finally:
        for k, val in saved.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val
    if case["graph_safe"] and case["device"]:
        v.enable_graph_capture(True)
    return v


def run_case(binding, oracle, case, rng, threads=8):
    """Returns the number of env-steps compared."""
    task, n = case["task"], case["n"]
    v = make_vec(binding, case)
    o = oracle.OracleVec(n, seed=case["seed"], cfg=make_cfg(oracle, task, case["over"]), threads=threads)
    tag = f"case {case}"
    v.reset(case["seed"])
    o.reset(case["seed"])
    sync(v)
    assert_bits_equal(o.observations, v.observations, tag + " reset obs")
    env_steps = 0
    compact = bool(case["over"]["compact_done"])
    history = []
    for i in range(case["ops"]):
        op = str(rng.choice(["step", "step", "sendrecv", "hostile", "repeat_last", "many", "many_policy", "step_repeat", "rollout", "log", "state", "checkpoint", "rebind", "fill"]))
        history.append(op)
        what = f"{tag} after {history[:-1]} op {i} {op}"
        if op in ("step", "sendrecv", "hostile", "repeat_last"):
            for _ in range(int(rng.integers(1, 6))):
                if op in ("step", "sendrecv"):
                    o.fill_random_actions()
                elif op == "hostile":
                    o.actions[:] = hostile_actions(rng, n)
                put(v.actions, o.actions)
                o.step()
                if op == "sendrecv":  # the step in two halves
                    v.step_send()
                    v.step_recv()
                else:
                    v.step()
                sync(v)
                assert_outputs_equal(o, v, what)
                if compact:
                    want = np.flatnonzero(o.terminals | o.truncations).astype(np.uint32)
                    assert_bits_equal(want, np.sort(v.done_list()), what + " done list")
                env_steps += n
        elif op in ("many", "many_policy", "step_repeat"):
            K = int(rng.integers(1, case.get("max_k", 9) + 1))
            bufs = v.alloc_step_many(K, pinned=bool(rng.integers(0, 2)))  # host handles: pinned blocks are accessed in place, others staged
            if op == "many":
                acts = np.stack([hostile_actions(rng, n) if rng.random() < 0.3 else rng.uniform(-1, 1, (n, 4)).astype(np.float32) for _ in range(K)])
                put(bufs.actions, acts)
                oo = o.step_many(K, acts)
                v.step_many(bufs)
            elif op == "many_policy":
                oo = o.step_many(K, None)
                v.step_many(bufs, policy=True)
            else:
                a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
                put(v.actions, a)
                oo = o.step_many(K, np.broadcast_to(a, (K, n, 4)))
                v.step_repeat(bufs)
            sync(v)
            for name, want, got in (("obs", oo[0], bufs.observations), ("rew", oo[1], bufs.rewards), ("term", oo[2], bufs.terminals), ("trunc", oo[3], bufs.truncations)):
                assert_bits_equal(want, got, f"{what} K={K} {name}")
            if compact:
                for k in range(K):
                    assert_bits_equal(oo[4][k], np.sort(v.done_list_at(k)), f"{what} done list of step {k}")
            env_steps += n * K
        elif op == "rollout":
            h = int(rng.integers(1, 12 if n > 65536 else 48))
            o.rollout(h)
            v.rollout(h)
            sync(v)
            assert_outputs_equal(o, v, what + f" horizon {h}")
            env_steps += n * h
        elif op == "log":
            lo, lv = o.log(), v.log()
            assert lo["n"] == lv["n"], f"{what}: episode count {lo} vs {lv}"
            for key in lo:  # means of float sums reduced in a different order: close, not bitwise (per-env sums are checked through the state)
                assert abs(lo[key] - lv[key]) <= 2e-5 * max(1.0, abs(lo[key])), f"{what}: {key} {lo[key]} vs {lv[key]}"
        elif op == "state":
            first = int(rng.integers(0, n))
            count = int(rng.integers(1, n - first + 1))
            assert_state_equal(o.get_state(first, count), v.get_state(first, count), what + f" rows [{first}, {first + count})")
        elif op == "rebind":
            # the caller moves to other buffers (a second output set for overlap, a pre-filled action buffer) and lets the old ones go
            sync(v)
            obs, act, rew, term, trunc = fresh_buffers(case)
            which = int(rng.integers(0, 3))
            if which != 1:
                v.bind_outputs(obs, rew, term, trunc)
            if which != 0:
                put(act, to_np_(v.actions))
                v.bind_actions(act)
        elif op == "fill":
            # the product's own random policy into ITS action buffer, then a step from it
            g = o.gstep
            o.fill_random_actions(g)
            v.fill_random_actions(g)
            sync(v)
            assert_bits_equal(o.actions, v.actions, what + " actions")
            o.step()
            v.step()
            sync(v)
            assert_outputs_equal(o, v, what)
            env_steps += n
        elif op == "checkpoint":
            # restore the oracle's rows into a FRESH product handle and continue from there
            rows, g = o.get_state(), o.gstep
            v2 = make_vec(binding, case)
            v2.reset(case["seed"])  # the seed keys every RNG stream of the handle: a checkpoint is restored under the same seed
            v2.rollout(3)           # some other state and step counter first
            v2.set_state(rows)
            v2.set_gstep(g)
            v.close()
            v = v2
            assert_state_equal(rows, v.get_state(), what + " restored rows")
    assert_state_equal(o.get_state(), v.get_state(), tag + " final state")
    assert v.gstep == o.gstep, f"{tag}: step counter {v.gstep} vs {o.gstep}"
    v.close()
    o.close()
    return env_steps


def soak(binding, oracle, seed, cases=None, minutes=None, log=None, threads=8, big=False):
    rng = np.random.default_rng(seed)
    t0 = time.time()
    done = env_steps = 0
    per_task = [0, 0, 0, 0]
    while (cases is None or done < cases) and (minutes is None or time.time() - t0 < minutes * 60):
        case = draw_case(rng, big)
        env_steps += run_case(binding, oracle, case, rng, threads)
        per_task[case["task"]] += 1
        done += 1
        if log and done % (5 if big else 50) == 0:
            print(f"[soak] {done} cases, {env_steps:.3e} env-steps compared, {time.time() - t0:.0f} s", file=log, flush=True)
    return {"cases": done, "env_steps_compared": env_steps, "cases_per_task": per_task, "seconds": time.time() - t0, "seed": seed}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--cases", type=int, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--log", default=None)
    ap.add_argument("--big", action="store_true", help="shard sizes 65 537 ... 2^20 + 3 instead of 1 ... 32 768 (fewer, shorter sequences)")
    a = ap.parse_args()
    from drone_amd import binding
    from oracle import pyoracle

    binding.load()
    pyoracle.lib()
    out = open(a.log, "a") if a.log else sys.stdout
    res = soak(binding, pyoracle, a.seed, cases=a.cases, minutes=a.minutes, log=out, threads=a.threads, big=a.big)
    print(f"[soak] PASS seed {res['seed']}{' --big' if a.big else ''}: {res['cases']} random call sequences (hover / waypoint / swarm / race: {res['cases_per_task']}), "
          f"{res['env_steps_compared']:.4e} env-steps compared bit for bit in {res['seconds']:.0f} s", file=out, flush=True)
    if a.log:
        out.close()


if __name__ == "__main__":
    main()

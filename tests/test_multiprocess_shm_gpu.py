"""PufferLib's Multiprocessing vec-env shape on the GPU path: ONE shared-memory block
per buffer kind, one worker PROCESS per shard writing its slice in place. With one
process per GPU this is the multi-GPU design for a host-side consumer — no collective
at all: every GPU's kernel stores its slice straight into the shared block over its own
PCIe link (host zero-copy transport). Here both workers share the box's single GPU."""
import multiprocessing as mp
import os
import sys
from multiprocessing import shared_memory

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = (("observations", np.float32, 20), ("actions", np.float32, 4), ("rewards", np.float32, 1), ("terminals", np.uint8, 1), ("truncations", np.uint8, 1))


def views(names, total):
    out, keep = {}, []
    for (k, dt, w), name in zip(KINDS, names):
        shm = shared_memory.SharedMemory(name=name)
        keep.append(shm)
        shape = (total, w) if w > 1 else (total,)
        out[k] = np.ndarray(shape, dtype=dt, buffer=shm.buf)
    return out, keep


def worker(rank, world, names, total, seed, steps, go, done):
    sys.path.insert(0, ROOT)
    from types import SimpleNamespace

    from drone_amd.env import Drone

    block, keep = views(names, total)
    n = total // world
    sl = slice(rank * n, (rank + 1) * n)
    buf = SimpleNamespace(**{k: v[sl] for k, v in block.items()})
    env = Drone(num_envs=n, task="waypoint", seed=seed, log_interval=0, buf=buf, env_offset=rank * n, horizon=45,
                host_pages_exclusive=1)  # a shared-memory block is a mapping of its own: the worker may vouch for its slices (round 5: alignment alone no longer registers)
    env.reset(seed)
    done.wait()
    for _ in range(steps):
        go.wait()           # the parent has written this step's actions into the shared block
        env.step(env.actions)
        done.wait()
    env.close()
    del block, buf
    for s in keep:
        s.close()


def test_two_worker_processes_share_one_block(oracle):
    total, world, seed, steps = 4096, 2, 19, 60
    shms = []
    for k, dt, w in KINDS:
        shms.append(shared_memory.SharedMemory(create=True, size=total * w * np.dtype(dt).itemsize))
    try:
        names = [s.name for s in shms]
        block, keep = views(names, total)
        ctx = mp.get_context("spawn")
        go, done = ctx.Barrier(world + 1), ctx.Barrier(world + 1)
        procs = [ctx.Process(target=worker, args=(r, world, names, total, seed, steps, go, done)) for r in range(world)]
        for p in procs:
            p.start()
        o = oracle.OracleVec(total, seed=seed, cfg=oracle.default_config(1, horizon=45), threads=8)
        o.reset(seed)
        done.wait(timeout=120)  # both workers have reset
        assert block["observations"].tobytes() == o.observations.tobytes()
        for t in range(steps):
            o.fill_random_actions()
            block["actions"][:] = o.actions
            go.wait(timeout=60)
            o.step()
            done.wait(timeout=60)
            assert block["observations"].tobytes() == o.observations.tobytes(), t
            assert block["rewards"].tobytes() == o.rewards.tobytes(), t
            assert block["terminals"].tobytes() == o.terminals.tobytes() and block["truncations"].tobytes() == o.truncations.tobytes(), t
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        del block
        for s in keep:
            s.close()
    finally:
        for s in shms:
            s.close()
            s.unlink()

"""N>1 path on CPU: world_size-2 gloo run of the sharding + host-boundary gather
logic (drone_amd/dist.py). The per-rank compute here is the CPU oracle standing
in for the GPU shard (the product has no CPU path); what is under test is the
shard split, the global-env-id offset and the gather order."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, steps, task, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from drone_amd import abi
    from drone_amd.dist import BoundaryGather, shard_range
    from oracle import pyoracle

    off, cnt = shard_range(total, rank, world)
    v = pyoracle.OracleVec(cnt, seed=5, cfg=pyoracle.default_config(task, env_offset=off, horizon=30))
    v.reset(5)
    g = BoundaryGather(total, abi.OBS_DIM, torch.device("cpu"))
    for _ in range(steps):
        v.fill_random_actions()
        v.step()
        obs, rew, term, trunc = g(torch.from_numpy(v.observations), torch.from_numpy(v.rewards),
                                  torch.from_numpy(v.terminals), torch.from_numpy(v.truncations))
    if rank == 0:
        q.put((obs.numpy().copy(), rew.numpy().copy(), term.numpy().copy(), trunc.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [256, 257])
def test_two_rank_shards_equal_single_vec(oracle, total):
    steps, task, world = 50, 1, 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29500 + (os.getpid() % 2000) + total % 7
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, steps, task, q)) for r in range(world)]
    for p in procs:
        p.start()
    obs, rew, term, trunc = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    whole = oracle.OracleVec(total, seed=5, cfg=oracle.default_config(task, horizon=30))
    whole.reset(5)
    for _ in range(steps):
        whole.fill_random_actions()
        whole.step()
    assert obs.tobytes() == whole.observations.tobytes()
    assert rew.tobytes() == whole.rewards.tobytes()
    assert term.tobytes() == whole.terminals.tobytes() and trunc.tobytes() == whole.truncations.tobytes()
    assert term.sum() + trunc.sum() >= 0


def test_shard_range_covers_everything():
    from drone_amd.dist import shard_counts, shard_range

    for total in (1, 7, 8, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (o0, c0), (o1, _) in zip(spans, spans[1:]):
                assert o0 + c0 == o1
            assert spans[-1][0] + spans[-1][1] == total
            assert sum(shard_counts(total, world)) == total
    assert shard_range(1 << 20, 3, 8) == (3 * 131072, 131072)

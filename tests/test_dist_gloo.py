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


def _root_worker(rank, world, port, total, steps, task, root, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from drone_amd import abi
    from drone_amd.dist import RootGather, shard_range
    from oracle import pyoracle

    off, cnt = shard_range(total, rank, world)
    v = pyoracle.OracleVec(cnt, seed=5, cfg=pyoracle.default_config(task, env_offset=off, horizon=30))
    v.reset(5)
    g = RootGather(total, abi.OBS_DIM, torch.device("cpu"), root=root)
    assert (g.obs is not None) == (rank == root)  # only the root holds global buffers
    out = None
    for _ in range(steps):
        v.fill_random_actions()
        v.step()
        out = g(torch.from_numpy(v.observations), torch.from_numpy(v.rewards), torch.from_numpy(v.terminals), torch.from_numpy(v.truncations))
        assert (out is not None) == (rank == root)
    if rank == root:
        q.put(tuple(t.numpy().copy() for t in out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total,root", [(2, 256, 0), (2, 257, 1), (3, 1000, 2), (8, 1029, 5)])  # the last: eight ranks, ragged shards (129 / 128 envs), a root in the middle
def test_gather_to_root_equals_single_vec(oracle, world, total, root):
    """RootGather on gloo: equal and ragged shards, first / last rank as the root; the root's batch is one oracle run's."""
    steps, task = 40, 1
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29700 + (os.getpid() % 2000) + total % 11 + root
    procs = [ctx.Process(target=_root_worker, args=(r, world, port, total, steps, task, root, q)) for r in range(world)]
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
    assert obs.tobytes() == whole.observations.tobytes() and rew.tobytes() == whole.rewards.tobytes()
    assert term.tobytes() == whole.terminals.tobytes() and trunc.tobytes() == whole.truncations.tobytes()


class _OracleAsVec:
    """Gives the CPU oracle the bind_outputs()/step() surface PipelinedGather drives."""

    def __init__(self, o):
        self.o = o
        self.observations = torch.from_numpy(o.observations)
        self.rewards = torch.from_numpy(o.rewards)
        self.terminals = torch.from_numpy(o.terminals)
        self.truncations = torch.from_numpy(o.truncations)
        self._bound = None

    def bind_outputs(self, obs, rew, term, trunc):
        self._bound = (obs, rew, term, trunc)

    def step(self):
        self.o.fill_random_actions()
        self.o.step()
        if self._bound is not None:
            for dst, src in zip(self._bound, (self.o.observations, self.o.rewards, self.o.terminals, self.o.truncations)):
                dst.copy_(torch.from_numpy(src))


def _pipelined_worker(rank, world, port, total, steps, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from drone_amd import abi
    from drone_amd.dist import PipelinedGather, shard_range
    from oracle import pyoracle

    off, cnt = shard_range(total, rank, world)
    o = pyoracle.OracleVec(cnt, seed=9, cfg=pyoracle.default_config(0, env_offset=off, horizon=25))
    o.reset(9)
    pg = PipelinedGather(_OracleAsVec(o), total, abi.OBS_DIM)
    trace = []
    for _ in range(steps):
        (obs, rew, term, trunc), ev = pg.step()
        assert ev is None
        trace.append((obs.numpy().copy(), rew.numpy().copy(), term.numpy().copy()))
    if rank == 0:
        q.put(trace)
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gather_alternates_output_sets(oracle):
    total, steps, world = 192, 7, 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29400 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pipelined_worker, args=(r, world, port, total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    trace = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    whole = oracle.OracleVec(total, seed=9, cfg=oracle.default_config(0, horizon=25))
    whole.reset(9)
    for t in range(steps):
        whole.fill_random_actions()
        whole.step()
        obs, rew, term = trace[t]
        assert obs.tobytes() == whole.observations.tobytes(), t
        assert rew.tobytes() == whole.rewards.tobytes() and term.tobytes() == whole.terminals.tobytes(), t


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


class _FakePeerVec:
    """The surface PeerStoreGather's CONSTRUCTOR touches, without a GPU: what is under test is its own choreography — the
    export on the root, the token and the page's name over torch.distributed, every rank learning whether all could join."""

    class _Cfg:
        task = 0

    def __init__(self, rank, fail_rank):
        self.torch_device, self.cfg, self.rank, self.fail_rank = torch.device("cpu"), self._Cfg(), rank, fail_rank
        self.closed = False

    def gather_peer_export(self, obs, rew, term, trunc):
        return bytes(288)

    def gather_init_peer(self, token, flags, rank, world, root=0, counts=None):
        assert len(token) == 288 and flags.shape == (1024,)
        if rank == self.fail_rank:
            raise RuntimeError(f"rank {rank} could not map the root's batch (injected)")

    def gather_close(self):
        self.closed = True


def _peer_ctor_worker(rank, world, port, fail_rank, q):
    import time

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from drone_amd.dist import PeerStoreGather

    v, t0, err = _FakePeerVec(rank, fail_rank), time.time(), None
    try:
        PeerStoreGather(v, 64 * world, root=0)
    except RuntimeError as exc:
        err = str(exc)
    q.put((rank, err, v.closed, time.time() - t0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [-1, 1])
def test_peer_store_gather_constructor_fails_on_every_rank_together(fail_rank):
    """a rank that could not join the exchange (memmap, IPC open) used to skip the barrier the others then sat in until
    the process-group timeout. Now every rank learns, in ONE collective that also serves as that barrier, whether all could
    join: either all return, or all raise at once — the ranks that had joined undo it — and the flag page's name is gone either way."""
    world = 2
    before = {f for f in os.listdir("/dev/shm") if f.startswith("drone_peer_flags_")}
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29900 + (os.getpid() % 1500) + fail_rank + 2
    procs = [ctx.Process(target=_peer_ctor_worker, args=(r, world, port, fail_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get() for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    if fail_rank < 0:
        assert all(err is None and not closed for _, err, closed, _ in got), got
    else:
        assert "injected" in got[1][1] and "could not join" in got[0][1] and got[0][2], got  # the healthy rank undid its join
    assert all(took < 30 for *_, took in got), got
    assert {f for f in os.listdir("/dev/shm") if f.startswith("drone_peer_flags_")} == before

"""Sharding of envs over the GPUs of one node.

Envs are independent, so the path shards with no data-path collective: rank r
owns a contiguous range of global env ids and passes its start as
``env_offset`` (the counter RNG is keyed on the GLOBAL id, SPEC.md §2, so the
trajectory of env e does not depend on the shard it lands in). The only
exchange is at the host boundary: an all-gather of observations / rewards /
flags for a consumer that wants the whole batch on every rank (RCCL over xGMI
when the tensors are in HBM; the same code runs on gloo with CPU tensors), or
— ``RootGather`` — a gather to ONE rank for a single consumer process: every
other rank sends its rows once and receives nothing; or — ``PeerStoreGather``
(round 4) — no collective at all: the other ranks' step kernels write their
rows straight into the root's IPC-mapped batch (``drone_vec_gather_init_peer``)
and the "gather" is a flag handshake.
"""
import contextlib

import torch
import torch.distributed as dist


def _one_launch(group, device):
    """The four all-gathers of one exchange as ONE grouped RCCL launch (ncclGroupStart / End) where torch offers it:
    at 11 MB per rank the per-collective launch latency is comparable to the transfer itself. Plain sequential
    collectives otherwise (gloo, or a torch without the coalescing manager)."""
    cm = getattr(dist.distributed_c10d, "_coalescing_manager", None)
    if cm is None or device.type != "cuda":
        return contextlib.nullcontext()
    return cm(group=group, device=device, async_ops=False)


def shard_range(total_envs, rank, world):
    """Contiguous split; the first ``total % world`` ranks get one extra env."""
    base, extra = divmod(int(total_envs), int(world))
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def shard_counts(total_envs, world):
    return [shard_range(total_envs, r, world)[1] for r in range(world)]


class BoundaryGather:
    """Pre-allocated all-gather of the per-step outputs of every shard.

    Equal shards use ``all_gather_into_tensor`` straight into the global
    buffers (one collective per buffer, no staging). Ragged shards pad to the
    largest shard and trim on arrival.
    """

    def __init__(self, total_envs, obs_dim, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.total = int(total_envs)
        self.counts = shard_counts(self.total, self.world)
        self.equal = len(set(self.counts)) == 1
        self.max_count = max(self.counts)
        rows = self.total if self.equal else self.max_count * self.world
        self._coalesce = True  # falls back to four plain collectives the first time the grouped launch is refused
        self.obs = torch.empty((rows, obs_dim), dtype=torch.float32, device=device)
        self.rew = torch.empty(rows, dtype=torch.float32, device=device)
        self.term = torch.empty(rows, dtype=torch.uint8, device=device)
        self.trunc = torch.empty(rows, dtype=torch.uint8, device=device)
        if not self.equal:
            self._pad = {
                "obs": torch.zeros((self.max_count, obs_dim), dtype=torch.float32, device=device),
                "rew": torch.zeros(self.max_count, dtype=torch.float32, device=device),
                "term": torch.zeros(self.max_count, dtype=torch.uint8, device=device),
                "trunc": torch.zeros(self.max_count, dtype=torch.uint8, device=device),
            }
            # rows of the padded gather that are real envs, in global env order (built once, not per step)
            self._keep = torch.cat([torch.arange(r * self.max_count, r * self.max_count + c, device=device) for r, c in enumerate(self.counts)])

    def __call__(self, obs, rew, term, trunc):
        """Gather this rank's outputs; returns global (obs, rew, term, trunc) in env-id order."""
        srcs = {"obs": obs, "rew": rew, "term": term, "trunc": trunc}
        dsts = {"obs": self.obs, "rew": self.rew, "term": self.term, "trunc": self.trunc}
        if not self.equal:
            mine = self.counts[self.rank]
            for k in srcs:
                self._pad[k][:mine].copy_(srcs[k])
            srcs = self._pad
        else:
            srcs = {k: t.contiguous() for k, t in srcs.items()}
        if self._coalesce:
            try:
                with _one_launch(self.group, self.obs.device):
                    for k in srcs:
                        dist.all_gather_into_tensor(dsts[k], srcs[k], group=self.group)
            except (RuntimeError, TypeError, ValueError):
                # raised at enqueue time on every rank alike (an unsupported grouping), before anything was launched
                self._coalesce = False
        if not self._coalesce:
            for k in srcs:
                dist.all_gather_into_tensor(dsts[k], srcs[k], group=self.group)
        if self.equal:
            return self.obs, self.rew, self.term, self.trunc
        keep = self._keep
        return self.obs[keep], self.rew[keep], self.term[keep], self.trunc[keep]


class RootGather:
    """Pre-allocated gather of the per-step outputs of every shard TO ONE RANK (the north-star's "RCCL gather").

    Point-to-point: every non-root rank posts one send per buffer, the root one receive per (rank, buffer) straight
    into that rank's rows of the global buffers — ragged shards need no padding — all batched into one launch
    (``batch_isend_irecv`` = ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on RCCL). Against the all-gather,
    the N-1 other GPUs stop receiving and writing (N-1)/N of the batch each; the root's inbound traffic is the same.
    ``__call__`` returns the global tensors on the root, ``None`` elsewhere.
    """

    def __init__(self, total_envs, obs_dim, device, root=0, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.root = int(root)
        if not 0 <= self.root < self.world:
            raise ValueError(f"root {root} outside the group of {self.world}")
        self.total = int(total_envs)
        self.counts = shard_counts(self.total, self.world)
        self.offsets = [shard_range(self.total, r, self.world)[0] for r in range(self.world)]
        self._peer = [r if group is None else dist.get_global_rank(group, r) for r in range(self.world)]
        self.obs = self.rew = self.term = self.trunc = None
        if self.rank == self.root:
            self.obs = torch.empty((self.total, obs_dim), dtype=torch.float32, device=device)
            self.rew = torch.empty(self.total, dtype=torch.float32, device=device)
            self.term = torch.empty(self.total, dtype=torch.uint8, device=device)
            self.trunc = torch.empty(self.total, dtype=torch.uint8, device=device)

    def __call__(self, obs, rew, term, trunc):
        srcs = (obs, rew, term, trunc)
        if self.rank != self.root:
            ops = [dist.P2POp(dist.isend, t.contiguous(), self._peer[self.root], self.group) for t in srcs]
        else:
            dsts = (self.obs, self.rew, self.term, self.trunc)
            mine = slice(self.offsets[self.rank], self.offsets[self.rank] + self.counts[self.rank])
            for d, t in zip(dsts, srcs):
                if d[mine].data_ptr() != t.data_ptr():  # in place when the env already writes into its rows
                    d[mine].copy_(t)
            ops = [dist.P2POp(dist.irecv, d[self.offsets[r]:self.offsets[r] + self.counts[r]], self._peer[r], self.group)
                   for r in range(self.world) if r != self.rank for d in dsts]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return (self.obs, self.rew, self.term, self.trunc) if self.rank == self.root else None


class PipelinedGather:
    """Overlap the host-boundary all-gather of step k with the kernel of step k+1.

    The env alternates between two output sets (``bind_outputs``): while the
    collective reads set A on a side stream, the next step writes set B on the
    main stream. ``step()`` returns the gathered tensors of the step it just
    launched plus the event that marks them complete; the consumer waits on that
    event (``torch.cuda.current_stream().wait_event(ev)``) before reading.
    On CPU tensors (gloo) the same code runs synchronously.
    """

    def __init__(self, vec, total_envs, obs_dim, group=None, launch=None):
        self.vec = vec
        # what one "step" launches on the env: vec.step() by default; e.g. `lambda: vec.rollout(128)` pipelines the
        # horizon gather of one fused rollout behind the next rollout's kernel (the device-side policy needs no observations)
        self.launch = launch if launch is not None else vec.step
        first = (vec.observations, vec.rewards, vec.terminals, vec.truncations)
        second = tuple(torch.empty_like(t) for t in first)
        self.sets = [first, second]
        dev = first[0].device
        self.cuda = dev.type == "cuda"
        self.gathers = [BoundaryGather(total_envs, obs_dim, dev, group), BoundaryGather(total_envs, obs_dim, dev, group)]
        self.k = 0
        if self.cuda:
            self.comm = torch.cuda.Stream(device=dev)
            self.done = [torch.cuda.Event(), torch.cuda.Event()]
            self.ready = [torch.cuda.Event(), torch.cuda.Event()]  # reused: no per-step event construction
            for ev in self.done:
                ev.record(torch.cuda.current_stream(dev))

    def step(self):
        i = self.k & 1
        self.k += 1
        outs = self.sets[i]
        if not self.cuda:
            self.vec.bind_outputs(*outs)
            self.launch()
            return self.gathers[i](*outs), None
        main = torch.cuda.current_stream(outs[0].device)
        if hasattr(self.vec, "use_torch_stream"):
            self.vec.use_torch_stream()  # launch on the stream the events below are recorded on (a no-op while it is unchanged)
        main.wait_event(self.done[i])  # the gather that last read this set (two steps ago) has finished
        self.vec.bind_outputs(*outs)
        self.launch()
        ready = self.ready[i]
        ready.record(main)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ready)
            gathered = self.gathers[i](*outs)
            self.done[i].record(self.comm)
        return gathered, self.done[i]


class PeerStoreGather:
    """The host-boundary exchange WITHOUT a collective, for torch consumers (``include/drone_vec.h``:
    ``drone_vec_gather_peer_export`` / ``drone_vec_gather_init_peer``).

    The root allocates the global batch in its HBM and exports it (IPC handles); the token and the name of a 4 KiB flag
    page in ``/dev/shm`` travel over ``torch.distributed`` once, here in the constructor (any backend: it is the only use
    of it); every rank then binds its env's OUTPUT pointers to its rows of the root's batch. From then on each rank's step
    kernel stores its observations / rewards / flags into the root's memory itself — xGMI stores on the other GPUs of a
    node — and ``__call__`` (once per launch on every rank, like the collectives) is only the handshake: the other ranks
    publish "landed", the root's stream waits for all of them. Returns the global tensors on the root, ``None`` elsewhere.
    The env's own ``observations`` / ``rewards`` / ... tensors are not written while this is active; ``close()`` hands
    the env its buffers back. Device-buffer envs only (``binding.DroneVec`` with ``device=``)."""

    def __init__(self, vec, total_envs, root=0, group=None, counts=None):
        import os
        import secrets

        import numpy as np

        from . import abi

        self.vec, self.group = vec, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.root = int(root)
        self.total = int(total_envs)
        self.counts = list(counts) if counts is not None else shard_counts(self.total, self.world)
        dev, od = vec.torch_device, abi.obs_dim(vec.cfg.task)
        self.obs = self.rew = self.term = self.trunc = None
        box = [None, None]
        self._flagfile = None
        src = self.root if group is None else dist.get_global_rank(group, self.root)
        try:
            if self.rank == self.root:
                self.obs = torch.zeros((self.total, od), dtype=torch.float32, device=dev)
                self.rew = torch.zeros(self.total, dtype=torch.float32, device=dev)
                self.term = torch.zeros(self.total, dtype=torch.uint8, device=dev)
                self.trunc = torch.zeros(self.total, dtype=torch.uint8, device=dev)
                # A job that is SIGKILLed between creating the page and unlinking it (below: once every rank has mapped it) leaves 4 KiB
                # behind under a name that carries its pid: sweep what dead processes of this user left, then make a fresh name,
                # created exclusively (never an existing file or a symlink someone planted under a guessable name), private to this user.
                # Only pages older than ten minutes go: with /dev/shm shared across PID namespaces (containers with host
                # IPC) a LIVE job's pid may not exist here, and its page lives for the seconds between its creation and the barrier below.
                import time

                for stale in os.listdir("/dev/shm"):
                    if stale.startswith("drone_peer_flags_"):
                        path = os.path.join("/dev/shm", stale)
                        try:
                            if time.time() - os.stat(path).st_mtime < 600.0:
                                continue
                            os.kill(int(stale.split("_")[3]), 0)
                        except ProcessLookupError:
                            with contextlib.suppress(OSError):
                                os.unlink(path)
                        except (ValueError, IndexError, PermissionError, OSError):
                            pass
                name = f"/dev/shm/drone_peer_flags_{os.getpid()}_{secrets.token_hex(8)}"
                fd = os.open(name, os.O_CREAT | os.O_EXCL | os.O_RDWR | getattr(os, "O_NOFOLLOW", 0), 0o600)
                self._flagfile = name
                try:
                    os.write(fd, b"\0" * 4096)
                finally:
                    os.close(fd)
                box = [vec.gather_peer_export(self.obs, self.rew, self.term, self.trunc), self._flagfile]
        finally:
            # every rank reaches the broadcast, also when the root failed above: the others then see [None, None] and raise instead of hanging
            dist.broadcast_object_list(box, src=src, group=group)
        failure = None
        try:
            if box[0] is None:
                raise RuntimeError("PeerStoreGather: the root could not export its batch")
            self._flags = np.memmap(box[1], dtype=np.uint32, mode="r+", shape=(1024,))  # one shared page, page-aligned
            vec.gather_init_peer(box[0], self._flags, self.rank, self.world, root=self.root, counts=self.counts)
        except Exception as exc:  # noqa: BLE001  (re-raised below, once every rank knows)
            failure = exc
        try:
            # Every rank reaches this exchange whether or not it could map the page (a rank that raised above used to skip
            # the barrier and leave the others in it until the process-group timeout); it doubles as the barrier — every rank has
            # mapped the page, its NAME can go now, the memory lives as long as the mappings.
            oks = [None] * self.world
            dist.all_gather_object(oks, failure is None, group=group)
        finally:
            self._unlink()
        if failure is not None or not all(oks):
            if failure is None:
                with contextlib.suppress(Exception):
                    vec.gather_close()
                failure = RuntimeError(f"PeerStoreGather: rank(s) {[r for r, ok in enumerate(oks) if not ok]} could not join the exchange")
            raise failure

    def _unlink(self):
        import os

        if self._flagfile:
            try:
                os.unlink(self._flagfile)
            except OSError:
                pass
            self._flagfile = None

    def __call__(self):
        self.vec.gather()
        return (self.obs, self.rew, self.term, self.trunc) if self.rank == self.root else None

    def close(self):
        self.vec.gather_close()
        self._unlink()  # (already gone: unlinked in the constructor once every rank had mapped it)
        self._flags = None

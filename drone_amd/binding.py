"""ctypes binding of ``libdrone_hip.so`` — the C-ABI of ``include/drone_vec.h``.

This is the host-side mirror of what PufferLib's env binding does for the
drone env's vec path (vec_init / vec_reset / vec_step / vec_log / vec_close
over shared buffers; SURVEY.md §3, §8b — the reference binding itself is not
in the snapshot, ``/root/reference/.gitmodules:1-3``).

The product path is HIP only: loading fails loudly if the shared object has
not been built, and ``DroneVec`` raises if ``drone_vec_init`` returns NULL
(no GPU, wrong arch). There is no CPU fallback anywhere in ``drone_amd``.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_DIR = os.path.dirname(os.path.abspath(__file__))
# DRONE_HIP_LIB: alternative build of the same library (tuning experiments only)
LIB_PATH = os.environ.get("DRONE_HIP_LIB") or os.path.join(_DIR, "libdrone_hip.so")
_lib = None
_fns = None


def load():
    """Load libdrone_hip.so (once). torch is imported first so that the
    library binds to the same HIP runtime instance torch uses."""
    global _lib, _fns
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C drone_amd/csrc)")
        try:
            import torch  # noqa: F401  (same libamdhip64 for tensors, streams and RCCL)
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _fns = abi.bind(_lib)
    return _lib


def last_error():
    load()
    return _fns["drone_last_error"]().decode()


def default_config(task=abi.TASK_HOVER, **overrides):
    load()
    cfg = abi.DroneConfig()
    _fns["drone_config_default"](C.byref(cfg), task)
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(f"DroneConfig has no field {k!r}")
        setattr(cfg, k, v)
    return cfg


def gather_unique_id():
    """128 opaque bytes naming a new RCCL communicator; one rank makes them, all ranks pass them to ``gather_init``."""
    load()
    buf = (C.c_ubyte * abi.GATHER_ID_BYTES)()
    if _fns["drone_gather_unique_id"](C.cast(buf, C.c_void_p)) != 0:
        raise RuntimeError("libdrone_hip: " + _fns["drone_last_error"]().decode())
    return bytes(buf)


def page_buffer(shape, dtype):
    """A zeroed numpy array that OWNS its pages: backed by an anonymous mapping of its own, so it starts on a page
    boundary and nothing else lives in its pages (the mapping is padded to whole pages and stays alive as long as the
    array does). Only such host buffers are pinned for the zero-copy transport — registering memory that shares a page
    with other heap allocations trips a ROCm runtime fault (include/drone_vec.h: DroneConfig.host_pages_exclusive)."""
    import mmap

    count = int(np.prod(shape))
    nbytes = max(1, count * np.dtype(dtype).itemsize)
    m = mmap.mmap(-1, (nbytes + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE)
    return np.frombuffer(m, dtype=dtype, count=count).reshape(shape)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x):
    return x.data_ptr() if _is_torch(x) else x.ctypes.data


def load_variant(path):
    """Bind another build of the library (tuning A/B only): returns its function table."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    return abi.bind(C.CDLL(path))


class StepManyBuffers:
    """The K-major blocks of one ``step_many`` call: actions [K][N][4] in, observations [K][N][O] / rewards [K][N] /
    terminals [K][N] / truncations [K][N] out (numpy for host-buffer handles, torch tensors in HBM otherwise)."""

    def __init__(self, k_steps, actions, observations, rewards, terminals, truncations):
        self.k_steps = int(k_steps)
        self.actions, self.observations, self.rewards, self.terminals, self.truncations = actions, observations, rewards, terminals, truncations


class DroneVec:
    """One shard of envs on one GPU.

    ``device=None``: numpy host buffers (each step copies actions in and
    observations / rewards / flags out). ``device="cuda:0"`` (or an index):
    torch tensors in HBM, zero-copy, launches asynchronous on torch's current
    stream for that device.
    """

    def __init__(self, num_envs, seed=0, task=abi.TASK_HOVER, device=None, cfg=None, fns=None, buffers=None, **overrides):
        load()
        self._f = _fns if fns is None else fns
        self.num_envs = int(num_envs)
        n = self.num_envs
        # a private copy: the handle adjusts buffer_kind / device / host_pages_exclusive below, and a caller's cfg object
        # reused for a second handle must not inherit them (page ownership vouched for by accident)
        self.cfg = abi.DroneConfig.from_buffer_copy(cfg) if cfg is not None else default_config(task, **overrides)
        self._h = None
        if buffers is not None:
            # caller-owned buffers (the PufferLib contract: the vec-env allocates them, possibly as
            # slices of one shared-memory block, and every env writes into its slice)
            obs, act, rew, term, trunc = buffers
            self._check_buffers(obs, act, rew, term, trunc, n, abi.obs_dim(self.cfg.task))
            self.observations, self.actions, self.rewards, self.terminals, self.truncations = obs, act, rew, term, trunc
            if _is_torch(obs):
                self.torch_device = obs.device
                self.cfg.buffer_kind = abi.BUFFERS_DEVICE
                self.cfg.device = obs.device.index if obs.device.index is not None else 0
            else:
                self.torch_device = None
                self.cfg.buffer_kind = abi.BUFFERS_HOST
        elif device is None:
            self.torch_device = None
            self.cfg.buffer_kind = abi.BUFFERS_HOST
            self.cfg.host_pages_exclusive = 1  # page_buffer: each array is a mapping of its own
            self.observations = page_buffer((n, abi.obs_dim(self.cfg.task)), np.float32)
            self.actions = page_buffer((n, abi.ACT_DIM), np.float32)
            self.rewards = page_buffer((n,), np.float32)
            self.terminals = page_buffer((n,), np.uint8)
            self.truncations = page_buffer((n,), np.uint8)
        else:
            import torch

            dev = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
            if dev.type != "cuda":
                raise ValueError("device buffers must live on a GPU")
            self.torch_device = dev
            self.cfg.buffer_kind = abi.BUFFERS_DEVICE
            self.cfg.device = dev.index if dev.index is not None else torch.cuda.current_device()
            self.observations = torch.zeros((n, abi.obs_dim(self.cfg.task)), dtype=torch.float32, device=dev)
            self.actions = torch.zeros((n, abi.ACT_DIM), dtype=torch.float32, device=dev)
            self.rewards = torch.zeros(n, dtype=torch.float32, device=dev)
            self.terminals = torch.zeros(n, dtype=torch.uint8, device=dev)
            self.truncations = torch.zeros(n, dtype=torch.uint8, device=dev)
        self._seed = int(seed)
        self._pinned_blocks = []
        self._gathered = None      # global buffers of an active exchange (kept alive here)
        self._peer_flags = None
        self._h = self._f["drone_vec_init"](
            _ptr(self.observations), _ptr(self.actions), _ptr(self.rewards), _ptr(self.terminals), _ptr(self.truncations),
            n, seed, C.byref(self.cfg))
        if not self._h:
            raise RuntimeError("drone_vec_init failed: " + self._f["drone_last_error"]().decode())
        self._step = self._f["drone_vec_step"]
        self._status = self._f["drone_vec_status"]
        if self.torch_device is not None:
            self.use_torch_stream()

    @staticmethod
    def _check_buffers(obs, act, rew, term, trunc, n, obs_dim):
        want = ((obs, (n, obs_dim), "float32"), (act, (n, abi.ACT_DIM), "float32"), (rew, (n,), "float32"),
                (term, (n,), "uint8"), (trunc, (n,), "uint8"))
        kinds = {_is_torch(b) for b, _, _ in want}
        if len(kinds) != 1:
            raise TypeError("buffers must be all numpy arrays or all torch tensors")
        for b, shape, dt in want:
            if tuple(b.shape) != shape:
                raise ValueError(f"buffer shape {tuple(b.shape)} != {shape}")
            if str(b.dtype).replace("torch.", "") not in (dt, "bool" if dt == "uint8" else dt):
                raise TypeError(f"buffer dtype {b.dtype} != {dt}")
            contiguous = b.is_contiguous() if _is_torch(b) else b.flags["C_CONTIGUOUS"]
            if not contiguous:
                raise ValueError("buffers must be C-contiguous")
            if _is_torch(b) and b.device.type != "cuda":
                raise ValueError("torch buffers must live on a GPU (pass numpy arrays for host buffers)")

    # -- stream plumbing --
    def use_torch_stream(self):
        import torch

        s = torch.cuda.current_stream(self.torch_device)
        self._check(self._f["drone_vec_set_stream"](self._h, C.c_void_p(s.cuda_stream)))

    def sync(self):
        self._check(self._f["drone_vec_sync"](self._h))

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("libdrone_hip: " + self._f["drone_last_error"]().decode())

    def _raise_if_failed(self):
        """reset / step / rollout / log return void in the C-ABI (PufferLib's
        convention); a failed launch or copy sticks to the handle instead."""
        if self._status(self._h):
            msg = self._f["drone_vec_status_message"](self._h).decode()
            raise RuntimeError("libdrone_hip: " + msg)

    # -- the path --
    def reset(self, seed=0):
        self._seed = int(seed)
        self._f["drone_vec_reset"](self._h, seed)
        self._raise_if_failed()

    def step(self):
        self._step(self._h)
        if self._status(self._h):
            self._raise_if_failed()

    def step_send(self):
        """First half of ``step``: read the actions, enqueue the step; returns without waiting (``drone_vec_step_send``)."""
        self._f["drone_vec_step_send"](self._h)
        if self._status(self._h):
            self._raise_if_failed()

    def step_recv(self):
        """Second half: wait for the sent step and deliver its outputs into the buffers (``drone_vec_step_recv``)."""
        self._f["drone_vec_step_recv"](self._h)
        if self._status(self._h):
            self._raise_if_failed()

    def rollout(self, horizon):
        self._f["drone_vec_rollout"](self._h, int(horizon))
        self._raise_if_failed()

    def alloc_step_many(self, k_steps, pinned=True):
        """Blocks of the handle's buffer kind for ``step_many`` (allocate once, reuse every call). Host handles:
        page-owning blocks, pinned (``drone_vec_host_pin``) unless ``pinned=False`` — the kernel then reads / writes
        them in place over PCIe; unpinned ones go through device staging and copies."""
        n, od, K = self.num_envs, abi.obs_dim(self.cfg.task), int(k_steps)
        if self.torch_device is None:
            blocks = (page_buffer((K, n, abi.ACT_DIM), np.float32), page_buffer((K, n, od), np.float32), page_buffer((K, n), np.float32),
                      page_buffer((K, n), np.uint8), page_buffer((K, n), np.uint8))
            if pinned:
                for b in blocks:
                    self._check(self._f["drone_vec_host_pin"](self._h, b.ctypes.data, b.nbytes, 1))
                    self._pinned_blocks.append(b)  # unpinned by free_step_many or at close, before the arrays can go away
            return StepManyBuffers(K, *blocks)
        import torch

        dev = self.torch_device
        return StepManyBuffers(K, torch.zeros((K, n, abi.ACT_DIM), dtype=torch.float32, device=dev), torch.zeros((K, n, od), dtype=torch.float32, device=dev),
                               torch.zeros((K, n), dtype=torch.float32, device=dev), torch.zeros((K, n), dtype=torch.uint8, device=dev),
                               torch.zeros((K, n), dtype=torch.uint8, device=dev))

    def free_step_many(self, bufs):
        """Unpin and forget the host blocks of an ``alloc_step_many`` result (a caller that changes K allocates new
        blocks; without this the old pinned K x N blocks stay registered and referenced until ``close``). Device blocks
        are plain tensors: nothing to do."""
        if self.torch_device is not None or not getattr(self, "_h", None):
            return
        for b in (bufs.actions, bufs.observations, bufs.rewards, bufs.terminals, bufs.truncations):
            for k, held in enumerate(self._pinned_blocks):
                if held is b:
                    self._check(self._f["drone_vec_host_unpin"](self._h, b.ctypes.data))
                    del self._pinned_blocks[k]
                    break

    def step_many(self, bufs, policy=False):
        """K env steps in one launch with every step's outputs (``drone_vec_step_many``): reads ``bufs.actions`` (or,
        with ``policy=True``, draws the SPEC.md random policy in the kernel) and fills the four output blocks."""
        n, od, K = self.num_envs, abi.obs_dim(self.cfg.task), bufs.k_steps
        for t, shape in ((bufs.actions, (K, n, abi.ACT_DIM)), (bufs.observations, (K, n, od)), (bufs.rewards, (K, n)), (bufs.terminals, (K, n)), (bufs.truncations, (K, n))):
            if tuple(t.shape) != shape or _is_torch(t) != (self.torch_device is not None):
                raise ValueError(f"step_many block of shape {tuple(t.shape)}, expected {shape} of the handle's buffer kind")
        self._f["drone_vec_step_many"](self._h, K, None if policy else _ptr(bufs.actions), _ptr(bufs.observations), _ptr(bufs.rewards),
                                       _ptr(bufs.terminals), _ptr(bufs.truncations))
        if self._status(self._h):
            self._raise_if_failed()
        return bufs

    def step_repeat(self, bufs, actions=None):
        """Action repeat / frame skip: ``bufs.k_steps`` env steps in one launch under ONE action block ``[N][4]``
        (default: the handle's bound ``actions``), every step's outputs in ``bufs`` (``drone_vec_step_repeat``)."""
        actions = self.actions if actions is None else actions
        if tuple(actions.shape) != (self.num_envs, abi.ACT_DIM) or _is_torch(actions) != (self.torch_device is not None):
            raise ValueError("step_repeat: actions must be [num_envs][4] of the handle's buffer kind")
        self._f["drone_vec_step_repeat"](self._h, bufs.k_steps, _ptr(actions), _ptr(bufs.observations), _ptr(bufs.rewards),
                                         _ptr(bufs.terminals), _ptr(bufs.truncations))
        if self._status(self._h):
            self._raise_if_failed()
        return bufs

    def done_list_at(self, k):
        """ids of the envs that finished in step ``k`` of the last ``step_many`` (compact_done=1)."""
        ids = np.zeros(self.num_envs, dtype=np.uint32)
        cnt = self._f["drone_vec_done_list_at"](self._h, int(k), ids.ctypes.data, self.num_envs)
        if cnt < 0:
            raise RuntimeError("libdrone_hip: " + self._f["drone_last_error"]().decode())
        return ids[:cnt]

    def log(self):
        out = abi.DroneLog()
        self._f["drone_vec_log"](self._h, C.byref(out))
        self._raise_if_failed()
        return out.as_dict()

    def close(self):
        if getattr(self, "_h", None):
            for b in getattr(self, "_pinned_blocks", []):
                self._f["drone_vec_host_unpin"](self._h, b.ctypes.data)
            self._pinned_blocks = []
            self._f["drone_vec_close"](self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- around the path --
    @property
    def gstep(self):
        return self._f["drone_vec_gstep"](self._h)

    @property
    def bytes_per_env_step(self):
        """Algorithmic HBM bytes per env-step of the per-step kernel for this handle (task + state layout)."""
        return self._f["drone_vec_bytes_per_env_step"](self._h)

    @property
    def variant(self):
        """Which per-step kernel instantiation and launch choices this handle uses (``drone_vec_variant``), as text and
        parsed: {'task': 0, 'compact': 0, 'mem': 0, 'dt': 1, 'order': 1, 'line_complete': 0, 'packed_rk4': 0, 'bytes': 262}."""
        import re

        text = self._f["drone_vec_variant"](self._h).decode()
        return text, {k: int(v) for k, v in re.findall(r"(\w+)=(\d+)", text)}

    @property
    def host_transport(self):
        """'zero-copy' (the kernel reads / writes the pinned host buffers over PCIe), 'stand-in' (the same through
        pinned stand-ins the library owns for buffers it may not pin, copied on the host around each step), 'stand-in-mt'
        (mid-size shards: the stand-ins moved by the library's host thread pool, the outputs while the kernel still runs),
        'mirror' (device mirrors + DMA copies), or None for device buffers."""
        t = self._f["drone_vec_host_transport"](self._h)
        return {1: "zero-copy", 2: "stand-in", 3: "stand-in-mt", 0: "mirror"}.get(t)

    def enable_graph_capture(self, on=True):
        """Counters in HBM, advanced by the kernels: a captured step / rollout (torch.cuda.graph) replays correctly."""
        self._check(self._f["drone_vec_enable_graph_capture"](self._h, 1 if on else 0))

    def set_gstep(self, gstep):
        self._check(self._f["drone_vec_set_gstep"](self._h, int(gstep)))

    def status(self):
        return self._f["drone_vec_status"](self._h), self._f["drone_vec_status_message"](self._h).decode()

    def clear_status(self):
        self._f["drone_vec_clear_status"](self._h)

    # -- host-boundary all-gather through the C-ABI (RCCL called from the library, not torch.distributed) --
    def gather_init(self, unique_id, rank, world, all_observations, all_rewards, all_terminals, all_truncations, counts=None, root=-1):
        """``unique_id``: the DRONE_GATHER_ID_BYTES bytes rank 0 got from ``gather_unique_id()``. ``root`` = -1: all-gather
        (every rank receives the batch); ``root`` >= 0: gather to that rank only — the other ranks pass ``None`` buffers."""
        idbuf = (C.c_ubyte * abi.GATHER_ID_BYTES).from_buffer_copy(bytes(unique_id))
        cnt = None
        if counts is not None:
            cnt = (C.c_int * world)(*[int(c) for c in counts])
        self._gathered = (all_observations, all_rewards, all_terminals, all_truncations)  # keep alive
        ptrs = [_ptr(b) if b is not None else None for b in self._gathered]
        self._check(self._f["drone_vec_gather_init_root"](self._h, C.cast(idbuf, C.c_void_p), int(rank), int(world),
                                                           C.cast(cnt, C.c_void_p) if cnt is not None else None, int(root), *ptrs))

    def gather(self):
        self._check(self._f["drone_vec_gather"](self._h))
        return self._gathered

    # -- the same exchange without a collective: peer stores (drone_vec_gather_peer_export / _init_peer) --
    def gather_peer_export(self, all_observations, all_rewards, all_terminals, all_truncations):
        """Root only: IPC handles of its four global device buffers as ``abi.PEER_TOKEN_BYTES`` opaque bytes; ship them to
        the other ranks (any means), then every rank calls ``gather_init_peer``."""
        tok = (C.c_ubyte * abi.PEER_TOKEN_BYTES)()
        self._gathered = (all_observations, all_rewards, all_terminals, all_truncations)  # keep alive
        self._check(self._f["drone_vec_gather_peer_export"](self._h, *[_ptr(b) for b in self._gathered], C.cast(tok, C.c_void_p)))
        return bytes(tok)

    def gather_init_peer(self, token, shared_flags, rank, world, root=0, counts=None):
        """Every rank: from now on this handle's kernels write its rows of the ROOT's global buffers (xGMI stores on the
        other GPUs), and ``gather()`` is only the handshake. ``shared_flags``: a page-aligned 4 KiB numpy uint32 array
        (or anything with ``ctypes.data``) backed by memory all ranks share, zeroed by its creator. The handle's
        ``observations`` / ``rewards`` / ``terminals`` / ``truncations`` attributes no longer name what the kernels write
        until ``gather_close``."""
        tokbuf = (C.c_ubyte * abi.PEER_TOKEN_BYTES).from_buffer_copy(bytes(token))
        cnt = (C.c_int * world)(*[int(c) for c in counts]) if counts is not None else None
        self._peer_flags = shared_flags  # keep the mapping alive
        self._check(self._f["drone_vec_gather_init_peer"](self._h, C.cast(tokbuf, C.c_void_p), _ptr(shared_flags), int(rank), int(world),
                                                           C.cast(cnt, C.c_void_p) if cnt is not None else None, int(root)))

    def gather_close(self):
        self._f["drone_vec_gather_close"](self._h)
        self._gathered = None
        self._peer_flags = None

    def bind_actions(self, actions):
        self._check(self._f["drone_vec_bind_actions"](self._h, _ptr(actions)))
        self.actions = actions

    def bind_outputs(self, observations, rewards, terminals, truncations):
        self._check(self._f["drone_vec_bind_outputs"](self._h, _ptr(observations), _ptr(rewards), _ptr(terminals), _ptr(truncations)))
        self.observations, self.rewards, self.terminals, self.truncations = observations, rewards, terminals, truncations

    def fill_random_actions(self, gstep=None, out=None):
        out = self.actions if out is None else out
        g = self.gstep if gstep is None else gstep
        self._check(self._f["drone_vec_fill_random_actions"](self._h, _ptr(out), g))
        return out

    def get_state(self, first=0, count=None):
        count = self.num_envs - first if count is None else count
        rows = np.zeros(count, dtype=abi.state_row_dtype())
        self._check(self._f["drone_vec_get_state"](self._h, rows.ctypes.data, first, count))
        return rows

    def set_state(self, rows, first=0):
        rows = np.ascontiguousarray(rows, dtype=abi.state_row_dtype())
        self._check(self._f["drone_vec_set_state"](self._h, rows.ctypes.data, first, len(rows)))

    # -- checkpoint / resume: everything a run needs to continue bit for bit --
    # where the buffers live, and how the device lays the state out, are the resuming process's business (rows are layout-free)
    _CKPT_CFG_SKIP = ("struct_size", "buffer_kind", "device", "host_pages_exclusive", "state_layout")
    # What the persisted rows MEAN is the spec's business: SPEC v5 (round 4) turned the hover / swarm per-env log sums from
    # ratios into counts and changed the reset attitude, so a file written under v4 would load without complaint, mix the
    # two kinds of sums and no longer continue bit for bit. Bump together with SPEC.md's version.
    SPEC_VERSION = 5

    def save_checkpoint(self, path):
        """Write the shard's state rows, the vec-level step counter, the seed the RNG streams are keyed on, the env
        config and the current contents of the five caller buffers to ``path`` (numpy .npz). A handle restored with
        ``load_checkpoint`` continues exactly like this one: same trajectories, resets, wind, policy draws and logs."""
        self.sync()
        cfg = {k: v for k, v in self.cfg.as_dict().items() if k not in self._CKPT_CFG_SKIP}
        to_np = lambda x: x.cpu().numpy() if _is_torch(x) else np.asarray(x)  # noqa: E731
        with open(path, "wb") as fh:
            np.savez(fh, spec_version=np.int64(self.SPEC_VERSION), rows=self.get_state(), gstep=np.uint32(self.gstep), seed=np.uint64(self._seed), num_envs=np.int64(self.num_envs),
                     cfg_keys=np.array(list(cfg.keys())), cfg_vals=np.array([float(v) for v in cfg.values()], dtype=np.float64),
                     observations=to_np(self.observations), actions=to_np(self.actions), rewards=to_np(self.rewards),
                     terminals=to_np(self.terminals), truncations=to_np(self.truncations))

    def load_checkpoint(self, path):
        """Restore a ``save_checkpoint`` file into this handle (same num_envs and env config; host or device buffers
        alike). Refuses a file written under a different config, shard size or SPEC version rather than continuing a different env."""
        with np.load(path) as z:
            ver = int(z["spec_version"]) if "spec_version" in z.files else 4  # files older than the field were written under SPEC v4 or earlier
            if ver != self.SPEC_VERSION:
                raise ValueError(f"checkpoint was written under SPEC v{ver}{' or earlier' if 'spec_version' not in z.files else ''}, this library implements "
                                 f"SPEC v{self.SPEC_VERSION}: its log sums and reset draws mean something else (SPEC.md sections 6 and 8); it cannot continue bit for bit")
            if int(z["num_envs"]) != self.num_envs:
                raise ValueError(f"checkpoint holds {int(z['num_envs'])} envs, this handle {self.num_envs}")
            mine = self.cfg.as_dict()
            for k, val in zip(z["cfg_keys"].tolist(), z["cfg_vals"].tolist()):
                if float(mine[k]) != val:
                    raise ValueError(f"checkpoint was written with {k} = {val}, this handle has {mine[k]}")
            self.reset(int(z["seed"]))  # re-keys every RNG stream, clears the log planes
            self.set_state(z["rows"])
            self.set_gstep(int(z["gstep"]))
            for name in ("observations", "actions", "rewards", "terminals", "truncations"):
                dst = getattr(self, name)
                if _is_torch(dst):
                    import torch

                    dst.copy_(torch.from_numpy(z[name]).view(dst.dtype) if dst.dtype == torch.bool else torch.from_numpy(z[name]))
                else:
                    dst[...] = z[name]
        self.sync()

    def done_list(self):
        ids = np.zeros(self.num_envs, dtype=np.uint32)
        cnt = self._f["drone_vec_done_list"](self._h, ids.ctypes.data, self.num_envs)
        if cnt < 0:
            raise RuntimeError("libdrone_hip: " + self._f["drone_last_error"]().decode())
        return ids[:cnt]

    def timer_start(self):
        self._check(self._f["drone_vec_timer_start"](self._h))

    def timer_stop(self):
        ms = C.c_float(0)
        self._check(self._f["drone_vec_timer_stop"](self._h, C.byref(ms)))
        return ms.value

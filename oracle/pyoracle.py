"""ctypes loader for the CPU oracle. TEST INFRASTRUCTURE ONLY.

Importable from tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg — never from ``drone_amd``. PARITY UNPINNED: the oracle
restates this repo's SPEC.md, not a reference file (the reference snapshot has
no simulator source; /root/reference/.gitmodules:1-3).

The shared object is (re)built with the host's own gcc on first use, so a
``-march=native`` binary built in one container is never run on another CPU.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

from drone_amd import abi

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libdrone_oracle.so")
_STAMP = os.path.join(_DIR, ".oracle_build_stamp")
_lib = None
_fns = None
# C-ABI symbols the oracle library does not mirror (plumbing around the path, the exchange, the K-step form)
_NOT_IN_ORACLE = (
    "drone_vec_set_stream", "drone_vec_sync", "drone_vec_bind_actions", "drone_vec_bind_outputs", "drone_vec_done_list",
    "drone_vec_timer_start", "drone_vec_timer_stop", "drone_last_error", "drone_device_count", "drone_vec_set_gstep", "drone_vec_enable_graph_capture", "drone_vec_status",
    "drone_vec_status_message", "drone_vec_clear_status", "drone_gather_unique_id", "drone_vec_gather_init",
    "drone_vec_gather", "drone_vec_gather_close", "drone_vec_step_many", "drone_vec_step_repeat", "drone_vec_done_list_at", "drone_vec_gather_init_root", "drone_vec_host_transport", "drone_vec_bytes_per_env_step", "drone_vec_buffers", "drone_vec_device", "drone_vec_step_send", "drone_vec_step_recv", "drone_vec_host_pin", "drone_vec_host_unpin", "drone_vec_variant", "drone_vec_gather_peer_export", "drone_vec_gather_init_peer", "drone_device_malloc", "drone_device_free", "drone_vec_copy_to_host")


def _host_signature():
    h = hashlib.sha1()
    for f in ("drone_oracle.h", "drone_oracle_vec.c", "Makefile", os.path.join("..", "include", "drone_vec.h")):
        with open(os.path.join(_DIR, f), "rb") as fh:
            h.update(fh.read())
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("flags"):
                    h.update(line.encode())
                    break
    except OSError:
        pass
    return h.hexdigest()


def build(force=False):
    sig = _host_signature()
    stamp = None
    if os.path.exists(_STAMP):
        with open(_STAMP) as fh:
            stamp = fh.read().strip()
    if force or not os.path.exists(_SO) or stamp != sig:
        subprocess.run(["make", "-C", _DIR, "-B", "libdrone_oracle.so"], check=True, capture_output=True)
        with open(_STAMP, "w") as fh:
            fh.write(sig)
    return _SO


def lib():
    global _lib, _fns
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _fns = _bind(_lib)
        _lib.oracle_hash32.restype = C.c_uint32
        _lib.oracle_hash32.argtypes = [C.c_uint32]
        _lib.oracle_stream_key.restype = C.c_uint32
        _lib.oracle_stream_key.argtypes = [C.c_uint64, C.c_uint32]
        _lib.oracle_rng_draw.restype = C.c_uint32
        _lib.oracle_rng_draw.argtypes = [C.c_uint32] * 4
        _lib.oracle_params_derive.argtypes = [C.POINTER(abi.DroneConfig), C.c_void_p]
        _lib.oracle_omp_max_threads.restype = C.c_int
    return _lib


def _bind(cdll):
    names = [n for n in abi.SYMBOLS if n not in _NOT_IN_ORACLE]
    fns = abi.bind(cdll, prefix_to="oracle_", names=names)
    cdll.oracle_set_threads.argtypes = [C.c_void_p, C.c_int]
    return fns


def variant(cc, flags, out):
    """Another build of the same oracle sources (tests: the golden vectors must come out bit for bit from every
    compiler and optimisation level that honours the numerics contract, i.e. -ffp-contract=off and no fast-math).
    Returns what ``OracleVec(..., fns=...)`` takes."""
    src = os.path.join(_DIR, "drone_oracle_vec.c")
    subprocess.run([cc, *flags, "-fPIC", "-shared", "-std=gnu11", "-o", out, src, "-lm"], check=True, capture_output=True)
    cdll = C.CDLL(out)
    return _bind(cdll), cdll


def default_config(task=0, **overrides):
    lib()
    cfg = abi.DroneConfig()
    _fns["drone_config_default"](C.byref(cfg), task)
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def params(cfg):
    out = np.zeros(32, dtype=np.float32)
    lib().oracle_params_derive(C.byref(cfg), out.ctypes.data)
    return out


class OracleVec:
    """Scalar C env looped over ``num_envs`` with numpy-owned buffers."""

    def __init__(self, num_envs, seed=0, cfg=None, task=0, threads=1, fns=None, **overrides):
        lib()
        self._f, self._l = (_fns, _lib) if fns is None else fns  # fns: another build of the oracle (`variant`)
        self.cfg = cfg if cfg is not None else default_config(task, **overrides)
        self.num_envs = int(num_envs)
        n = self.num_envs
        self.observations = np.zeros((n, abi.obs_dim(self.cfg.task)), dtype=np.float32)
        self.actions = np.zeros((n, abi.ACT_DIM), dtype=np.float32)
        self.rewards = np.zeros(n, dtype=np.float32)
        self.terminals = np.zeros(n, dtype=np.uint8)
        self.truncations = np.zeros(n, dtype=np.uint8)
        self._h = self._f["drone_vec_init"](
            self.observations.ctypes.data, self.actions.ctypes.data, self.rewards.ctypes.data,
            self.terminals.ctypes.data, self.truncations.ctypes.data, n, seed, C.byref(self.cfg))
        if not self._h:
            raise RuntimeError("oracle_vec_init failed")
        self._l.oracle_set_threads(self._h, threads)

    def reset(self, seed=0):
        self._f["drone_vec_reset"](self._h, seed)

    def step(self):
        self._f["drone_vec_step"](self._h)

    def rollout(self, horizon):
        self._f["drone_vec_rollout"](self._h, horizon)

    def step_many(self, k_steps, actions=None):
        """What the product's drone_vec_step_many must reproduce: K plain c_step passes. ``actions`` [K][N][4], or None
        for the random policy. Returns K-major (observations, rewards, terminals, truncations, done_ids per step)."""
        n, od = self.num_envs, self.observations.shape[1]
        obs = np.zeros((k_steps, n, od), np.float32)
        rew = np.zeros((k_steps, n), np.float32)
        term = np.zeros((k_steps, n), np.uint8)
        trunc = np.zeros((k_steps, n), np.uint8)
        done = []
        for k in range(k_steps):
            if actions is None:
                self.fill_random_actions()
            else:
                self.actions[:] = actions[k]
            self.step()
            obs[k], rew[k], term[k], trunc[k] = self.observations, self.rewards, self.terminals, self.truncations
            done.append(np.flatnonzero(self.terminals | self.truncations).astype(np.uint32))
        return obs, rew, term, trunc, done

    def fill_random_actions(self, gstep=None, out=None):
        out = self.actions if out is None else out
        g = self.gstep if gstep is None else gstep
        self._f["drone_vec_fill_random_actions"](self._h, out.ctypes.data, g)
        return out

    @property
    def gstep(self):
        return self._f["drone_vec_gstep"](self._h)

    def log(self):
        out = abi.DroneLog()
        self._f["drone_vec_log"](self._h, C.byref(out))
        return out.as_dict()

    def get_state(self, first=0, count=None):
        count = self.num_envs - first if count is None else count
        rows = np.zeros(count, dtype=abi.state_row_dtype())
        rc = self._f["drone_vec_get_state"](self._h, rows.ctypes.data, first, count)
        if rc != 0:
            raise RuntimeError("oracle get_state failed")
        return rows

    def set_state(self, rows, first=0):
        rows = np.ascontiguousarray(rows, dtype=abi.state_row_dtype())
        rc = self._f["drone_vec_set_state"](self._h, rows.ctypes.data, first, len(rows))
        if rc != 0:
            raise RuntimeError("oracle set_state failed")

    def close(self):
        if self._h:
            self._f["drone_vec_close"](self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

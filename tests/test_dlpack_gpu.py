"""DLPack at the boundary ("hand obs/reward/done to a torch-ROCm policy as device tensors (DLPack)").

Export: an env created with five None buffers owns its HBM buffers (drone_vec_init with NULL pointers,
drone_vec_buffers) and vec_dlpack hands them out as DLPack capsules that torch.from_dlpack wraps zero-copy.
Import: any DLPack producer on a ROCm device can be the env's buffers — here torch tensors hidden behind a wrapper
that offers nothing but __dlpack__ (what a cupy / jax array would look like to the binding).
Both are driven against the oracle, bit for bit."""
import gc
import os
import subprocess

import numpy as np
import pytest

from drone_amd import abi
from helpers import assert_bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ext(hip):
    subprocess.run(["make", "-C", os.path.join(ROOT, "bindings")], check=True, capture_output=True)
    import torch  # noqa: F401

    from drone_amd import drone_binding

    return drone_binding


class OnlyDLPack:
    """A DLPack producer and nothing else: no data_ptr(), no buffer protocol."""

    def __init__(self, t):
        self._t = t

    def __dlpack__(self, stream=None):
        return self._t.__dlpack__()

    def __dlpack_device__(self):
        return self._t.__dlpack_device__()


def views(ext, h):
    import torch

    return [torch.from_dlpack(ext.vec_dlpack(h, name)) for name in ("observations", "actions", "rewards", "terminals", "truncations")]


def drive(ext, oracle, h, o, obs, act, rew, term, trunc, steps, seed):
    import torch

    ext.vec_reset(h, seed)
    o.reset(seed)
    torch.cuda.synchronize()
    assert_bits_equal(o.observations, obs.cpu().numpy(), "reset obs")
    for t in range(steps):
        o.fill_random_actions()
        act.copy_(torch.from_numpy(o.actions))
        torch.cuda.synchronize()
        o.step()
        ext.vec_step(h)
        torch.cuda.synchronize()
        assert_bits_equal(o.observations, obs.cpu().numpy(), f"obs {t}")
        assert_bits_equal(o.rewards, rew.cpu().numpy(), f"rew {t}")
        assert_bits_equal(o.terminals, term.cpu().numpy(), f"term {t}")
        assert_bits_equal(o.truncations, trunc.cpu().numpy(), f"trunc {t}")


@pytest.mark.gpu
@pytest.mark.parametrize("task", [0, 1, 2, 3])
def test_library_owned_buffers_exported_as_dlpack(ext, oracle, task):
    import torch

    n, seed = 4096, 23
    h = ext.vec_init(None, None, None, None, None, n, seed, task=task, horizon=40)
    obs, act, rew, term, trunc = views(ext, h)
    assert obs.shape == (n, abi.obs_dim(task)) and obs.dtype == torch.float32 and obs.is_cuda and obs.is_contiguous()
    assert act.shape == (n, 4) and act.dtype == torch.float32
    assert rew.shape == (n,) and rew.dtype == torch.float32
    assert term.shape == (n,) and term.dtype == torch.uint8 and trunc.dtype == torch.uint8
    assert len({t.data_ptr() for t in (obs, act, rew, term, trunc)}) == 5
    assert float(obs.abs().sum()) == 0.0  # the library hands out zeroed buffers
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(task, horizon=40), threads=8)
    drive(ext, oracle, h, o, obs, act, rew, term, trunc, 100, seed)
    # a second export of the same buffer is the same memory
    again = torch.from_dlpack(ext.vec_dlpack(h, "observations"))
    assert again.data_ptr() == obs.data_ptr()
    # the default action buffer of vec_fill_random_actions is the library-owned one
    ext.vec_fill_random_actions(h)
    o.fill_random_actions()
    torch.cuda.synchronize()
    assert_bits_equal(o.actions, act.cpu().numpy(), "random actions in the library-owned buffer")
    # close is refused while views are alive, and works once they are gone
    with pytest.raises(RuntimeError, match="DLPack view"):
        ext.vec_close(h)
    ext.vec_step(h)  # the env is still usable after the refused close
    del obs, act, rew, term, trunc, again
    gc.collect()
    ext.vec_close(h)
    with pytest.raises(ValueError, match="closed"):
        ext.vec_dlpack(h, "observations")


@pytest.mark.gpu
def test_exported_tensors_keep_the_env_alive(ext):
    import torch

    n = 2048
    h = ext.vec_init(None, None, None, None, None, n, 5)
    ext.vec_reset(h, 5)
    obs = torch.from_dlpack(ext.vec_dlpack(h, "observations"))
    torch.cuda.synchronize()
    want = obs.clone()
    del h  # the last Python reference to the env; the tensor's manager context holds another
    gc.collect()
    junk = [torch.full((n, 20), 7.0, device="cuda") for _ in range(8)]  # would land in the freed block if it were freed
    torch.cuda.synchronize()
    assert torch.equal(obs, want)
    del junk
    # an unconsumed capsule releases its reference too (nothing to assert beyond "does not crash or leak the env")
    h2 = ext.vec_init(None, None, None, None, None, n, 5)
    cap = ext.vec_dlpack(h2, "rewards")
    del cap
    gc.collect()
    ext.vec_close(h2)  # no live views: allowed


@pytest.mark.gpu
def test_dlpack_export_argument_checks(ext):
    import torch

    n = 256
    h = ext.vec_init(None, None, None, None, None, n, 0)
    with pytest.raises(ValueError, match="unknown buffer"):
        ext.vec_dlpack(h, "nope")
    ext.vec_close(h)
    host = (np.zeros((n, 20), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros(n, np.uint8))
    hh = ext.vec_init(*host, n, 0)
    with pytest.raises(TypeError, match="host buffers"):
        ext.vec_dlpack(hh, "observations")
    ext.vec_close(hh)
    with pytest.raises(TypeError, match="None five times"):
        ext.vec_init(None, torch.zeros(n, 4, device="cuda"), None, None, None, n, 0)
    # caller-owned torch tensors can be exported too (e.g. to hand them to a second framework)
    t = (torch.zeros(n, 20, device="cuda"), torch.zeros(n, 4, device="cuda"), torch.zeros(n, device="cuda"),
         torch.zeros(n, dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda"))
    ht = ext.vec_init(*t, n, 0)
    v = torch.from_dlpack(ext.vec_dlpack(ht, "actions"))
    assert v.data_ptr() == t[1].data_ptr()
    del v
    gc.collect()
    ext.vec_close(ht)


@pytest.mark.gpu
@pytest.mark.parametrize("task", [0, 2])
def test_dlpack_producers_as_env_buffers(ext, oracle, task):
    import torch

    n, seed = 4096, 31
    od = abi.obs_dim(task)
    obs, act, rew = torch.zeros(n, od, device="cuda"), torch.zeros(n, 4, device="cuda"), torch.zeros(n, device="cuda")
    term, trunc = torch.zeros(n, dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.bool, device="cuda")
    h = ext.vec_init(*(OnlyDLPack(t) for t in (obs, act, rew, term, trunc)), n, seed, task=task, horizon=40)
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(task, horizon=40), threads=8)
    drive(ext, oracle, h, o, obs, act, rew, term, trunc.view(torch.uint8), 60, seed)
    # K-major blocks through DLPack producers as well
    K = 4
    b_act = torch.zeros(K, n, 4, device="cuda")
    b_obs, b_rew = torch.zeros(K, n, od, device="cuda"), torch.zeros(K, n, device="cuda")
    b_term, b_trunc = torch.zeros(K, n, dtype=torch.uint8, device="cuda"), torch.zeros(K, n, dtype=torch.uint8, device="cuda")
    acts = np.zeros((K, n, 4), np.float32)
    for k in range(K):
        o.fill_random_actions()
        acts[k] = o.actions
        o.step()
    b_act.copy_(torch.from_numpy(acts))
    torch.cuda.synchronize()
    ext.vec_step_many(h, K, *(OnlyDLPack(t) for t in (b_act, b_obs, b_rew, b_term, b_trunc)))
    torch.cuda.synchronize()
    assert_bits_equal(o.observations, b_obs[K - 1].cpu().numpy(), "step_many through DLPack blocks: last obs")
    assert_bits_equal(o.rewards, b_rew[K - 1].cpu().numpy(), "step_many through DLPack blocks: last rewards")
    ext.vec_fill_random_actions(h, OnlyDLPack(act))
    o.fill_random_actions()
    torch.cuda.synchronize()
    assert_bits_equal(o.actions, act.cpu().numpy(), "fill_random_actions through a DLPack producer")
    ext.vec_close(h)


@pytest.mark.gpu
def test_dlpack_import_refuses_wrong_tensors(ext):
    import torch

    n = 256
    good = [torch.zeros(n, 20, device="cuda"), torch.zeros(n, 4, device="cuda"), torch.zeros(n, device="cuda"),
            torch.zeros(n, dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda")]

    def init_with(i, bad):
        b = [OnlyDLPack(t) for t in good]
        b[i] = OnlyDLPack(bad)
        return ext.vec_init(*b, n, 0)

    with pytest.raises(TypeError, match="float32"):
        init_with(0, torch.zeros(n, 20, dtype=torch.float64, device="cuda"))
    with pytest.raises(TypeError, match="uint8"):
        init_with(3, torch.zeros(n, dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError, match="too small"):
        init_with(1, torch.zeros(n, 2, device="cuda"))
    with pytest.raises(ValueError, match="contiguous"):
        init_with(0, torch.zeros(n, 40, device="cuda")[:, ::2])
    with pytest.raises(ValueError, match="not on a ROCm device"):
        init_with(2, torch.zeros(n))
    with pytest.raises(TypeError, match="all host buffers or all device"):
        ext.vec_init(np.zeros((n, 20), np.float32), *(OnlyDLPack(t) for t in good[1:]), n, 0)
    # nothing above may have leaked a consumed capsule: the tensors are still usable and freeable
    h = ext.vec_init(*(OnlyDLPack(t) for t in good), n, 0)
    ext.vec_reset(h, 0)
    ext.vec_close(h)


@pytest.mark.gpu
def test_c_abi_library_owned_buffers(hip):
    """The same through the plain C-ABI (ctypes): NULL pointers on a device handle, drone_vec_buffers, drone_vec_device."""
    import ctypes as C

    import torch

    from drone_amd import binding

    lib = binding.load()
    cfg = abi.DroneConfig()
    lib.drone_config_default(C.byref(cfg), 0)
    cfg.buffer_kind = abi.BUFFERS_DEVICE
    v = lib.drone_vec_init(None, None, None, None, None, 1024, 3, C.byref(cfg))
    assert v, lib.drone_last_error()
    ptrs = [C.c_void_p() for _ in range(5)]
    assert lib.drone_vec_buffers(v, *(C.byref(p) for p in ptrs)) == 0
    assert all(p.value for p in ptrs) and len({p.value for p in ptrs}) == 5
    assert ptrs[0].value % 16 == 0 and ptrs[1].value % 16 == 0
    assert lib.drone_vec_device(v) == 0
    lib.drone_vec_reset(v, 3)
    lib.drone_vec_step(v)
    assert lib.drone_vec_sync(v) == 0 and lib.drone_vec_status(v) == 0
    lib.drone_vec_close(v)
    # a host handle may not pass NULL, and a device handle may not pass only some
    cfg.buffer_kind = abi.BUFFERS_HOST
    assert not lib.drone_vec_init(None, None, None, None, None, 1024, 3, C.byref(cfg))
    assert b"NULL" in lib.drone_last_error()
    cfg.buffer_kind = abi.BUFFERS_DEVICE
    t = torch.zeros(1024, 20, device="cuda")
    assert not lib.drone_vec_init(C.c_void_p(t.data_ptr()), None, None, None, None, 1024, 3, C.byref(cfg))


@pytest.mark.gpu
def test_protocol_objects_for_newer_consumers(ext, oracle):
    """drone_amd.dlpack.Buffer: __dlpack__ / __dlpack_device__ over the capsules (consumers that refuse bare capsules)."""
    import torch

    from drone_amd import dlpack

    n, seed = 2048, 9
    h = ext.vec_init(None, None, None, None, None, n, seed, task=1)
    objs = dlpack.buffers(h)
    assert objs[0].__dlpack_device__() == (10, ext.vec_device(h)) == (10, 0)  # kDLROCM
    obs, act, rew, term, trunc = (torch.from_dlpack(o) for o in objs)  # torch passes stream= and max_version=
    assert obs.shape == (n, 20) and act.shape == (n, 4) and term.dtype == torch.uint8 and obs.is_cuda
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(1), threads=8)
    drive(ext, oracle, h, o, obs, act, rew, term, trunc, 30, seed)
    side = torch.cuda.Stream()
    ext.vec_set_stream(h, side.cuda_stream)
    o.fill_random_actions()
    act.copy_(torch.from_numpy(o.actions))
    torch.cuda.synchronize()
    o.step()
    ext.vec_step(h)  # asynchronous on `side`
    again = torch.from_dlpack(dlpack.buffer(h, "observations"))  # the export drains the env's stream for the consumer's
    assert_bits_equal(o.observations, again.cpu().numpy(), "observations exported right after an asynchronous step")
    with pytest.raises(ValueError, match="unknown buffer"):
        dlpack.buffer(h, "nope")
    ext.vec_sync(h)
    del obs, act, rew, term, trunc, again
    gc.collect()
    ext.vec_close(h)

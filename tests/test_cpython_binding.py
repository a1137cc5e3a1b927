"""The compiled CPython binding (bindings/drone_binding.c): PufferLib-style
vec_init / vec_reset / vec_step / vec_log / vec_close over the caller's buffers.
CPU: it builds with gcc alone, imports, parses arguments and fails loudly without
a GPU. GPU: driven against the oracle through host (numpy) and device (torch) buffers."""
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
    import torch  # noqa: F401  (same HIP runtime instance as the tensors)

    from drone_amd import drone_binding

    return drone_binding


def host_buffers(n, task):
    return (np.zeros((n, abi.obs_dim(task)), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32),
            np.zeros(n, np.uint8), np.zeros(n, np.uint8))


def test_module_surface_and_argument_checks(ext):
    for name in ("vec_init", "vec_reset", "vec_step", "vec_log", "vec_close", "vec_rollout", "vec_set_stream", "vec_fill_random_actions", "vec_gstep",
                 "vec_send", "vec_recv", "vec_step_many", "vec_step_repeat", "vec_done_list_at", "vec_dlpack", "vec_device", "vec_sync", "vec_host_transport", "vec_host_pin", "vec_host_unpin", "vec_variant"):
        assert callable(getattr(ext, name))
    assert ext.obs_dim(0) == 20 and ext.obs_dim(3) == 24 and ext.TASK_SWARM == 2
    b = host_buffers(16, 0)
    with pytest.raises(TypeError, match="unknown env kwarg"):
        ext.vec_init(*b, 16, 0, not_a_field=1)
    with pytest.raises(ValueError, match="observations holds"):
        ext.vec_init(np.zeros((16, 10), np.float32), *b[1:], 16, 0)
    with pytest.raises((BufferError, ValueError, TypeError)):
        ext.vec_init(b[0][:, ::2], *b[1:], 16, 0)  # not C-contiguous
    ro = np.zeros(16, np.float32)
    ro.setflags(write=False)
    with pytest.raises((BufferError, ValueError, TypeError)):
        ext.vec_init(b[0], b[1], ro, b[3], b[4], 16, 0)  # read-only rewards
    with pytest.raises(ValueError):
        ext.vec_init(*b, 0, 0)


def test_no_gpu_means_loud_failure(ext):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="drone_vec_init failed"):
        ext.vec_init(*host_buffers(16, 0), 16, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("task", [0, 1, 2, 3])
def test_binding_matches_oracle_host_buffers(ext, oracle, task):
    n, seed = 4096, 17
    obs, act, rew, term, trunc = host_buffers(n, task)
    h = ext.vec_init(obs, act, rew, term, trunc, n, seed, task=task, horizon=40)
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(task, horizon=40), threads=8)
    ext.vec_reset(h, seed)
    o.reset(seed)
    assert_bits_equal(o.observations, obs, "reset obs")
    for t in range(120):
        o.fill_random_actions()
        act[:] = o.actions
        o.step()
        ext.vec_step(h)
        assert_bits_equal(o.observations, obs, f"obs {t}")
        assert_bits_equal(o.rewards, rew, f"rew {t}")
        assert_bits_equal(o.terminals, term, f"term {t}")
        assert_bits_equal(o.truncations, trunc, f"trunc {t}")
    assert ext.vec_gstep(h) == 120
    lh, lo = ext.vec_log(h), o.log()
    assert lh["n"] == lo["n"] and lh["n"] > 0
    for k in ("perf", "score", "episode_return", "episode_length", "oob"):
        assert lh[k] == pytest.approx(lo[k], rel=1e-6, abs=1e-7)
    ext.vec_close(h)
    with pytest.raises(ValueError, match="closed"):
        ext.vec_step(h)
    ext.vec_close(h)  # idempotent


@pytest.mark.gpu
def test_binding_device_tensors_and_rollout(ext, oracle):
    import torch

    n, seed, task = 10000, 3, 1
    dev = torch.device("cuda:0")
    obs = torch.zeros((n, 20), dtype=torch.float32, device=dev)
    act = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    rew = torch.zeros(n, dtype=torch.float32, device=dev)
    term = torch.zeros(n, dtype=torch.uint8, device=dev)
    trunc = torch.zeros(n, dtype=torch.uint8, device=dev)
    h = ext.vec_init(obs, act, rew, term, trunc, n, seed, task=task, horizon=30, wind_sigma=2.0)
    ext.vec_set_stream(h, torch.cuda.current_stream().cuda_stream)
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(task, horizon=30, wind_sigma=2.0), threads=8)
    ext.vec_reset(h, seed)
    o.reset(seed)
    for t in range(50):
        o.fill_random_actions()
        ext.vec_fill_random_actions(h)
        o.step()
        ext.vec_step(h)
    torch.cuda.synchronize()
    assert_bits_equal(o.actions, act, "device random policy")
    assert_bits_equal(o.observations, obs, "obs")
    assert_bits_equal(o.rewards, rew, "rew")
    o.rollout(33)
    ext.vec_rollout(h, 33)
    torch.cuda.synchronize()
    assert_bits_equal(o.observations, obs, "rollout obs")
    assert_bits_equal(o.rewards, rew, "rollout reward sums")
    assert_bits_equal(o.truncations, trunc, "rollout truncations")
    with pytest.raises(RuntimeError, match="horizon must be positive"):
        ext.vec_rollout(h, 0)
    assert ext.vec_variant(h).startswith("drone_step_kernel<task=1,compact=0,mem=0,dt=0>") and "bytes=310" in ext.vec_variant(h)
    del h  # capsule destructor closes the env
    # round 4: the state layout as an env kwarg (DroneConfig.state_layout) reaches the library through the kwargs table
    o2 = [torch.zeros((512, 20), dtype=torch.float32, device=dev), torch.zeros((512, 4), dtype=torch.float32, device=dev), torch.zeros(512, dtype=torch.float32, device=dev),
          torch.zeros(512, dtype=torch.uint8, device=dev), torch.zeros(512, dtype=torch.uint8, device=dev)]
    h2 = ext.vec_init(*o2, 512, seed, task=0, state_layout=2)
    assert "dt=1" in ext.vec_variant(h2) and "bytes=262" in ext.vec_variant(h2)
    del h2
    with pytest.raises(RuntimeError, match="state_layout"):
        ext.vec_init(*o2, 512, seed, task=1, state_layout=2)


def test_buffer_format_and_size_checks(ext):
    """a buffer of sufficient byte length but the wrong item type (float64, int32) must be refused, not
    written through a wrong layout; vec_fill_random_actions checks the size of what it is handed."""
    b = host_buffers(16, 0)
    with pytest.raises(TypeError, match="observations: expected float32"):
        ext.vec_init(np.zeros((16, 20), np.float64), *b[1:], 16, 0)
    with pytest.raises(TypeError, match="rewards: expected float32"):
        ext.vec_init(b[0], b[1], np.zeros(16, np.int32), b[3], b[4], 16, 0)
    with pytest.raises(TypeError, match="terminals: expected uint8"):
        ext.vec_init(b[0], b[1], b[2], np.zeros(16, np.float32), b[4], 16, 0)
    assert callable(ext.vec_step_many) and callable(ext.vec_step_repeat) and callable(ext.vec_done_list_at)


@pytest.mark.gpu
def test_fill_random_actions_refuses_an_undersized_buffer(ext):
    n = 1024
    h = ext.vec_init(*host_buffers(n, 0), n, 0)
    ext.vec_reset(h, 0)
    with pytest.raises(ValueError, match="actions holds"):
        ext.vec_fill_random_actions(h, np.zeros((n // 2, 4), np.float32))  # the library would copy n*16 bytes into it
    with pytest.raises(TypeError, match="expected float32"):
        ext.vec_fill_random_actions(h, np.zeros((n, 4), np.float64))
    ok = np.zeros((n, 4), np.float32)
    ext.vec_fill_random_actions(h, ok, 5)
    assert np.abs(ok).max() > 0
    ext.vec_close(h)


@pytest.mark.gpu
@pytest.mark.parametrize("task,device", [(0, False), (1, True), (2, True), (3, False)])
def test_binding_step_many_matches_oracle(ext, oracle, task, device):
    """vec_step_many through the compiled binding: K in {1, 2, 7, 32} against K plain oracle steps, ragged env count."""
    import torch

    n, seed = 1000, 23
    od = abi.obs_dim(task)
    mk = (lambda *shape, dt=np.float32: np.zeros(shape, dt))
    if device:
        tmap = {np.float32: torch.float32, np.uint8: torch.uint8}
        mk = (lambda *shape, dt=np.float32: torch.zeros(shape, dtype=tmap[dt], device="cuda:0"))
    base = (mk(n, od), mk(n, 4), mk(n), mk(n, dt=np.uint8), mk(n, dt=np.uint8))
    h = ext.vec_init(*base, n, seed, task=task, horizon=20, compact_done=1)
    if device:
        ext.vec_set_stream(h, torch.cuda.current_stream().cuda_stream)
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(task, horizon=20), threads=8)
    ext.vec_reset(h, seed)
    o.reset(seed)
    host = (lambda t: t.cpu().numpy() if device else t)
    for K in (1, 2, 7, 32):
        acts = np.stack([o.fill_random_actions(gstep=o.gstep + k).copy() for k in range(K)])
        a_blk = mk(K, n, 4)
        if device:
            a_blk.copy_(torch.from_numpy(acts))
        else:
            a_blk[:] = acts
        blocks = (mk(K, n, od), mk(K, n), mk(K, n, dt=np.uint8), mk(K, n, dt=np.uint8))
        obs, rew, term, trunc, done = o.step_many(K, acts)
        ext.vec_step_many(h, K, a_blk, *blocks)
        if device:
            torch.cuda.synchronize()
        for k in range(K):
            assert_bits_equal(obs[k], host(blocks[0])[k], f"K={K} obs {k}")
            assert_bits_equal(rew[k], host(blocks[1])[k], f"K={K} rew {k}")
            assert_bits_equal(term[k], host(blocks[2])[k], f"K={K} term {k}")
            assert_bits_equal(trunc[k], host(blocks[3])[k], f"K={K} trunc {k}")
            ids = np.sort(np.frombuffer(ext.vec_done_list_at(h, k), dtype=np.uint32))
            assert_bits_equal(done[k], ids, f"K={K} done ids {k}")
    assert ext.vec_gstep(h) == o.gstep == 42
    # the random policy in the kernel (actions = None), and a short block refused
    blocks = (mk(5, n, od), mk(5, n), mk(5, n, dt=np.uint8), mk(5, n, dt=np.uint8))
    obs, rew, term, trunc, _ = o.step_many(5, None)
    ext.vec_step_many(h, 5, None, *blocks)
    if device:
        torch.cuda.synchronize()
    assert_bits_equal(obs, host(blocks[0]), "policy obs")
    assert_bits_equal(rew, host(blocks[1]), "policy rew")
    with pytest.raises(ValueError, match="observations holds"):
        ext.vec_step_many(h, 6, None, *blocks)
    # action repeat: ONE [N][4] block for the five steps
    acts = o.fill_random_actions(gstep=777).copy()
    one = mk(n, 4)
    if device:
        one.copy_(torch.from_numpy(acts))
    else:
        one[:] = acts
    obs, rew, term, trunc, _ = o.step_many(5, np.broadcast_to(acts, (5, n, 4)))
    ext.vec_step_repeat(h, 5, one, *blocks)
    if device:
        torch.cuda.synchronize()
    assert_bits_equal(obs, host(blocks[0]), "repeat obs")
    assert_bits_equal(trunc, host(blocks[3]), "repeat truncations")
    with pytest.raises(ValueError, match="actions is None"):
        ext.vec_step_repeat(h, 5, None, *blocks)
    ext.vec_close(h)


def test_dlpack_argument_checks_need_no_gpu(ext):
    """The DLPack import path validates the managed tensor before the library (and with it the GPU) is touched."""
    import torch

    class OnlyDLPack:
        def __init__(self, t):
            self._t = t

        def __dlpack__(self, stream=None):
            return self._t.__dlpack__()

    n = 16
    cpu = [torch.zeros(n, 20), torch.zeros(n, 4), torch.zeros(n), torch.zeros(n, dtype=torch.uint8), torch.zeros(n, dtype=torch.uint8)]
    with pytest.raises(ValueError, match="not on a ROCm device"):
        ext.vec_init(*(OnlyDLPack(t) for t in cpu), n, 0)
    with pytest.raises(TypeError, match="None five times"):
        ext.vec_init(None, None, None, None, np.zeros(n, np.uint8), n, 0)
    with pytest.raises(TypeError, match="all host buffers or all device"):
        ext.vec_init(np.zeros((n, 20), np.float32), *(OnlyDLPack(t) for t in cpu[1:]), n, 0)
    # a refused producer keeps its memory: the capsule was not consumed
    assert float(cpu[0].sum()) == 0.0 and cpu[0].data_ptr() != 0
    assert callable(ext.vec_dlpack)


@pytest.mark.gpu
def test_binding_pinned_step_many_blocks(ext, oracle):
    """vec_host_pin: page-owning numpy blocks pinned once, then written by vec_step_many in place."""
    import mmap

    n, K, seed = 1024, 5, 2

    def page_array(shape, dtype):
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        m = mmap.mmap(-1, (nbytes + 4095) // 4096 * 4096)
        return np.frombuffer(m, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    obs, act, rew, term, trunc = host_buffers(n, 0)
    h = ext.vec_init(obs, act, rew, term, trunc, n, seed)
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(0), threads=4)
    ext.vec_reset(h, seed)
    o.reset(seed)
    blocks = [page_array((K, n, 4), np.float32), page_array((K, n, 20), np.float32), page_array((K, n), np.float32), page_array((K, n), np.uint8), page_array((K, n), np.uint8)]
    for b in blocks:
        ext.vec_host_pin(h, b, 1)
    with pytest.raises(RuntimeError, match="4 KiB"):
        ext.vec_host_pin(h, np.zeros(5000, np.uint8)[8:], 0)
    assert ext.vec_gstep(h) == 0  # the refused pin did not poison the handle
    for rep in range(3):
        acts = np.random.default_rng(rep).uniform(-1, 1, (K, n, 4)).astype(np.float32)
        blocks[0][...] = acts
        want = o.step_many(K, acts)
        ext.vec_step_many(h, K, *blocks)
        assert_bits_equal(want[0], blocks[1], f"obs {rep}")
        assert_bits_equal(want[1], blocks[2], f"rew {rep}")
        assert_bits_equal(want[2], blocks[3], f"term {rep}")
    for b in blocks:
        ext.vec_host_unpin(h, b)
    ext.vec_close(h)

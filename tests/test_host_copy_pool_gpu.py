"""Host-buffer transport 3 (round 5): mid-size shards whose buffers the library may not pin — a vec-env
worker's unaligned slices of shared memory, 16 384 ... ~10^5 envs. The kernel reads / writes pinned stand-ins over PCIe; a
small pool of host threads moves the action rows in as parallel slices and the outputs out WHILE the step kernel runs, each
256-drone chunk as soon as its workgroup says its rows have landed (per-chunk words in pinned host memory). Everything must
stay bit-exact against the oracle: every step's outputs, the synchronous form and the send / recv halves, resets and
rollouts in between (whole-buffer parallel copies), rebinding, two such handles out of phase on one thread (the second
finds the pool busy and copies alone), close between send and recv."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import assert_outputs_equal, assert_state_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def heap_buffers(n, od, skew=16):
    """Plain heap arrays, deliberately off any page boundary: nothing the library may pin."""
    mk = lambda shape, dt: np.zeros(int(np.prod(shape)) + skew, dt)[skew:].reshape(shape)  # noqa: E731
    return (mk((n, od), np.float32), mk((n, 4), np.float32), mk((n,), np.float32), mk((n,), np.uint8), mk((n,), np.uint8))


@pytest.mark.parametrize("task,n", [(0, 16384), (0, 70001), (1, 40000), (2, 32768), (3, 20011)])
def test_pool_transport_is_bit_exact(hip, oracle, task, n):
    from drone_amd import abi

    seed = 13
    kw = dict(horizon=30, compact_done=1)
    h = hip.DroneVec(n, seed=seed, task=task, buffers=heap_buffers(n, abi.obs_dim(task)), **kw)
    assert h.host_transport == "stand-in-mt"
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(task, **kw), threads=8)
    o.reset(seed)
    h.reset(seed)
    assert_outputs_equal(o, h, "reset")
    for t in range(70):
        o.fill_random_actions()
        h.actions[:] = o.actions
        o.step()
        if t % 3 == 2:
            h.step()
        else:
            h.step_send()
            h.step_recv()
        assert_outputs_equal(o, h, f"step {t}")
        if t % 10 == 9:
            want = np.flatnonzero(o.terminals | o.truncations).astype(np.uint32)
            assert np.array_equal(want, np.sort(h.done_list())), f"done list {t}"
        if t == 33:  # a rollout and a reset in between: whole-buffer deliveries through the same pool
            o.rollout(9)
            h.rollout(9)
            assert_outputs_equal(o, h, "rollout")
        if t == 50:
            o.reset(seed + 1)
            h.reset(seed + 1)
            assert_outputs_equal(o, h, "second reset")
    # stand-ins deliver to wherever the caller points: rebinding keeps the transport
    new = heap_buffers(n, abi.obs_dim(task), skew=48)
    h.bind_actions(new[1])
    h.bind_outputs(new[0], new[2], new[3], new[4])
    for t in range(5):
        o.fill_random_actions()
        new[1][:] = o.actions
        o.step()
        h.step()
        assert_outputs_equal(o, h, f"after rebinding, step {t}")
    assert h.host_transport == "stand-in-mt"
    assert_state_equal(o.get_state(), h.get_state(), "state")
    assert h.status() == (0, "")
    h.close()


def test_two_pool_handles_out_of_phase_on_one_thread(hip, oracle):
    """a.send, b.send, a.recv, b.recv from ONE thread: the pool serves `a` from its send to its recv; `b` must notice that
    and do its own copying instead of waiting for a recv that can only come after its own send returns."""
    n, seed = 20000, 3
    a = hip.DroneVec(n, seed=seed, task=1, env_offset=0, buffers=heap_buffers(n, 20))
    b = hip.DroneVec(n, seed=seed, task=1, env_offset=n, buffers=heap_buffers(n, 20))
    assert a.host_transport == b.host_transport == "stand-in-mt"
    o = oracle.OracleVec(2 * n, seed=seed, cfg=oracle.default_config(1), threads=8)
    o.reset(seed)
    a.reset(seed)
    b.reset(seed)
    for t in range(40):
        o.fill_random_actions()
        a.actions[:] = o.actions[:n]
        b.actions[:] = o.actions[n:]
        o.step()
        a.step_send()
        b.step_send()
        a.step_recv()
        b.step_recv()
        assert a.observations.tobytes() == o.observations[:n].tobytes() and b.observations.tobytes() == o.observations[n:].tobytes(), f"step {t}"
        assert a.rewards.tobytes() == o.rewards[:n].tobytes() and b.rewards.tobytes() == o.rewards[n:].tobytes(), f"step {t}"
        assert b.terminals.tobytes() == o.terminals[n:].tobytes() and a.truncations.tobytes() == o.truncations[:n].tobytes(), f"step {t}"
    a.close()
    b.close()


def test_close_between_send_and_recv_releases_the_pool(hip, oracle):
    n, seed = 30000, 5
    a = hip.DroneVec(n, seed=seed, task=0, buffers=heap_buffers(n, 20))
    a.reset(seed)
    a.step_send()
    a.close()  # the pool was delivering this handle's outputs: close must let it finish and give it back
    b = hip.DroneVec(n, seed=seed, task=0, buffers=heap_buffers(n, 20))
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(0), threads=8)
    o.reset(seed)
    b.reset(seed)
    for t in range(10):
        o.fill_random_actions()
        b.actions[:] = o.actions
        o.step()
        b.step_send()
        b.step_recv()
        assert_outputs_equal(o, b, f"the next handle, step {t}")
    b.close()


def test_without_a_pool_such_shards_take_the_mirror_transport():
    """DRONE_HOST_COPY_THREADS=1 is read when the process-wide pool starts: a process of its own."""
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
            "import numpy as np\nfrom drone_amd import binding\nfrom test_host_copy_pool_gpu import heap_buffers\n"
            "h = binding.DroneVec(16384, seed=1, task=0, buffers=heap_buffers(16384, 20))\nprint('TRANSPORT', h.host_transport)\nh.reset(1); h.step(); h.close()\n"
            # between the pool's hand-over (512 KiB of unpinnable buffers) and the single memcpy's own budget (1 MiB): one memcpy, as before the pool
            "h = binding.DroneVec(8192, seed=1, task=0, buffers=heap_buffers(8192, 20))\nprint('TRANSPORT8192', h.host_transport)\nh.reset(1); h.step(); h.close()\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, DRONE_HOST_COPY_THREADS="1"))
    assert r.returncode == 0, r.stderr[-1500:]
    assert "TRANSPORT mirror" in r.stdout and "TRANSPORT8192 stand-in\n" in r.stdout, r.stdout


@pytest.mark.parametrize("n,want", [(4096, "stand-in"), (5000, "stand-in"), (6144, "stand-in-mt"), (8192, "stand-in-mt")])
def test_where_the_pool_takes_over_from_the_single_memcpy(hip, oracle, n, want):
    """DRONE_HOST_POOL_MIN_BYTES, default 512 KiB of unpinnable buffers (102 B per hover env): measured at equal cost at
    4 096 envs and 1.4x ahead at 8 192 (profiles/r05_ab/pool_hand_over.txt). Both sides of it bit for bit."""
    seed = 5
    h = hip.DroneVec(n, seed=seed, task=0, buffers=heap_buffers(n, 20), horizon=20)
    assert h.host_transport == want
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(0, horizon=20), threads=8)
    o.reset(seed)
    h.reset(seed)
    for t in range(45):
        o.fill_random_actions()
        h.actions[:] = o.actions
        o.step()
        h.step()
        assert_outputs_equal(o, h, f"step {t}")
    h.close()

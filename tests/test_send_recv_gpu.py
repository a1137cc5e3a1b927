"""drone_vec_step_send / drone_vec_step_recv: step in two halves (a vec-env's async send / recv). send + recv equals step
bit for bit on every transport; two handles stepping out of phase stay exact; misuse fails loudly and sticks."""
import numpy as np
import pytest

from helpers import assert_outputs_equal, assert_state_equal


def heap_buffers(n, od):
    return (np.zeros((n, od), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros(n, np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["zero-copy", "stand-in", "mirror", "device"])
@pytest.mark.parametrize("task", [0, 1, 2, 3])
def test_send_recv_equals_step(hip, oracle, monkeypatch, kind, task):
    from drone_amd import abi

    n, seed = 4096, 19
    kw = dict(horizon=30, compact_done=1)
    if kind == "mirror":
        monkeypatch.setenv("DRONE_HOST_ZEROCOPY", "0")
    bufs = heap_buffers(n, abi.obs_dim(task)) if kind == "stand-in" else None
    h = hip.DroneVec(n, seed=seed, task=task, device="cuda:0" if kind == "device" else None, buffers=bufs, **kw)
    if kind != "device":
        assert h.host_transport == kind
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(task, **kw), threads=8)
    o.reset(seed)
    h.reset(seed)
    for t in range(90):
        o.fill_random_actions()
        if kind == "device":
            import torch

            h.actions.copy_(torch.from_numpy(o.actions))
            torch.cuda.synchronize()
        else:
            h.actions[:] = o.actions
        o.step()
        if t % 3 == 2:
            h.step()  # the synchronous form in between: same state machine
        else:
            h.step_send()
            h.step_recv()
        if kind == "device":
            h.sync()
        assert_outputs_equal(o, h, f"{kind} step {t}")
        want = np.flatnonzero(o.terminals | o.truncations).astype(np.uint32)
        assert np.array_equal(want, np.sort(h.done_list())), f"done list {t}"
    assert_state_equal(o.get_state(), h.get_state(), "state")
    assert h.gstep == o.gstep
    h.close()


@pytest.mark.gpu
def test_two_handles_out_of_phase(hip, oracle):
    """The use the pair exists for: two host-buffer shards, each sent before the other is received."""
    n, seed = 2048, 3
    a = hip.DroneVec(n, seed=seed, task=1, env_offset=0)
    b = hip.DroneVec(n, seed=seed, task=1, env_offset=n)
    o = oracle.OracleVec(2 * n, seed=seed, cfg=oracle.default_config(1), threads=8)
    o.reset(seed)
    a.reset(seed)
    b.reset(seed)
    o.fill_random_actions()
    a.actions[:] = o.actions[:n]
    a.step_send()
    for t in range(60):
        b.actions[:] = o.actions[n:]
        b.step_send()  # B goes out while A is in flight
        a.step_recv()
        o.step()
        assert np.array_equal(o.observations[:n].view(np.uint32), a.observations.view(np.uint32)), f"A obs {t}"
        assert np.array_equal(o.rewards[:n], a.rewards) and np.array_equal(o.terminals[:n], a.terminals)
        b.step_recv()
        assert np.array_equal(o.observations[n:].view(np.uint32), b.observations.view(np.uint32)), f"B obs {t}"
        assert np.array_equal(o.truncations[n:], b.truncations)
        o.fill_random_actions()
        a.actions[:] = o.actions[:n]
        a.step_send()  # A's next step goes out before B's outputs are even looked at by a consumer
    a.step_recv()
    a.close()
    b.close()


@pytest.mark.gpu
def test_send_recv_misuse_is_loud(hip):
    h = hip.DroneVec(512, seed=1)
    h.reset(1)
    with pytest.raises(RuntimeError, match="no step was sent"):
        h.step_recv()
    h.clear_status()
    h.step_send()
    for call, pattern in ((h.step_send, "not been received"), (h.step, "not been received"), (lambda: h.rollout(4), "not been received"),
                          (lambda: h.reset(1), "not been received"), (h.log, "not been received")):
        with pytest.raises(RuntimeError, match=pattern):
            call()
        h.clear_status()
    with pytest.raises(RuntimeError, match="not been received"):
        h.get_state()
    h.clear_status()
    h.sync()  # allowed
    h.step_recv()
    assert h.status() == (0, "")
    assert h.gstep == 1
    h.step()
    h.step_send()
    h.close()  # closing with a step in flight drains it


@pytest.mark.gpu
def test_env_class_send_recv(hip, oracle):
    from drone_amd.env import Drone

    e = Drone(num_envs=1024, task="hover", seed=4, log_interval=0)
    o = oracle.OracleVec(1024, seed=4, cfg=oracle.default_config(0), threads=4)
    o.reset(4)
    e.reset(4)
    for _ in range(20):
        o.fill_random_actions()
        e.send(o.actions)
        obs, rew, term, trunc, infos = e.recv()
        o.step()
        assert np.array_equal(o.observations.view(np.uint32), obs.view(np.uint32)) and np.array_equal(o.rewards, rew)
    assert e.tick == 20
    e.close()

"""Graph-safe stepping (drone_vec_enable_graph_capture): the step counters live in HBM and the kernels advance them,
so a step / rollout captured into a hipGraph (here through torch.cuda.graph) replays with advancing wind, policy and
done-list parity — bit-exact against the oracle stepping the same number of times."""
import numpy as np
import pytest

from drone_amd import abi
from helpers import assert_bits_equal, assert_outputs_equal, assert_state_equal

pytestmark = pytest.mark.gpu


def make_pair(oracle, hip, n, seed, task, **over):
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(task, **over), threads=8)
    h = hip.DroneVec(n, seed=seed, cfg=hip.default_config(task, **over), device="cuda:0")
    o.reset(seed)
    h.reset(seed)
    return o, h


@pytest.mark.parametrize("task", [0, 1, 3])
def test_device_counters_without_a_graph(oracle, hip, task):
    """Same calls as always, counters read from HBM: steps, fused rollouts in between, the done list, set_gstep."""
    import torch

    n = 5000
    o, h = make_pair(oracle, hip, n, 12, task, horizon=11, compact_done=1)
    h.enable_graph_capture(True)
    for rnd in range(6):
        for _ in range(7):
            o.fill_random_actions()
            h.fill_random_actions()
            o.step()
            h.step()
        want = np.flatnonzero(o.terminals | o.truncations).astype(np.uint32)
        assert_bits_equal(want, np.sort(h.done_list()), f"round {rnd}: done ids")
        o.rollout(5)
        h.rollout(5)
        assert h.gstep == o.gstep
    torch.cuda.synchronize()
    assert_outputs_equal(o, h, "outputs")
    assert_state_equal(o.get_state(), h.get_state(), "state")
    h.enable_graph_capture(False)  # back to launch-argument counters, in step with the device's
    for _ in range(5):
        o.fill_random_actions()
        h.fill_random_actions()
        o.step()
        h.step()
    torch.cuda.synchronize()
    assert h.gstep == o.gstep
    assert_state_equal(o.get_state(), h.get_state(), "state after switching back")


def test_captured_step_replays_with_advancing_counters(oracle, hip):
    """torch.cuda.graph around ONE env step of the waypoint task (its wind is keyed on the step counter), replayed 80
    times with a constant action buffer: equals 80 oracle steps. A captured launch with the counter in its arguments
    would replay step 0's wind 80 times."""
    import torch

    n, seed, replays = 20000, 5, 80
    o, h = make_pair(oracle, hip, n, seed, 1, horizon=30, compact_done=1)
    h.enable_graph_capture(True)
    h.fill_random_actions(gstep=123)          # one fixed action batch, used every step on both sides
    actions = h.actions.cpu().numpy().copy()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        h.use_torch_stream()
        for _ in range(3):                    # warm-up on the capture stream, as torch asks
            h.step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        h.use_torch_stream()                  # the capturing stream
        h.step()                              # captured, not executed
    for _ in range(replays):
        g.replay()
    torch.cuda.synchronize()
    assert h.gstep == 3 + replays
    for _ in range(3 + replays):
        o.actions[:] = actions
        o.step()
    assert_outputs_equal(o, h, "after replays")
    assert_state_equal(o.get_state(), h.get_state(), "state after replays")
    want = np.flatnonzero(o.terminals | o.truncations).astype(np.uint32)
    assert_bits_equal(want, np.sort(h.done_list()), "done ids of the last replayed step")
    # and the negative control: without device counters the replayed wind stays at the captured step's
    o2, h2 = make_pair(oracle, hip, 4096, seed, 1, horizon=10**6, bound=1e6)
    h2.fill_random_actions(gstep=7)
    with torch.cuda.stream(side):
        h2.use_torch_stream()
        h2.step()
    torch.cuda.current_stream().wait_stream(side)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=side):
        h2.use_torch_stream()
        h2.step()
    for _ in range(20):
        g2.replay()
    torch.cuda.synchronize()
    o2.actions[:] = h2.actions.cpu().numpy()
    for _ in range(21):
        o2.step()
    assert not np.array_equal(o2.get_state()["wind"], h2.get_state()["wind"]), "frozen launch-argument counter expected to diverge"


def test_captured_rollout_replays(oracle, hip):
    import torch

    n, seed = 8192, 9
    o, h = make_pair(oracle, hip, n, seed, 0, horizon=50)
    h.enable_graph_capture(True)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        h.use_torch_stream()
        h.rollout(16)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        h.use_torch_stream()
        h.rollout(16)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    for _ in range(6):
        o.rollout(16)
    assert h.gstep == o.gstep == 96
    assert_outputs_equal(o, h, "rollout replays")
    assert_state_equal(o.get_state(), h.get_state(), "state")


def test_host_buffers_refuse_graph_mode(hip):
    h = hip.DroneVec(256)
    with pytest.raises(RuntimeError, match="device buffers"):
        h.enable_graph_capture(True)
    h.clear_status()


def test_captured_step_many_replays(oracle, hip):
    """drone_vec_step_many under torch.cuda.graph: K = 6 steps with the in-kernel policy and per-step done lists captured
    ONCE (the counters in HBM, the done-count memset a graph node of its own), replayed 9 times = 54 oracle steps; the
    blocks after the last replay hold that replay's six steps. Storage is grown by a first call outside the capture."""
    import torch

    n, seed, K, replays = 12000, 6, 6, 9
    o, h = make_pair(oracle, hip, n, seed, 1, horizon=17, compact_done=1)
    h.enable_graph_capture(True)
    bufs = h.alloc_step_many(K)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        h.use_torch_stream()
        h.step_many(bufs, policy=True)        # warm-up: sizes the per-step list storage (a sync, not capturable)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        h.use_torch_stream()
        h.step_many(bufs, policy=True)
    for _ in range(replays):
        g.replay()
    torch.cuda.synchronize()
    assert h.gstep == K * (1 + replays)
    last = None
    for _ in range(1 + replays):
        last = o.step_many(K, None)
    obs, rew, term, trunc, done = last
    assert_bits_equal(obs, bufs.observations, "observations of the last replay")
    assert_bits_equal(rew, bufs.rewards, "rewards of the last replay")
    assert_bits_equal(term, bufs.terminals, "terminals")
    assert_bits_equal(trunc, bufs.truncations, "truncations")
    for k in range(K):
        assert_bits_equal(done[k], np.sort(h.done_list_at(k)), f"done ids of step {k} of the last replay")
    assert_state_equal(o.get_state(), h.get_state(), "state after the replays")

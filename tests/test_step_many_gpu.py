"""drone_vec_step_many (K env steps per launch WITH every step's outputs) against K plain c_step passes of the CPU
oracle: bit-exact observations / rewards / flags of every step, done-id lists per step, state and log sums afterwards.
Tasks 0-3, ragged env counts (the last workgroup takes the guarded path, the others the static one), K in {1, 2, 7, 32},
host and device blocks, caller-staged actions and the in-kernel random policy, through the ctypes C-ABI binding (the
compiled CPython binding has its own test in test_cpython_binding.py)."""
import numpy as np
import pytest

from helpers import assert_bits_equal, assert_state_equal, to_np

pytestmark = pytest.mark.gpu


def pair(oracle, hip, n, seed, task, device=None, **over):
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(task, **over), threads=8)
    h = hip.DroneVec(n, seed=seed, cfg=hip.default_config(task, **over), device=device)
    o.reset(seed)
    h.reset(seed)
    return o, h


def stage_actions(o, bufs, K, g0):
    """The K action rows of the random policy for steps g0..g0+K-1 (any staged rows would do; these give resets)."""
    acts = np.stack([o.fill_random_actions(gstep=g0 + k).copy() for k in range(K)])
    if type(bufs.actions).__module__.startswith("torch"):
        import torch

        bufs.actions.copy_(torch.from_numpy(acts))
    else:
        bufs.actions[:] = acts
    return acts


def check_call(o, h, bufs, K, what, policy=False, lists=False):
    acts = None if policy else stage_actions(o, bufs, K, o.gstep)
    obs, rew, term, trunc, done = o.step_many(K, acts)
    h.step_many(bufs, policy=policy)
    if h.torch_device is not None:
        import torch

        torch.cuda.synchronize()
    assert h.gstep == o.gstep
    for k in range(K):
        assert_bits_equal(obs[k], to_np(bufs.observations)[k], f"{what} obs step {k}")
        assert_bits_equal(rew[k], to_np(bufs.rewards)[k], f"{what} rew step {k}")
        assert_bits_equal(term[k], to_np(bufs.terminals)[k], f"{what} term step {k}")
        assert_bits_equal(trunc[k], to_np(bufs.truncations)[k], f"{what} trunc step {k}")
        if lists:
            assert_bits_equal(done[k], np.sort(h.done_list_at(k)), f"{what} done ids step {k}")
    return int(term.sum()) + int(trunc.sum())


@pytest.mark.parametrize("task", [0, 1, 2, 3])
@pytest.mark.parametrize("device", [None, "cuda:0"])
def test_step_many_equals_k_plain_steps(oracle, hip, task, device):
    n = 1000 if task != 2 else 1000 // 8 * 8  # ragged: 3 full workgroups + a partial one
    o, h = pair(oracle, hip, n, 17 + task, task, device=device, horizon=20)  # 49 steps: every env is truncated and reset twice
    ends = 0
    for j, K in enumerate((1, 2, 7, 32, 7)):
        bufs = h.alloc_step_many(K, pinned=bool(j % 2))  # host handles: staged blocks and pinned ones (accessed in place), alternating
        ends += check_call(o, h, bufs, K, f"task {task} K={K}")
        assert_state_equal(o.get_state(), h.get_state(), f"task {task} state after K={K}")
    assert ends >= 2 * n  # episode ends (and the resets inside the K steps) were exercised
    # and plain stepping continues from where step_many left off
    o.fill_random_actions()
    if device is None:
        h.actions[:] = o.actions
    else:
        h.fill_random_actions()
    o.step()
    h.step()
    assert_bits_equal(o.observations, h.observations, "plain step after step_many")
    lo, lh = o.log(), h.log()
    assert lo["n"] == lh["n"] and lo["n"] > 0
    for key in lo:
        assert lh[key] == pytest.approx(lo[key], rel=1e-6, abs=1e-7), key


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 256, 257, 4097, 4096 + 16])
def test_step_many_ragged_sizes(oracle, hip, n):
    """n % 16 != 0 sends the flag bytes down the byte path (step k's slice of the [K][n] flag blocks is unaligned);
    n = 4112 is 16-byte aligned with a partial last workgroup."""
    o, h = pair(oracle, hip, n, 5, 1, device="cuda:0", horizon=20)
    bufs = h.alloc_step_many(9)
    for rep in range(4):
        check_call(o, h, bufs, 9, f"n={n} call {rep}")
    assert_state_equal(o.get_state(), h.get_state(), f"n={n}")


@pytest.mark.parametrize("task", [0, 3])
def test_step_many_device_policy(oracle, hip, task):
    """actions == NULL: the kernel draws the SPEC.md random policy itself — a fused rollout that keeps every step's outputs."""
    o, h = pair(oracle, hip, 3000, 3, task, device="cuda:0", horizon=64)
    bufs = h.alloc_step_many(40)
    for rep in range(3):
        check_call(o, h, bufs, 40, f"policy task {task} call {rep}", policy=True)
    # the same trajectory as the fused rollout kernel and as plain stepping
    h2 = hip.DroneVec(3000, seed=3, cfg=hip.default_config(task, horizon=64), device="cuda:0")
    h2.reset(3)
    h2.rollout(120)
    assert_state_equal(h.get_state(), h2.get_state(), "step_many(policy) vs fused rollout")


def test_step_many_done_lists(oracle, hip):
    n = 3000
    o, h = pair(oracle, hip, n, 8, 0, device="cuda:0", horizon=40, compact_done=1)
    bufs = h.alloc_step_many(16)
    ends = 0
    for rep in range(6):
        ends += check_call(o, h, bufs, 16, f"lists call {rep}", lists=True)
    assert ends > n
    # a smaller K afterwards reuses the storage; a plain step invalidates the per-step lists of the last call
    small = h.alloc_step_many(3)
    check_call(o, h, small, 3, "lists K=3", lists=True)
    with pytest.raises(RuntimeError, match="outside the last step_many"):
        h.done_list_at(3)
    assert h.status()[0] != 0  # like every failed call it sticks to the handle until cleared
    h.clear_status()
    h.fill_random_actions()
    o.fill_random_actions()
    h.step()
    o.step()
    with pytest.raises(RuntimeError, match="not drone_vec_step_many"):
        h.done_list_at(0)
    h.clear_status()
    assert_bits_equal(np.flatnonzero(o.terminals | o.truncations).astype(np.uint32), np.sort(h.done_list()), "plain done list after step_many")


def test_step_many_rejects_bad_arguments(hip):
    import torch

    h = hip.DroneVec(512, seed=0, device="cuda:0")
    h.reset(0)
    bufs = h.alloc_step_many(4)
    bufs.k_steps = 0
    with pytest.raises((RuntimeError, ValueError)):
        h.step_many(bufs)
    h.clear_status()
    bad = h.alloc_step_many(2)
    bad.observations = torch.zeros(2 * 512 * 20 + 1, dtype=torch.float32, device="cuda:0")[1:].view(2, 512, 20)  # 4 bytes off a 16-byte boundary
    with pytest.raises(RuntimeError, match="16-byte aligned"):
        h.step_many(bad)
    h.clear_status()
    assert h.gstep == 0  # neither call advanced the env
    h.step_many(h.alloc_step_many(2), policy=True)
    assert h.gstep == 2


def test_step_many_at_config2_size_matches_plain_stepping(hip):
    """65 536 envs (BASELINE configs[1]), K = 32: step_many against the per-step kernel on the device itself, every step's
    outputs (the oracle comparison above runs at sizes it finishes in seconds)."""
    import torch

    n, K = 65536, 32
    a = hip.DroneVec(n, seed=1, device="cuda:0")
    b = hip.DroneVec(n, seed=1, device="cuda:0")
    a.reset(1)
    b.reset(1)
    bufs = a.alloc_step_many(K)
    for rep in range(4):
        for k in range(K):
            a.fill_random_actions(gstep=rep * K + k, out=bufs.actions[k])
        a.step_many(bufs)
        for k in range(K):
            b.bind_actions(bufs.actions[k])
            b.step()
            assert torch.equal(b.observations.view(torch.int32), bufs.observations[k].view(torch.int32)), (rep, k)
            assert torch.equal(b.rewards.view(torch.int32), bufs.rewards[k].view(torch.int32))
            assert torch.equal(b.terminals, bufs.terminals[k]) and torch.equal(b.truncations, bufs.truncations[k])
    torch.cuda.synchronize()
    assert_state_equal(a.get_state(), b.get_state(), "65536 envs, 128 steps")


@pytest.mark.parametrize("task,device", [(0, None), (1, "cuda:0"), (3, "cuda:0")])
def test_step_repeat_is_k_steps_under_one_action_block(oracle, hip, task, device):
    """drone_vec_step_repeat (action repeat / frame skip): K env steps in one launch with ONE [N][4] action block —
    exactly K plain steps with an unchanged action buffer, every step's outputs kept."""
    n = 2500
    o, h = pair(oracle, hip, n, 33, task, device=device, horizon=25)
    bufs = h.alloc_step_many(8)
    for rep in range(6):
        acts = o.fill_random_actions(gstep=1000 + rep).copy()
        if device is None:
            h.actions[:] = acts
        else:
            import torch

            h.actions.copy_(torch.from_numpy(acts))
        obs, rew, term, trunc, _ = o.step_many(8, np.broadcast_to(acts, (8, n, 4)))
        h.step_repeat(bufs)
        if device is not None:
            torch.cuda.synchronize()
        assert_bits_equal(obs, to_np(bufs.observations), f"repeat {rep} obs")
        assert_bits_equal(rew, to_np(bufs.rewards), f"repeat {rep} rewards")
        assert_bits_equal(term, to_np(bufs.terminals), f"repeat {rep} terminals")
        assert_bits_equal(trunc, to_np(bufs.truncations), f"repeat {rep} truncations")
    assert_state_equal(o.get_state(), h.get_state(), "state after repeats")
    assert h.gstep == o.gstep == 48


@pytest.mark.gpu
def test_host_blocks_pinned_and_not(oracle, hip):
    """drone_vec_host_pin: pinned K-major host blocks are read / written by the kernel in place, unpinned ones go through
    staging; same results; the page rule is enforced; plain heap blocks still work (staged)."""
    import ctypes as C

    n, K, seed = 2048, 6, 8
    o, h = pair(oracle, hip, n, seed, 1, device=None, horizon=15, compact_done=1)
    for pinned in (True, False, True):
        bufs = h.alloc_step_many(K, pinned=pinned)
        check_call(o, h, bufs, K, f"pinned={pinned}", lists=True)
    heap = hip.StepManyBuffers(K, np.zeros((K, n, 4), np.float32), np.zeros((K, n, 20), np.float32), np.zeros((K, n), np.float32),
                               np.zeros((K, n), np.uint8), np.zeros((K, n), np.uint8))
    check_call(o, h, heap, K, "heap blocks", lists=True)
    # the rule: an unaligned block is refused, not registered
    odd = np.zeros(8192 + 64, np.uint8)[64:]
    assert h._f["drone_vec_host_pin"](h._h, odd.ctypes.data, odd.nbytes, 0) == -1
    assert b"4 KiB" in h._f["drone_last_error"]()
    h.clear_status()
    ok = hip.page_buffer((4096,), np.uint8)
    # round 5: page-aligned whole pages are not enough — the caller must vouch that the block is a mapping of its own
    assert h._f["drone_vec_host_pin"](h._h, ok.ctypes.data, ok.nbytes, 0) == -1 and b"vouched" in h._f["drone_last_error"]()
    h.clear_status()
    assert h._f["drone_vec_host_pin"](h._h, ok.ctypes.data, ok.nbytes, 1) == 0
    assert h._f["drone_vec_host_pin"](h._h, ok.ctypes.data, ok.nbytes, 0) == 0  # already pinned: fine
    assert h._f["drone_vec_host_unpin"](h._h, ok.ctypes.data) == 0
    assert_state_equal(o.get_state(), h.get_state(), "state")
    h.close()

"""The derived-target state layout (round 3): hover and swarm handles whose step is HBM-bound store
no target plane — the episode counter moves into P4, tick and score_count share a word, and the kernels re-derive the
target from (reset key, env, episode) — 262 instead of 278 bytes per env-step for hover. Forced on here at test sizes
(DRONE_DERIVED_TARGET=1; by default the host picks it from ~2^18 envs on, which the full-size tests of test_configs_gpu.py
then exercise): every kernel and the AoS state interface must behave exactly as with the six-plane layout."""
import numpy as np
import pytest

from helpers import assert_bits_equal, assert_outputs_equal, assert_state_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def force_layout(monkeypatch):
    monkeypatch.setenv("DRONE_DERIVED_TARGET", "1")


def pair(oracle, hip, n, seed, task, device=None, **over):
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(task, **over), threads=8)
    h = hip.DroneVec(n, seed=seed, cfg=hip.default_config(task, **over), device=device)
    o.reset(seed)
    h.reset(seed)
    return o, h


@pytest.mark.parametrize("task,n,device", [(0, 3001, None), (0, 4096, "cuda:0"), (2, 4096, None), (2, 1024 + 64, "cuda:0")])
def test_every_kernel_matches_the_oracle_in_the_derived_target_layout(oracle, hip, task, n, device):
    over = dict(horizon=30, env_offset=4096)
    if task == 2:
        over.update(agents_per_env=8, collision_radius=0.5)
    o, h = pair(oracle, hip, n, 21, task, device=device, **over)
    assert h.bytes_per_env_step == (262 if task == 0 else 278)  # 16 bytes fewer than the six-plane layout (278 / 294)
    assert_state_equal(o.get_state(), h.get_state(), "reset state (targets re-derived on the host)")
    for t in range(90):  # three horizons: every env resets at least twice, episodes advance, targets change
        o.fill_random_actions()
        if device is None:
            h.actions[:] = o.actions
        else:
            h.fill_random_actions()
        o.step()
        h.step()
        assert_outputs_equal(o, h, f"step {t}")
    assert_state_equal(o.get_state(), h.get_state(), "state after steps")
    o.rollout(45)
    h.rollout(45)
    assert_outputs_equal(o, h, "fused rollout")
    assert_state_equal(o.get_state(), h.get_state(), "state after the rollout")
    bufs = h.alloc_step_many(7)
    obs, rew, term, trunc, _ = o.step_many(7, None)
    h.step_many(bufs, policy=True)
    if device is not None:
        import torch

        torch.cuda.synchronize()
    assert_bits_equal(obs, bufs.observations, "step_many obs")
    assert_bits_equal(rew, bufs.rewards, "step_many rewards")
    st = h.get_state()
    assert_state_equal(o.get_state(), st, "state after step_many")
    assert st["episode"].min() >= 3 and len(np.unique(st["target"], axis=0)) > n // 2
    lo, lh = o.log(), h.log()
    assert lo["n"] == lh["n"] > 0


def test_state_rows_round_trip_and_what_the_layout_refuses(oracle, hip):
    n, seed = 2000, 9
    o, h = pair(oracle, hip, n, seed, 0, horizon=25)
    for t in range(60):
        o.fill_random_actions()
        h.actions[:] = o.actions
        o.step()
        h.step()
    rows = h.get_state()
    # checkpoint / restore through the AoS interface: a second handle continues bit for bit
    h2 = hip.DroneVec(n, seed=seed, cfg=hip.default_config(0, horizon=25))
    h2.reset(seed)
    h2.set_state(rows)
    h2.set_gstep(h.gstep)
    # unaligned sub-ranges (tiles at the edges hold neighbours)
    part = h.get_state(first=77, count=333)
    assert_state_equal(rows[77:77 + 333], part, "sub-range")
    h2.set_state(part, first=77)
    for t in range(40):
        o.fill_random_actions()
        h.actions[:] = o.actions
        h2.actions[:] = o.actions
        o.step()
        h.step()
        h2.step()
    assert_outputs_equal(o, h2, "restored handle")
    assert_state_equal(h.get_state(), h2.get_state(), "restored handle state")
    # a free-form target cannot be represented: refused loudly, not silently replaced
    bad = rows.copy()
    bad["target"][5] += 0.25
    with pytest.raises(RuntimeError, match="derived-target layout"):
        h.set_state(bad)
    h.clear_status()
    bad = rows.copy()
    bad["tick"][3] = 70000
    with pytest.raises(RuntimeError, match="derived-target layout"):
        h.set_state(bad)
    h.clear_status()


def test_tasks_and_horizons_the_layout_cannot_hold_keep_the_target_plane(hip):
    for task, over, want in ((1, {}, 310), (3, {}, 310), (0, {"horizon": 70000}, 278), (0, {"horizon": 65535}, 262)):
        h = hip.DroneVec(512, seed=0, cfg=hip.default_config(task, **over))
        assert h.bytes_per_env_step == want, (task, over)
        h.close()


def test_shards_in_the_derived_layout_equal_one_vec(oracle, hip):
    """The re-derived target is keyed on the GLOBAL env id like everything else."""
    n, seed = 3000, 4
    whole = hip.DroneVec(n, seed=seed, cfg=hip.default_config(0, horizon=20))
    parts = [hip.DroneVec(c, seed=seed, cfg=hip.default_config(0, horizon=20, env_offset=off)) for off, c in ((0, 1100), (1100, 1900))]
    for v in [whole] + parts:
        v.reset(seed)
    for t in range(50):
        whole.fill_random_actions()
        whole.step()
        for v in parts:
            v.fill_random_actions()
            v.step()
    joined = np.concatenate([v.get_state() for v in parts])
    assert_state_equal(whole.get_state(), joined, "sharded vs whole")
    assert_bits_equal(whole.observations, np.concatenate([v.observations for v in parts]), "observations")

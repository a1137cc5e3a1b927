"""The PufferLib-shaped env class (drone_amd/env.py) over the C-ABI."""
import numpy as np
import pytest


def test_env_class_needs_a_gpu(hip):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from drone_amd.env import Drone

    with pytest.raises(RuntimeError, match="drone_vec_init failed"):
        Drone(num_envs=8)


@pytest.mark.gpu
@pytest.mark.parametrize("device", [None, "cuda:0"])
def test_env_class_matches_oracle_and_reports_logs(oracle, hip, device):
    from drone_amd.env import Drone
    from helpers import assert_bits_equal

    n, seed = 512, 3
    env = Drone(num_envs=n, task="waypoint", device=device, seed=seed, log_interval=16, horizon=40)
    assert env.num_agents == n and env.single_observation_space.shape == (20,) and env.single_action_space.shape == (4,)
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(1, horizon=40))
    obs, infos = env.reset(seed)
    o.reset(seed)
    assert infos == []
    assert_bits_equal(o.observations, obs, "reset obs")
    got_log = []
    for t in range(96):
        a = o.fill_random_actions().copy()
        if device is not None:
            import torch

            a_in = torch.from_numpy(a).to(device)
        else:
            a_in = a
        obs, rew, term, trunc, infos = env.step(a_in)
        o.step()
        got_log += infos
    assert_bits_equal(o.observations, obs, "obs")
    assert_bits_equal(o.rewards, rew, "rewards")
    assert_bits_equal(o.terminals, term, "terminals")
    assert_bits_equal(o.truncations, trunc, "truncations")
    assert len(got_log) >= 1 and all(l["n"] > 0 and "episode_return" in l for l in got_log)
    total_n = sum(l["n"] for l in got_log)
    assert total_n == o.log()["n"]
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("device", [None, "cuda:0"])
def test_env_class_step_many(oracle, hip, device):
    """Drone.step_many: K-step action segments (and the device policy) through the PufferLib-shaped class."""
    from drone_amd.env import Drone
    from helpers import assert_bits_equal

    n, seed, K = 700, 4, 12
    env = Drone(num_envs=n, task="hover", device=device, seed=seed, log_interval=16, horizon=30)
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(0, horizon=30))
    env.reset(seed)
    o.reset(seed)
    logs = []
    for rnd in range(4):
        acts = np.stack([o.fill_random_actions(gstep=o.gstep + k).copy() for k in range(K)])
        a_in = acts
        if device is not None:
            import torch

            a_in = torch.from_numpy(acts).to(device)
        want = o.step_many(K, acts)
        obs, rew, term, trunc, infos = env.step_many(a_in)
        logs += infos
        for name, w, g in zip(("obs", "rew", "term", "trunc"), want[:4], (obs, rew, term, trunc)):
            assert_bits_equal(w, g, f"round {rnd} {name}")
    want = o.step_many(5, None)
    obs, rew, term, trunc, infos = env.step_many(k_steps=5)  # the device policy
    assert_bits_equal(want[0], obs, "policy obs")
    assert_bits_equal(want[1], rew, "policy rewards")
    assert len(logs) >= 1 and all(l["n"] > 0 for l in logs)  # the log is read whenever a call crosses a log_interval boundary
    env.close()


@pytest.mark.gpu
def test_env_class_on_slices_of_a_shared_block(oracle, hip):
    """The PufferLib vec-env contract: the caller allocates ONE block per buffer kind and
    hands every env (worker) a slice of it; two envs here fill one block between them."""
    from types import SimpleNamespace

    from drone_amd.env import Drone
    from helpers import assert_bits_equal

    n, seed = 300, 8  # per env; 300 * 80 B keeps the second slice 16-B aligned
    block = SimpleNamespace(observations=np.zeros((2 * n, 20), np.float32), actions=np.zeros((2 * n, 4), np.float32),
                            rewards=np.zeros(2 * n, np.float32), terminals=np.zeros(2 * n, np.uint8), truncations=np.zeros(2 * n, np.uint8))
    envs = []
    for w in range(2):
        sl = slice(w * n, (w + 1) * n)
        buf = SimpleNamespace(**{k: getattr(block, k)[sl] for k in vars(block)})
        envs.append(Drone(num_envs=n, task="hover", seed=seed, log_interval=0, buf=buf, env_offset=w * n, horizon=35))
    o = oracle.OracleVec(2 * n, seed=seed, cfg=oracle.default_config(0, horizon=35))
    o.reset(seed)
    for e in envs:
        e.reset(seed)
    assert_bits_equal(o.observations, block.observations, "reset obs in the shared block")
    for t in range(80):
        o.fill_random_actions()
        block.actions[:] = o.actions          # the "policy" writes the shared action block
        o.step()
        for e in envs:
            e.step(e.actions)                 # each env steps on its own slice, in place
    assert_bits_equal(o.observations, block.observations, "obs")
    assert_bits_equal(o.rewards, block.rewards, "rewards")
    assert_bits_equal(o.terminals, block.terminals, "terminals")
    assert_bits_equal(o.truncations, block.truncations, "truncations")
    for e in envs:
        e.close()


def test_buffer_validation(hip):
    from drone_amd.binding import DroneVec

    n = 16
    good = (np.zeros((n, 20), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros(n, np.uint8))
    bad_shape = (np.zeros((n, 19), np.float32),) + good[1:]
    bad_dtype = (good[0], np.zeros((n, 4), np.float64)) + good[2:]
    strided = (np.zeros((n, 40), np.float32)[:, ::2],) + good[1:]
    for bufs, exc in ((bad_shape, ValueError), (bad_dtype, TypeError), (strided, ValueError)):
        with pytest.raises(exc):
            DroneVec(n, buffers=bufs)

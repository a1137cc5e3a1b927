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

"""Checkpoint / resume through files: a handle restored with load_checkpoint continues bit for bit like
the one that wrote the file — trajectories, resets, wind, policy draws, per-env log sums — and like the oracle."""
import numpy as np
import pytest

from helpers import assert_bits_equal, assert_outputs_equal, assert_state_equal, to_np


def run(v, steps):
    out = []
    for _ in range(steps):
        v.fill_random_actions()
        v.step()
        v.sync()
        out.append([to_np(x).copy() for x in (v.observations, v.rewards, v.terminals, v.truncations)])
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("task", [0, 1, 2, 3])
@pytest.mark.parametrize("device", [None, "cuda:0"])
def test_resume_from_file_continues_bit_for_bit(hip, oracle, tmp_path, task, device):
    n, seed = 3000 if task != 2 else 3072, 77
    kw = dict(horizon=25, substeps=2)
    a = hip.DroneVec(n, seed=seed, task=task, device=device, **kw)
    a.reset(seed)
    run(a, 40)
    path = str(tmp_path / "shard.npz")
    a.save_checkpoint(path)
    rest_a = run(a, 40)
    # resumed in a handle of the OTHER buffer kind that was doing something else before
    b = hip.DroneVec(n, seed=5, task=task, device=None if device else "cuda:0", **kw)
    b.reset(5)
    run(b, 7)
    b.load_checkpoint(path)
    rest_b = run(b, 40)
    for t, (x, y) in enumerate(zip(rest_a, rest_b)):
        for name, p, q in zip(("obs", "rew", "term", "trunc"), x, y):
            assert_bits_equal(p, q, f"step {t} after the checkpoint: {name}")
    assert_state_equal(a.get_state(), b.get_state(), "final state")
    assert a.gstep == b.gstep == 80
    # and both are what the oracle computes without any interruption
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(task, **kw), threads=8)
    o.reset(seed)
    for _ in range(80):
        o.fill_random_actions()
        o.step()
    assert_outputs_equal(o, b, "resumed run against the uninterrupted oracle")
    assert_state_equal(o.get_state(), b.get_state(), "resumed state against the uninterrupted oracle")
    la, lb = a.log(), b.log()  # (drains the per-env sums: after the state comparisons)
    assert la["n"] == lb["n"] > 0 and abs(la["episode_return"] - lb["episode_return"]) <= 1e-6 * max(1.0, abs(la["episode_return"]))
    a.close()
    b.close()


@pytest.mark.gpu
def test_checkpoint_restores_the_buffers_and_refuses_another_env(hip, tmp_path):
    n = 512
    a = hip.DroneVec(n, seed=1, task=1, device="cuda:0")
    a.reset(1)
    run(a, 5)
    path = str(tmp_path / "c.npz")
    a.save_checkpoint(path)
    b = hip.DroneVec(n, seed=9, task=1, device="cuda:0")
    b.load_checkpoint(path)
    for name in ("observations", "actions", "rewards", "terminals", "truncations"):
        assert_bits_equal(getattr(a, name), getattr(b, name), f"restored {name}")
    with pytest.raises(ValueError, match="envs"):
        hip.DroneVec(n + 64, seed=1, task=1, device="cuda:0").load_checkpoint(path)
    with pytest.raises(ValueError, match="horizon"):
        hip.DroneVec(n, seed=1, task=1, device="cuda:0", horizon=77).load_checkpoint(path)
    with pytest.raises(ValueError, match="task"):
        hip.DroneVec(n, seed=1, task=0, device="cuda:0").load_checkpoint(path)
    # a file from another SPEC version (or from before the field existed: SPEC v4 and earlier) means other log sums and
    # reset draws: refused rather than continued
    with np.load(path) as z:
        fields = {k: z[k] for k in z.files}
    assert int(fields["spec_version"]) == hip.DroneVec.SPEC_VERSION == 5
    for name, drop in (("old.npz", True), ("other.npz", False)):
        alt = dict(fields)
        if drop:
            del alt["spec_version"]
        else:
            alt["spec_version"] = np.int64(4)
        with open(str(tmp_path / name), "wb") as fh:
            np.savez(fh, **alt)
        with pytest.raises(ValueError, match="SPEC v4"):
            b.load_checkpoint(str(tmp_path / name))


@pytest.mark.gpu
def test_env_class_save_and_load(hip, tmp_path):
    from drone_amd.env import Drone

    e = Drone(num_envs=1024, task="waypoint", device="cuda:0", seed=3, log_interval=0)
    obs, _ = e.reset(3)
    for _ in range(20):
        e.vec.fill_random_actions()
        e.step(e.actions)
    path = str(tmp_path / "env.npz")
    e.save(path)
    for _ in range(20):
        e.vec.fill_random_actions()
        e.step(e.actions)
    f = Drone(num_envs=1024, task="waypoint", device="cuda:0", seed=0, log_interval=0)
    got = f.load(path)
    assert f.tick == 20 and f.seed == 3 and got is f.observations
    for _ in range(20):
        f.vec.fill_random_actions()
        f.step(f.actions)
    f.vec.sync()
    e.vec.sync()
    assert_bits_equal(e.observations, f.observations, "env resumed from file")
    assert np.array_equal(to_np(e.rewards), to_np(f.rewards))
    e.close()
    f.close()

"""Independent checks of the oracle (the only parity lever this repo has: the
reference holds no source, SURVEY.md §8c — parity stays UNPINNED).

1. tests/spec_numpy.py restates the WHOLE env step — integer RNG, wind, RK4
   dynamics, reward, bounds, waypoint / gate dealing, episode end, reset draws,
   log sums, observation — in float64 numpy, written from SPEC.md without looking
   at oracle/drone_oracle.h's expression order. It is run one step ahead of the
   float32 oracle for 300 steps with short horizons (so hundreds of resets), for
   tasks 0, 1 and 3 and substeps 1 and 3.
2. Statistics of the OU wind (SPEC.md §5 step 2): zero mean, stationary variance
   sigma^2 dt / (1 - decay^2) ~ sigma^2 / (2 theta), lag-1 correlation = decay,
   clamp respected.
3. Conservation laws the integrator must respect: torque-free tumbling keeps
   |I w|, the rotational energy and the WORLD-frame angular momentum vector
   R(q) I w (which ties the quaternion kinematics' sign conventions to the
   dynamics'); a thrown, spinning, unpowered drone keeps its total energy.
"""
import numpy as np
import pytest

import spec_numpy as sn

FLOAT_FIELDS = ("pos", "vel", "quat", "omega", "rpm", "target", "wind", "ep_return", "perf_sum", "score_sum", "ret_sum", "len_sum", "n_sum", "oob_sum")
INT_FIELDS = ("tick", "episode", "score_count")


def cfgdict(cfg):
    return {k: (float(v) if isinstance(v, float) else v) for k, v in cfg.as_dict().items()}


def rows_to_f64(rows):
    d = {f: rows[f].astype(np.float64) for f in FLOAT_FIELDS}
    d.update({f: rows[f].astype(np.int64) for f in INT_FIELDS})
    return d


@pytest.mark.parametrize("task,substeps,extra", [
    (0, 1, {}), (0, 3, {}),
    (1, 1, {"waypoint_radius": 1.5}), (1, 3, {"waypoint_radius": 1.5, "wind_sigma": 3.0}),
    (3, 1, {"gate_radius": 2.5}), (3, 3, {"gate_radius": 2.5}),
    (2, 1, {"agents_per_env": 8, "collision_radius": 0.6}), (2, 1, {"agents_per_env": 64, "collision_radius": 0.3, "c_proximity": 1.5}),
])
def test_whole_step_tracks_independent_float64_statement(oracle, task, substeps, extra):
    n, seed, off, steps = 768, 1234 + task, 4096, 300
    cfg = oracle.default_config(task, substeps=substeps, horizon=40, env_offset=off, **extra)
    v = oracle.OracleVec(n, seed=seed, cfg=cfg, threads=4)
    v.reset(seed)
    c = sn.derived(cfgdict(cfg))
    env_ids = np.arange(off, off + n, dtype=np.uint64)
    # reset: the float64 restatement of §6 against the oracle's reset state and observation
    st = v.get_state()
    fresh = sn.reset_draws(c, seed, env_ids, np.zeros(n, np.int64), task)
    for f in ("pos", "target", "quat", "rpm", "wind"):
        np.testing.assert_allclose(st[f], fresh[f], rtol=2e-6, atol=2e-6, err_msg=f"reset {f}")
    S0 = [fresh[k] for k in ("pos", "vel", "quat", "omega", "rpm")]
    np.testing.assert_allclose(v.observations, sn.full_obs(c, S0, fresh["target"], fresh["wind"], task), rtol=1e-5, atol=2e-6)

    ignored = events = ends = 0
    rng = np.random.default_rng(seed)
    for t in range(steps):
        if task == 3 and t % 5 == 2:
            # random actions never find a gate: every few steps aim the drones at theirs — some through the
            # middle, some near the rim, some backwards — so passes, misses and re-dealt gates all occur
            rows = v.get_state()
            nrm, ctr = rows["wind"].copy(), rows["target"].copy()
            side = np.cross(nrm, np.array([0.3, -0.5, 0.8], np.float32))
            side /= np.linalg.norm(side, axis=1, keepdims=True)
            lateral = rng.uniform(0.0, 1.4 * cfg.gate_radius, size=(n, 1)).astype(np.float32)
            direction = np.where(rng.random((n, 1)) < 0.8, 1.0, -1.0).astype(np.float32)
            rows["pos"] = ctr - direction * 0.03 * nrm + lateral * side
            rows["vel"] = direction * 6.0 * nrm
            v.set_state(rows)
        before = rows_to_f64(v.get_state())
        g = v.gstep
        v.fill_random_actions()
        np.testing.assert_allclose(v.actions, sn.random_actions(seed, env_ids, g), rtol=0, atol=0)  # exact: 16-bit grid
        acts = v.actions.astype(np.float64)
        v.step()
        new, (rew, term, trunc), ob, margin = sn.env_step(c, seed, task, before, acts, g, env_ids)
        ok = margin > 2e-4  # a wall / radius / gate plane closer than this may be decided differently in float32
        ignored += int((~ok).sum())
        after = v.get_state()
        assert np.array_equal(v.terminals.astype(bool)[ok], term[ok]), f"step {t}: terminals"
        assert np.array_equal(v.truncations.astype(bool)[ok], trunc[ok]), f"step {t}: truncations"
        np.testing.assert_allclose(v.rewards[ok], rew[ok], rtol=1e-4, atol=3e-5, err_msg=f"step {t}: reward")
        for f, tol in (("pos", 3e-6), ("vel", 3e-5), ("quat", 2e-6), ("omega", 3e-4), ("rpm", 3e-2), ("target", 2e-6), ("wind", 3e-6),
                       ("ep_return", 1e-4), ("perf_sum", 1e-5), ("score_sum", 1e-5), ("ret_sum", 2e-4), ("len_sum", 0), ("n_sum", 0), ("oob_sum", 0)):
            np.testing.assert_allclose(after[f][ok], new[f][ok], rtol=2e-6, atol=tol, err_msg=f"step {t}: {f}")
        for f in INT_FIELDS:
            assert np.array_equal(after[f][ok].astype(np.int64), new[f][ok]), f"step {t}: {f}"
        np.testing.assert_allclose(v.observations[ok], ob[ok], rtol=1e-5, atol=3e-5, err_msg=f"step {t}: observation")
        ends += int((term | trunc).sum())
        events += int((new["score_count"] > before["score_count"]).sum())
    assert ends > 5 * n, "the run must cross many episode ends"
    assert ignored < (0.05 if task == 2 else 0.002) * n * steps, f"{ignored} borderline env-steps ignored"
    if task in (1, 3):
        assert events > 20, "waypoints / gates must actually be reached"
    if task == 2:
        assert v.log()["oob"] > 0.02, "collisions must actually occur"
    v.close()


def test_ou_wind_statistics(oracle):
    """No resets (huge box, huge horizon), 1500 steps = 7.5 time constants."""
    n, steps = 4096, 1500
    cfg = oracle.default_config(1, bound=1e6, horizon=10**6, waypoint_radius=1e-9, max_vel=1e6)
    v = oracle.OracleVec(n, seed=77, cfg=cfg, threads=8)
    v.reset(77)
    v.actions[:] = 0.0
    prev = None
    num = den = 0.0
    for t in range(steps):
        v.step()
        if t >= steps - 40:
            w = v.get_state()["wind"].astype(np.float64)
            if prev is not None:
                num += (w * prev).sum()
                den += (prev * prev).sum()
            prev = w
    assert v.terminals.sum() == 0 and v.get_state()["episode"].max() == 0
    w = v.get_state()["wind"].astype(np.float64)
    decay = 1.0 - cfg.wind_theta * cfg.dt
    var_want = cfg.wind_sigma ** 2 * cfg.dt / (1.0 - decay * decay)   # = 1.0025 ~ sigma^2 / (2 theta)
    assert abs(var_want - cfg.wind_sigma ** 2 / (2 * cfg.wind_theta)) < 0.01
    assert abs(w.mean()) < 0.03                                        # 12288 samples: s.e. 0.009
    assert abs(w.var() / var_want - 1.0) < 0.05                        # s.e. of the variance 1.3 %
    for k in range(3):
        assert abs(w[:, k].var() / var_want - 1.0) < 0.08
    assert abs(num / den - decay) < 0.002                              # lag-1 regression coefficient
    assert abs(np.corrcoef(w[:, 0], w[:, 1])[0, 1]) < 0.05            # components independent
    assert np.abs(w).max() <= cfg.wind_max
    # the drag couples it into the dynamics: against the same run without gusts (the attitude is unaffected —
    # drag acts at the centre of mass), the velocity difference is a low-passed copy of the wind
    calm = oracle.OracleVec(n, seed=77, cfg=oracle.default_config(1, bound=1e6, horizon=10**6, waypoint_radius=1e-9, max_vel=1e6, wind_sigma=0.0), threads=8)
    calm.reset(77)
    calm.actions[:] = 0.0
    for _ in range(steps):
        calm.step()
    assert np.all(calm.get_state()["wind"] == 0)
    np.testing.assert_allclose(calm.get_state()["quat"], v.get_state()["quat"], atol=1e-5)
    dv = v.get_state()["vel"].astype(np.float64) - calm.get_state()["vel"].astype(np.float64)
    assert np.abs(dv).max() > 0.1
    for k in range(3):
        assert np.corrcoef(dv[:, k], w[:, k])[0, 1] > 0.25
    calm.close()
    v.close()


def test_ou_wind_clamp(oracle):
    n = 2048
    cfg = oracle.default_config(1, bound=1e6, horizon=10**6, waypoint_radius=1e-9, wind_max=1.0, max_vel=1e6)
    v = oracle.OracleVec(n, seed=5, cfg=cfg, threads=8)
    v.reset(5)
    v.actions[:] = 0.0
    for _ in range(600):
        v.step()
    w = v.get_state()["wind"]
    assert np.abs(w).max() == np.float32(1.0)
    at = (np.abs(w) == np.float32(1.0)).mean()
    assert 0.01 < at < 0.5   # a 1-sigma clamp: a visible share sits exactly on the rail (it leaves it again at the next decay step)
    v.close()


def _world_L(st, I):
    q = st["quat"].astype(np.float64)
    R = sn.rot(q)
    return np.einsum("nij,nj->ni", R, I * st["omega"].astype(np.float64))


def test_torque_free_tumbling_conserves_angular_momentum_and_energy(oracle):
    """k_ang_damp = 0, all rotors at one constant speed (zero net torque), asymmetric inertia so the
    body genuinely tumbles: |I w|, 1/2 w.I w and the world-frame vector R(q) I w stay put."""
    cfg = oracle.default_config(0, k_ang_damp=0.0, motor_tau=1e9, bound=1e6, horizon=10**6, k_drag=0.0,
                                ixx=1.0e-5, iyy=1.6e-5, izz=2.4e-5, max_vel=1e6)
    n = 16
    v = oracle.OracleVec(n, seed=2, cfg=cfg)
    v.reset(2)
    st = v.get_state()
    rng = np.random.default_rng(0)
    st["omega"][:] = rng.uniform(-6, 6, size=(n, 3)).astype(np.float32)
    v.set_state(st)
    v.actions[:] = 0.0  # the rotors never move off hover speed (motor_tau = 1e9 s), equal speeds -> zero torque
    I = np.array([cfg.ixx, cfg.iyy, cfg.izz], np.float64)
    st0 = v.get_state()
    L0 = np.linalg.norm(I * st0["omega"].astype(np.float64), axis=1)
    E0 = 0.5 * (I * st0["omega"].astype(np.float64) ** 2).sum(1)
    Lw0 = _world_L(st0, I)
    for _ in range(300):
        v.step()
    st1 = v.get_state()
    assert np.abs(st1["omega"] - st0["omega"]).max() > 0.5, "the body must actually tumble"
    L1 = np.linalg.norm(I * st1["omega"].astype(np.float64), axis=1)
    E1 = 0.5 * (I * st1["omega"].astype(np.float64) ** 2).sum(1)
    np.testing.assert_allclose(L1, L0, rtol=3e-5)
    np.testing.assert_allclose(E1, E0, rtol=6e-5)
    np.testing.assert_allclose(_world_L(st1, I), Lw0, rtol=0, atol=2e-4 * L0.max())
    np.testing.assert_allclose(np.linalg.norm(st1["quat"].astype(np.float64), axis=1), 1.0, atol=2e-6)
    v.close()


def test_unpowered_throw_conserves_total_energy(oracle):
    """Rotors stopped, no drag, no damping: kinetic + potential + rotational energy is constant."""
    cfg = oracle.default_config(0, k_ang_damp=0.0, k_drag=0.0, motor_tau=1e9, bound=1e6, horizon=10**6, max_vel=1e6)
    n = 8
    v = oracle.OracleVec(n, seed=3, cfg=cfg)
    v.reset(3)
    st = v.get_state()
    rng = np.random.default_rng(1)
    st["rpm"][:] = 0
    st["vel"][:] = rng.uniform(-3, 3, size=(n, 3)).astype(np.float32)
    st["omega"][:] = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32)
    v.set_state(st)
    v.actions[:] = -1.0
    I = np.array([cfg.ixx, cfg.iyy, cfg.izz], np.float64)

    def energy(s):
        vel, pos, om = (s[k].astype(np.float64) for k in ("vel", "pos", "omega"))
        return 0.5 * cfg.mass * (vel * vel).sum(1) + cfg.mass * cfg.gravity * pos[:, 2] + 0.5 * (I * om * om).sum(1)

    e0 = energy(v.get_state())
    for _ in range(200):
        v.step()
    s1 = v.get_state()
    scale = 0.5 * cfg.mass * (s1["vel"].astype(np.float64) ** 2).sum(1)  # the kinetic term by now dominates
    np.testing.assert_allclose(energy(s1), e0, rtol=0, atol=2e-5 * scale.max())
    v.close()


def test_integrator_converges_at_fourth_order(oracle):
    """A property no transcription error survives: with the step size halved (substeps 1 -> 2 -> 4 at a deliberately
    coarse dt = 0.04 s) the oracle's distance from a finely resolved float64 solution (the numpy statement at 64
    substeps) falls by 2^4 per halving — RK4 for the rigid body AND the closed-form rotor lag evaluated at the right
    stage times (a first-order slip anywhere, e.g. rotor inputs taken at the wrong instant, would show as a ratio of 2).
    Float32 rounding sets the floor near 1e-5, so the third halving is only required to keep improving."""
    import drift

    n, seed, dt, steps = 256, 11, 0.04, 10

    def cfg(substeps):
        return oracle.default_config(0, substeps=substeps, dt=dt, bound=1.0e4, horizon=1 << 30, max_vel=1.0e4, max_omega=1.0e4)

    v0 = oracle.OracleVec(n, seed=seed, cfg=cfg(1), threads=2)
    v0.reset(seed)
    acts = [v0.fill_random_actions(gstep=t // 4).copy() * 0.5 for t in range(steps)]  # each action held for four steps
    rows = v0.get_state()
    c = sn.derived(cfgdict(cfg(64)))
    ref = [rows[f].astype(np.float64) for f in drift.FIELDS]
    for t in range(steps):
        ref = sn.step(c, ref, acts[t].astype(np.float64), np.zeros((n, 3)))
    errs = []
    for s in (1, 2, 4, 8):
        v = oracle.OracleVec(n, seed=seed, cfg=cfg(s), threads=2)
        v.reset(seed)
        for t in range(steps):
            v.actions[:] = acts[t]
            v.step()
            assert not v.terminals.any()
        got = v.get_state()
        errs.append(max(float(np.abs(got[f].astype(np.float64) - r).max()) for f, r in zip(drift.FIELDS[:4], ref[:4])))
    print("error vs substeps 1, 2, 4, 8:", ["%.3e" % e for e in errs])
    assert 12.0 < errs[0] / errs[1] < 20.0 and 12.0 < errs[1] / errs[2] < 20.0  # 2^4 = 16 per halving
    assert errs[3] < errs[2] / 4.0                                                # still falling, into the float32 floor

"""CPU tests of the oracle itself (-m "not gpu").

PARITY UNPINNED: the reference has no golden vectors, tests or source for this
path. What pins the oracle here: (1) the committed golden
vectors it generated (regression), (2) known-answer tests of its integer RNG
against a pure-Python statement of SPEC.md §2, (3) closed-form physics cases,
(4) an independent float64 numpy statement of the dynamics (tests/spec_numpy.py).
"""
import os

import numpy as np
import pytest

from helpers import assert_bits_equal, assert_state_equal

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
M32 = 0xFFFFFFFF


# ---- SPEC.md §2 in pure Python ----
def py_hash32(x):
    x &= M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return x


def py_stream_key(seed, stream):
    return py_hash32((seed & M32) ^ py_hash32(((seed >> 32) & M32) ^ ((0x9E3779B9 * (stream + 1)) & M32)))


def py_draw(key, env, ctr, d):
    base = py_hash32((py_hash32(key ^ env) + ctr * 0x9E3779B9) & M32)
    return py_hash32((base + d * 0x85EBCA6B) & M32)


def test_rng_known_answers(oracle):
    lib = oracle.lib()
    assert py_hash32(0) == 0 and lib.oracle_hash32(0) == 0
    for x in (1, 2, 0xDEADBEEF, 0xFFFFFFFF, 123456789):
        assert lib.oracle_hash32(x) == py_hash32(x)
    for seed in (0, 1, 20251017, 2**63 + 12345, 2**64 - 1):
        for s in range(4):
            assert lib.oracle_stream_key(seed, s) == py_stream_key(seed, s)
    key = py_stream_key(42, 1)
    for env, ctr, d in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (2**20 - 1, 999, 8), (2**32 - 1, 2**32 - 1, 50)):
        assert lib.oracle_rng_draw(key, env, ctr, d) == py_draw(key, env, ctr, d)


def test_random_policy_known_answers(oracle):
    n, seed, off = 257, 77, 1000
    v = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(0, env_offset=off))
    key = py_stream_key(seed, 1)
    for g in (0, 5, 2**31):
        a = v.fill_random_actions(gstep=g)
        for i in (0, 1, 100, 256):
            k = py_hash32(key ^ (off + i))
            h0, h1 = py_hash32((k + (2 * g) * 0x9E3779B9) & M32), py_hash32((k + (2 * g + 1) * 0x9E3779B9) & M32)
            want = [((h0 & 0xFFFF) - 32768) / 32768.0, ((h0 >> 16) - 32768) / 32768.0,
                    ((h1 & 0xFFFF) - 32768) / 32768.0, ((h1 >> 16) - 32768) / 32768.0]
            assert a[i].tolist() == want
    assert a.min() >= -1.0 and a.max() < 1.0
    big = oracle.OracleVec(1 << 16, seed=3).fill_random_actions(gstep=9)
    assert abs(float(big.mean())) < 0.01 and abs(float(big.var()) - 1 / 3) < 0.01


def test_reset_known_answers(oracle):
    seed, off = 9, 5
    cfg = oracle.default_config(0, env_offset=off)
    v = oracle.OracleVec(8, seed=seed, cfg=cfg)
    v.reset(seed)
    st = v.get_state()
    key = py_stream_key(seed, 0)
    f32 = np.float32
    for i in range(8):
        u = [py_draw(key, off + i, 0, d) for d in range(5)]
        # nine 16-bit halves of five draws, low half first; (h - 32768) / 32768 is exact in float32
        halves = [(u[j // 2] >> 16) if j & 1 else (u[j // 2] & 0xFFFF) for j in range(9)]
        sym = [f32((h - 32768) / 32768.0) for h in halves]
        assert st["pos"][i].tolist() == [float(f32(cfg.spawn_extent) * s) for s in sym[0:3]]
        assert st["target"][i].tolist() == [float(f32(cfg.target_extent) * s) for s in sym[3:6]]
        t = [f32(cfg.tilt_init) * s for s in sym[6:9]]
        q = st["quat"][i].astype(np.float64)
        assert abs(np.linalg.norm(q) - 1) < 1e-6
        np.testing.assert_allclose(q[1:] / q[0], np.array(t, np.float64), rtol=1e-6)
    assert np.all(st["rpm"] == st["rpm"][0, 0]) and np.all(st["vel"] == 0) and np.all(st["tick"] == 0)
    hover = np.sqrt(0.027 * 9.81 / (4 * 3.16e-10))
    assert abs(st["rpm"][0, 0] - hover) / hover < 1e-6


GOLDEN_CASES = [("hover", 0, 1024, 0, {}), ("waypoint", 1, 1024, 0, {}), ("hover_h100_off", 0, 100, 1 << 20, {}),
                ("swarm", 2, 300, 64, {"collision_radius": 0.5}), ("race", 3, 200, 0, {"gate_radius": 2.5})]


@pytest.mark.parametrize("name,task,horizon,off,extra", GOLDEN_CASES)
def test_oracle_reproduces_golden_vectors(oracle, name, task, horizon, off, extra):
    check_golden(oracle, None, name, task, horizon, off, extra)


# Every build that honours the numerics contract (-ffp-contract=off, no fast-math: only the fmaf() calls written in the
# source fuse; / and sqrtf correctly rounded) must reproduce the committed vectors bit for bit — the vectors pin the
# SOURCE's arithmetic, not one compiler's code generation. gcc and the image's clang (AMD clang 22, the same front end
# hipcc uses for the kernels), each with and without optimisation, with and without hardware FMA / AVX code paths.
COMPILERS = [("gcc", ["-O0"]), ("gcc", ["-O2", "-ffp-contract=off", "-fno-fast-math"]), ("gcc", ["-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-fno-math-errno"]),
             ("/opt/rocm/lib/llvm/bin/clang", ["-O0", "-ffp-contract=off"]), ("/opt/rocm/lib/llvm/bin/clang", ["-O3", "-ffp-contract=off", "-fno-fast-math"]),
             ("/opt/rocm/lib/llvm/bin/clang", ["-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-fno-math-errno"])]


@pytest.fixture(scope="module", params=range(len(COMPILERS)), ids=[f"{os.path.basename(c)}{''.join(f)}" for c, f in COMPILERS])
def other_build(request, oracle, tmp_path_factory):
    cc, flags = COMPILERS[request.param]
    if not (os.path.exists(cc) or cc == "gcc"):
        pytest.skip(f"{cc} not in this image")
    return oracle.variant(cc, flags, str(tmp_path_factory.mktemp("cc") / f"liboracle_{request.param}.so"))


@pytest.mark.parametrize("name,task,horizon,off,extra", GOLDEN_CASES)
def test_golden_vectors_from_every_compiler(oracle, other_build, name, task, horizon, off, extra):
    """the oracle was only ever built with gcc -O3 -march=native."""
    check_golden(oracle, other_build, name, task, horizon, off, extra)


def check_golden(oracle, fns, name, task, horizon, off, extra):
    g = np.load(os.path.join(GOLD, f"golden_{name}.npz"))
    n, seed = 64, 20251017
    v = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(task, horizon=horizon, env_offset=off, **extra), threads=2, fns=fns)
    v.reset(seed)
    assert_state_equal(g["reset_state"], v.get_state(), "reset_state")
    assert_bits_equal(g["reset_obs"], v.observations, "reset_obs")
    terms = np.unpackbits(g["terminals"], axis=1)[:, :n]
    truncs = np.unpackbits(g["truncations"], axis=1)[:, :n]
    for t in range(1, 1001):
        v.fill_random_actions()
        if t == 1:
            assert_bits_equal(g["actions_step1"], v.actions, "actions_step1")
        v.step()
        assert_bits_equal(g["rewards"][t - 1], v.rewards, f"rewards@{t}")
        assert_bits_equal(terms[t - 1], v.terminals, f"terminals@{t}")
        assert_bits_equal(truncs[t - 1], v.truncations, f"truncations@{t}")
        if t in (1, 10, 100, 1000):
            assert_state_equal(g[f"state_{t}"], v.get_state(), f"state_{t}")
            assert_bits_equal(g[f"obs_{t}"], v.observations, f"obs_{t}")
    log = v.log()
    assert_bits_equal(g["log"], np.array([log[k] for k in ("perf", "score", "episode_return", "episode_length", "oob", "n")], np.float32), "log")


def _cfgdict(cfg):
    return {k: (float(v) if isinstance(v, float) else v) for k, v in cfg.as_dict().items()}


def test_dynamics_track_float64_numpy_statement(oracle):
    """40 steps from reset under random actions (no env leaves the box that fast):
    float32 oracle vs float64 numpy statement of SPEC.md, loose tolerance."""
    import spec_numpy as sn

    n, seed = 512, 31
    for task in (0,):
        cfg = oracle.default_config(task, substeps=2)
        v = oracle.OracleVec(n, seed=seed, cfg=cfg)
        v.reset(seed)
        c = sn.derived(_cfgdict(cfg))
        st = v.get_state()
        S = [st[f].astype(np.float64) for f in ("pos", "vel", "quat", "omega", "rpm")]
        tgt = st["target"].astype(np.float64)
        wind = np.zeros((n, 3))
        for t in range(40):
            v.fill_random_actions()
            a = v.actions.astype(np.float64)
            v.step()
            S = sn.step(c, S, a, wind)
            assert v.terminals.sum() == 0
            np.testing.assert_allclose(v.rewards, sn.hover_reward(c, S, tgt, a), rtol=2e-4, atol=2e-5)
        st = v.get_state()
        for f, ref, tol in zip(("pos", "vel", "quat", "omega", "rpm"), S, (2e-5, 2e-4, 2e-5, 2e-3, 1e-1)):
            np.testing.assert_allclose(st[f], ref, rtol=1e-4, atol=tol, err_msg=f)
        np.testing.assert_allclose(v.observations, sn.obs(c, S, tgt), rtol=1e-3, atol=2e-4)


def test_free_fall_is_exact_parabola(oracle):
    """Rotors stopped, no drag: z(t) = z0 - g t^2 / 2 (RK4 is exact for it)."""
    cfg = oracle.default_config(0, k_drag=0.0, bound=1000.0, motor_tau=1e9)
    v = oracle.OracleVec(4, seed=1, cfg=cfg)
    v.reset(1)
    rows = v.get_state()
    rows["rpm"][:] = 0
    rows["pos"][:] = (0, 0, 100)
    rows["quat"][:] = (1, 0, 0, 0)
    v.set_state(rows)
    v.actions[:] = -1.0
    for _ in range(100):
        v.step()
    st = v.get_state()
    t = 100 * 0.01
    np.testing.assert_allclose(st["pos"][:, 2], 100 - 0.5 * 9.81 * t * t, rtol=1e-5)
    np.testing.assert_allclose(st["vel"][:, 2], -9.81 * t, rtol=1e-5)
    assert np.all(st["pos"][:, :2] == 0) and np.all(st["omega"] == 0)


def test_hover_equilibrium_and_torque_signs(oracle):
    cfg = oracle.default_config(0, k_drag=0.0)
    v = oracle.OracleVec(4, seed=1, cfg=cfg)
    v.reset(1)
    rows = v.get_state()
    hover = float(rows["rpm"][0, 0])
    rows["pos"][:] = 0
    rows["quat"][:] = (1, 0, 0, 0)
    v.set_state(rows)
    a_hover = 2 * hover / 21702.0 - 1
    v.actions[:] = a_hover
    # env 1: left rotors (0,1) faster -> +roll (omega_x > 0); env 2: rear rotors (1,2) faster -> +pitch-axis torque (omega_y > 0)
    # env 3: rotors 0,2 faster -> +yaw torque
    v.actions[1] = (a_hover + 0.1, a_hover + 0.1, a_hover - 0.1, a_hover - 0.1)
    v.actions[2] = (a_hover - 0.1, a_hover + 0.1, a_hover + 0.1, a_hover - 0.1)
    v.actions[3] = (a_hover + 0.1, a_hover - 0.1, a_hover + 0.1, a_hover - 0.1)
    for _ in range(20):
        v.step()
    st = v.get_state()
    assert np.abs(st["pos"][0]).max() < 1e-4 and np.abs(st["vel"][0]).max() < 1e-3 and np.abs(st["omega"][0]).max() < 1e-4
    assert st["omega"][1, 0] > 0.1 and abs(st["omega"][1, 1]) < 1e-3
    assert st["omega"][2, 1] > 0.1 and abs(st["omega"][2, 0]) < 1e-3
    assert st["omega"][3, 2] > 0.01 and abs(st["omega"][3, 0]) < 1e-3
    # +roll about body x tips the thrust toward -y (right-hand rule): y acceleration negative
    assert st["vel"][1, 1] < 0 and st["vel"][2, 0] > 0
    np.testing.assert_allclose(np.linalg.norm(st["quat"].astype(np.float64), axis=1), 1.0, atol=1e-6)


def test_truncation_reset_and_log(oracle):
    n = 128
    v = oracle.OracleVec(n, seed=2, cfg=oracle.default_config(0, horizon=10))
    v.reset(2)
    first_obs = v.observations.copy()
    for t in range(10):
        v.actions[:] = 0.34  # ~hover
        v.step()
        if t < 9:
            assert v.truncations.sum() == 0
    assert v.truncations.sum() == n and v.terminals.sum() == 0
    st = v.get_state()
    assert np.all(st["tick"] == 0) and np.all(st["episode"] == 1) and np.all(st["n_sum"] == 1)
    assert not np.array_equal(first_obs, v.observations)  # first obs of the NEXT episode
    log = v.log()
    assert log["n"] == n and log["episode_length"] == 10 and log["oob"] == 0
    assert v.log()["n"] == 0  # drained


def test_rollout_outputs_definition(oracle):
    n, T = 64, 25
    a = oracle.OracleVec(n, seed=4, cfg=oracle.default_config(1, horizon=12))
    b = oracle.OracleVec(n, seed=4, cfg=oracle.default_config(1, horizon=12))
    a.reset(4)
    b.reset(4)
    a.rollout(T)
    rs = np.zeros(n, np.float32)
    anyt = np.zeros(n, np.uint8)
    for _ in range(T):
        b.fill_random_actions()
        b.step()
        rs = rs + b.rewards
        anyt |= b.truncations
    assert_state_equal(a.get_state(), b.get_state(), "rollout state")
    assert_bits_equal(a.rewards, rs, "reward sums")
    assert_bits_equal(a.truncations, anyt, "truncation any")
    assert_bits_equal(a.observations, b.observations, "final obs")
    assert a.gstep == b.gstep == T


def test_swarm_collisions_are_mutual_and_neighbour_obs(oracle):
    """SPEC.md §10: squared distances are bitwise symmetric, so collisions come in
    pairs (or larger clusters); neighbour observations follow the definition."""
    n, A = 512, 8
    cfg = oracle.default_config(2, agents_per_env=A, collision_radius=0.7, horizon=10_000)
    v = oracle.OracleVec(n, seed=3, cfg=cfg)
    v.reset(3)
    st = v.get_state()
    pos = st["pos"].astype(np.float64).reshape(n // A, A, 3)
    d = np.linalg.norm(pos[:, :, None, :] - pos[:, None, :, :], axis=-1) + np.eye(A) * 1e9
    np.testing.assert_allclose(v.observations[:, 23].reshape(n // A, A), (d.min(-1) / 5.0) ** 2, rtol=1e-5)
    collisions = 0
    for t in range(150):
        v.fill_random_actions()
        before = v.get_state()["episode"].copy()
        v.step()
        st = v.get_state()
        crashed = v.terminals.astype(bool).reshape(n // A, A)
        for g in np.flatnonzero(crashed.any(1)):
            # an agent that crashed without leaving the box must have a crashed partner in its swarm
            collisions += int(crashed[g].sum() >= 2)
        assert np.all((st["episode"] - before) == (v.terminals | v.truncations))
    assert collisions > 0
    assert np.isfinite(v.observations).all()


def test_swarm_of_one_is_hover_plus_far_neighbour(oracle):
    a = oracle.OracleVec(64, seed=5, cfg=oracle.default_config(2, agents_per_env=1, c_proximity=0.7))
    b = oracle.OracleVec(64, seed=5, cfg=oracle.default_config(0))
    a.reset(5)
    b.reset(5)
    for t in range(200):
        a.fill_random_actions()
        b.fill_random_actions()
        a.step()
        b.step()
        assert_bits_equal(a.rewards, b.rewards, f"rewards {t}")  # proximity term is exactly zero beyond the range
    assert_state_equal(a.get_state(), b.get_state(), "A=1 swarm == hover")
    assert_bits_equal(a.observations[:, :20].copy(), b.observations, "first 20 obs")
    assert np.all(a.observations[:, 20:23] == 0) and np.all(a.observations[:, 23] == np.float32(16.0))


def test_swarm_rejects_bad_grouping(oracle):
    with pytest.raises(RuntimeError):
        oracle.OracleVec(100, cfg=oracle.default_config(2, agents_per_env=8))   # 100 % 8 != 0
    with pytest.raises(RuntimeError):
        oracle.OracleVec(64, cfg=oracle.default_config(2, agents_per_env=6))    # not a power of two
    with pytest.raises(RuntimeError):
        oracle.OracleVec(64, cfg=oracle.default_config(2, agents_per_env=8, env_offset=4))


def _toward_gates(v, offset, speed, lateral=0.0):
    """Place every drone `offset` metres behind its gate plane (plus `lateral` metres off-axis), flying along +n."""
    st = v.get_state()
    n, c = st["wind"].copy(), st["target"].copy()
    side = np.cross(n, np.array([0.3, -0.5, 0.8], np.float32))
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    st["pos"] = c - offset * n + lateral * side
    st["vel"] = speed * n
    st["omega"] = 0
    v.set_state(st)
    return st


def test_race_gate_passing(oracle):
    """SPEC.md §11: forward crossing inside the ring scores and deals the next gate;
    a crossing outside the ring or backwards does not."""
    n = 256
    cfg = oracle.default_config(3, gate_radius=0.75, bound=50.0, horizon=10_000)
    v = oracle.OracleVec(n, seed=12, cfg=cfg)
    v.reset(12)
    st0 = v.get_state()
    np.testing.assert_allclose(np.linalg.norm(st0["wind"].astype(np.float64), axis=1), 1.0, atol=1e-6)
    # through the middle
    before = _toward_gates(v, 0.03, 6.0)
    v.actions[:] = 0.3
    v.step()
    st = v.get_state()
    assert np.all(st["score_count"] == 1) and np.all(v.rewards > 0.9)
    assert not np.array_equal(st["target"], before["target"])
    np.testing.assert_allclose(np.linalg.norm(st["wind"].astype(np.float64), axis=1), 1.0, atol=1e-6)
    want_n = (st["target"] - before["target"]).astype(np.float64)
    want_n /= np.linalg.norm(want_n, axis=1, keepdims=True)
    np.testing.assert_allclose(st["wind"], want_n, atol=1e-6)
    # signed plane distance of the NEW gate in obs[23]
    d = ((st["pos"] - st["target"]).astype(np.float64) * st["wind"]).sum(1) / 50.0
    np.testing.assert_allclose(v.observations[:, 23], d, atol=1e-6)
    # outside the ring: a miss
    _toward_gates(v, 0.03, 6.0, lateral=1.0)
    v.step()
    assert np.all(v.get_state()["score_count"] == 1)
    # backwards through the ring: nothing
    _toward_gates(v, -0.03, -6.0)
    v.step()
    assert np.all(v.get_state()["score_count"] == 1) and v.terminals.sum() == 0

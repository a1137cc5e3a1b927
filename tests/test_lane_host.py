"""The kernel's per-lane math (drone_amd/csrc/drone_lane.hpp), compiled for the
host by g++ as a TEST-ONLY harness, against the scalar oracle — bit for bit.
Lets the numerics contract (explicit fma only, correctly rounded / and sqrt) be
checked here without a GPU; the -m gpu tests then check the same code as built
by hipcc for gfx950."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from drone_amd import abi
from helpers import assert_bits_equal, assert_state_equal

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "lane_host", "lane_host.cpp")
SO = os.path.join(HERE, "lane_host", "liblane_host.so")


# Both forms of the RK4 substep: scalar, and the two-wide one (drone_pk.hpp; on the host its packed operations are the
# same expression trees as plain scalar code, so these tests pin the trees — pair layout, half selection, the carried
# sign of dq0 — against the oracle before the gfx950 build is ever run).
@pytest.fixture(scope="module", params=[0, 1], ids=["scalar_rk4", "packed_rk4"])
def lane(request):
    so = SO.replace(".so", f"_pk{request.param}.so")
    subprocess.run(["g++", "-O2", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-fPIC", "-shared",
                    f"-DDRONE_PK_DEFAULT={request.param}", "-o", so, SRC], check=True)
    return C.CDLL(so)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("task,horizon,substeps", [(0, 1024, 1), (1, 1024, 1), (0, 37, 1), (1, 41, 2), (2, 1024, 1), (2, 60, 2), (3, 1024, 1), (3, 50, 3)])
def test_lane_math_bit_exact_vs_oracle(lane, oracle, task, horizon, substeps):
    n, seed = 1024, 4242
    extra = dict(collision_radius=0.6, agents_per_env=16, env_offset=784) if task == 2 else dict(env_offset=777)
    cfg = oracle.default_config(task, horizon=horizon, substeps=substeps, **extra)
    v = oracle.OracleVec(n, seed=seed, cfg=cfg, threads=4)
    v.reset(seed)
    rows = np.zeros(n, dtype=abi.state_row_dtype())
    obs = np.zeros((n, abi.obs_dim(task)), np.float32)
    act = np.zeros((n, 4), np.float32)
    rew = np.zeros(n, np.float32)
    term = np.zeros(n, np.uint8)
    trunc = np.zeros(n, np.uint8)
    lane.lane_host_reset(C.byref(cfg), C.c_uint64(seed), p(rows), p(obs), n)
    assert_state_equal(v.get_state(), rows, "reset")
    assert_bits_equal(v.observations, obs, "reset obs")
    for t in range(400):
        v.fill_random_actions()
        g = v.gstep
        v.step()
        lane.lane_host_step(C.byref(cfg), C.c_uint64(seed), C.c_uint32(g), p(rows), p(act), p(obs), p(rew), p(term), p(trunc), n, 1)
        assert_bits_equal(v.actions, act, f"policy actions {t}")
        assert_bits_equal(v.observations, obs, f"obs {t}")
        assert_bits_equal(v.rewards, rew, f"rew {t}")
        assert_bits_equal(v.terminals, term, f"term {t}")
        assert_bits_equal(v.truncations, trunc, f"trunc {t}")
    assert_state_equal(v.get_state(), rows, "final state")
    assert rows["episode"].sum() > 0


@pytest.mark.parametrize("task,horizon,substeps", [(0, 40, 1), (1, 33, 2), (2, 50, 1), (3, 45, 3)])
def test_carried_rotor_inputs_equal_recomputed_ones(lane, oracle, task, horizon, substeps):
    """The register-resident kernels (fused rollout, step_many) carry the rotor inputs from step to step and through
    resets instead of recomputing them from the rotor speeds (round 3: 18 operations per substep). Same function of the
    same floats: the carried form must give the oracle's rollout bit for bit, over hundreds of resets."""
    n, seed = 768, 2718
    extra = dict(collision_radius=0.6, agents_per_env=16, env_offset=784) if task == 2 else dict(env_offset=777)
    cfg = oracle.default_config(task, horizon=horizon, substeps=substeps, **extra)
    v = oracle.OracleVec(n, seed=seed, cfg=cfg, threads=4)
    v.reset(seed)
    rows = v.get_state()
    rsum = np.zeros(n, np.float32)
    for T in (1, 128, 37, 300):
        g = v.gstep
        v.rollout(T)
        lane.lane_host_rollout(C.byref(cfg), C.c_uint64(seed), C.c_uint32(g), T, p(rows), p(rsum), n)
        assert_state_equal(v.get_state(), rows, f"carried rollout T={T}")
        assert_bits_equal(v.rewards, rsum, f"reward sums T={T}")
    assert rows["episode"].sum() > 4 * n


def test_fused_s16_and_sym_equal_the_specs_literal_forms(lane):
    """SPEC.md §2: s16(h) = (h - 32768) 2^-15, sym(u) = 2 (u >> 8) 2^-24 - 1. The product computes each as one fused
    multiply-add on the converted integer; all intermediates are exact either way, so they must agree on every input."""
    got = np.zeros(65536, np.float32)
    lane.lane_host_s16_all(p(got))
    h = np.arange(65536, dtype=np.int64)
    want = ((h - 32768).astype(np.float32) * np.float32(2.0 ** -15)).astype(np.float32)
    assert_bits_equal(want, got, "s16 over all 2^16 halves")
    rng = np.random.default_rng(1)
    u = np.concatenate([rng.integers(0, 2**32, 1 << 20, dtype=np.uint64).astype(np.uint32), np.array([0, 255, 256, 0x7FFFFFFF, 0x80000000, 0x800000FF, 0xFFFFFFFF], np.uint32)])
    out = np.zeros(len(u), np.float32)
    lane.lane_host_sym(p(u), p(out), len(u))
    m = (u >> 8).astype(np.float32)
    want = (np.float32(2.0) * (m * np.float32(2.0 ** -24)) - np.float32(1.0)).astype(np.float32)  # 2 m 2^-24 and the subtraction are exact in float32
    assert_bits_equal(want, out, "sym")


def test_kparams_match_oracle_params(lane, oracle):
    cfg = oracle.default_config(1, substeps=3, dt=0.02)
    want = oracle.params(cfg)
    kp = np.zeros(57, np.uint32)
    lane.lane_host_kparams(C.byref(cfg), C.c_uint64(9), p(kp))
    f = kp.view(np.float32)
    # oracle Params order -> KParams word index
    idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 21, 25, 26, 27, 28, 29, 36, 37, 44, 45, 46, 48]
    assert_bits_equal(want, f[idx].copy(), "derived params")
    keys = [oracle.lib().oracle_stream_key(9, s) for s in range(4)]
    assert list(kp[51:55]) == keys


def test_random_configs_bit_exact(lane, oracle):
    """Twenty random (but physical) configurations — masses, inertias, step sizes,
    substeps, bounds, wind and reward weights — lane math vs oracle, bit for bit."""
    rng = np.random.default_rng(123)
    n = 256
    for trial in range(20):
        task = trial % 4
        over = dict(
            horizon=int(rng.integers(5, 120)), substeps=int(rng.integers(1, 4)), dt=float(rng.uniform(0.002, 0.03)),
            mass=float(rng.uniform(0.02, 1.5)), arm=float(rng.uniform(0.03, 0.3)),
            ixx=float(rng.uniform(1e-5, 1e-2)), iyy=float(rng.uniform(1e-5, 1e-2)), izz=float(rng.uniform(2e-5, 2e-2)),
            k_thrust=float(rng.uniform(1e-10, 1e-7)), k_torque=float(rng.uniform(1e-12, 1e-9)),
            k_drag=float(rng.uniform(0, 0.05)), k_ang_damp=float(rng.uniform(0, 1e-4)), gravity=float(rng.uniform(1.6, 12)),
            max_rpm=float(rng.uniform(5000, 30000)), motor_tau=float(rng.uniform(0.01, 0.2)),
            max_vel=float(rng.uniform(5, 40)), max_omega=float(rng.uniform(10, 80)), bound=float(rng.uniform(2, 20)),
            spawn_extent=float(rng.uniform(0.5, 2)), target_extent=float(rng.uniform(0.5, 2)), tilt_init=float(rng.uniform(0, 0.5)),
            hover_radius=float(rng.uniform(0.1, 2)), waypoint_radius=float(rng.uniform(0.1, 3)),
            wind_theta=float(rng.uniform(0, 2)), wind_sigma=float(rng.uniform(0, 3)), wind_max=float(rng.uniform(1, 8)),
            c_omega=float(rng.uniform(0, 1e-3)), c_action=float(rng.uniform(0, 0.1)), crash_penalty=float(rng.uniform(0, 5)),
            progress_scale=float(rng.uniform(0.1, 3)), waypoint_bonus=float(rng.uniform(0, 3)),
            env_offset=int(rng.integers(0, 2**25)) * 64, agents_per_env=int(2 ** rng.integers(0, 7)),
            collision_radius=float(rng.uniform(0.05, 1.0)), proximity_radius=float(rng.uniform(0.3, 3.0)), c_proximity=float(rng.uniform(0, 2)), gate_radius=float(rng.uniform(0.2, 3.0)))
        seed = int(rng.integers(0, 2**63))
        cfg = oracle.default_config(task, **over)
        v = oracle.OracleVec(n, seed=seed, cfg=cfg)
        v.reset(seed)
        rows = np.zeros(n, dtype=abi.state_row_dtype())
        obs = np.zeros((n, abi.obs_dim(task)), np.float32)
        act = np.zeros((n, 4), np.float32)
        rew = np.zeros(n, np.float32)
        term = np.zeros(n, np.uint8)
        trunc = np.zeros(n, np.uint8)
        lane.lane_host_reset(C.byref(cfg), C.c_uint64(seed), p(rows), p(obs), n)
        assert_state_equal(v.get_state(), rows, f"trial {trial} reset")
        for t in range(150):
            v.fill_random_actions()
            g = v.gstep
            v.step()
            lane.lane_host_step(C.byref(cfg), C.c_uint64(seed), C.c_uint32(g), p(rows), p(act), p(obs), p(rew), p(term), p(trunc), n, 1)
            assert_bits_equal(v.observations, obs, f"trial {trial} obs {t}")
            assert_bits_equal(v.rewards, rew, f"trial {trial} rew {t}")
        assert_state_equal(v.get_state(), rows, f"trial {trial} final")


def test_race_forced_gate_passes(lane, oracle):
    """Gate passes are rare under the random policy: steer every drone through its gate
    (and some past it) and compare lane math with the oracle on exactly those steps."""
    n, seed = 512, 99
    cfg = oracle.default_config(3, gate_radius=1.0, horizon=500)
    v = oracle.OracleVec(n, seed=seed, cfg=cfg)
    v.reset(seed)
    obs = np.zeros((n, 24), np.float32)
    act = np.zeros((n, 4), np.float32)
    rew = np.zeros(n, np.float32)
    term = np.zeros(n, np.uint8)
    trunc = np.zeros(n, np.uint8)
    rng = np.random.default_rng(5)
    passes = 0
    for rnd in range(12):
        st = v.get_state()
        nrm, c = st["wind"], st["target"]
        side = np.cross(nrm, np.array([0.3, -0.5, 0.8], np.float32)).astype(np.float32)
        st["pos"] = (c - rng.uniform(0.0, 0.08, (n, 1)).astype(np.float32) * nrm + rng.uniform(0, 1.6, (n, 1)).astype(np.float32) * side).astype(np.float32)
        st["vel"] = (rng.uniform(2, 8, (n, 1)).astype(np.float32) * nrm).astype(np.float32)
        v.set_state(st)
        rows = v.get_state()
        before = rows["score_count"].copy()
        for t in range(3):
            v.fill_random_actions()
            g = v.gstep
            v.step()
            lane.lane_host_step(C.byref(cfg), C.c_uint64(seed), C.c_uint32(g), p(rows), p(act), p(obs), p(rew), p(term), p(trunc), n, 1)
            assert_bits_equal(v.observations, obs, f"obs round {rnd} step {t}")
            assert_bits_equal(v.rewards, rew, f"rew round {rnd} step {t}")
        assert_state_equal(v.get_state(), rows, f"round {rnd}")
        passes += int((rows["score_count"] > before).sum())
    assert passes > n  # both outcomes occur: passes and misses
    assert passes < 12 * n

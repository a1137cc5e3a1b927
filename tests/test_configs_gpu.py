"""BASELINE.json's configs at their full sizes. The oracle cannot run 2^20 envs
x 1000 steps in seconds, but envs are independent and keyed by their GLOBAL id,
so the oracle replays sampled blocks of envs (via env_offset) and must match
the corresponding slices of the full-size HIP run bit for bit; the rest is
covered by size-independent properties (shard invariance, fused == stepped,
run-to-run determinism, checksum of checksums)."""
import zlib

import numpy as np
import pytest

from helpers import assert_bits_equal, assert_state_equal, to_np

pytestmark = pytest.mark.gpu


def sampled_blocks(n, width=256):
    starts = sorted({0, 255, n // 3, n // 2 - width // 2, n - width})
    return [(s, min(width, n - s)) for s in starts if s >= 0]


def check_against_sampled_oracle(oracle, h, task, seed, steps, base_offset=0, fused=0, **over):
    n = h.num_envs
    for start, count in sampled_blocks(n):
        o = oracle.OracleVec(count, seed=seed, cfg=oracle.default_config(task, env_offset=base_offset + start, **over), threads=4)
        o.reset(seed)
        for _ in range(steps):
            o.fill_random_actions()
            o.step()
        if fused:
            o.rollout(fused)
        assert_state_equal(o.get_state(), h.get_state(start, count), f"envs [{start},{start + count})")
        assert_bits_equal(o.observations, to_np(h.observations)[start:start + count], f"obs [{start},{start + count})")
        assert_bits_equal(o.rewards, to_np(h.rewards)[start:start + count], f"rew [{start},{start + count})")
        assert_bits_equal(o.terminals, to_np(h.terminals)[start:start + count], f"term [{start},{start + count})")


def run_steps(h, steps):
    for _ in range(steps):
        h.fill_random_actions()
        h.step()


def test_config2_65536_hover_1000_steps(oracle, hip):
    """configs[1]: 65 536 envs, hover, 1000-step random-action rollout — full oracle run."""
    n, seed = 65536, 0
    h = hip.DroneVec(n, seed=seed, task=0, device="cuda:0")
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(0), threads=16)
    h.reset(seed)
    o.reset(seed)
    for t in range(1000):
        h.fill_random_actions()
        h.step()
        o.fill_random_actions()
        o.step()
    so, sh = o.get_state(), h.get_state()
    assert_state_equal(so, sh, "config 2 state after 1000 steps")
    assert_bits_equal(o.observations, h.observations, "config 2 obs")
    worst = max(float(np.max(np.abs(so[f].astype(np.float64) - sh[f].astype(np.float64)))) for f in ("pos", "vel", "quat", "omega", "rpm"))
    assert worst == 0.0  # north-star: <= 1e-5 relative over 1000 steps
    lo, lh = o.log(), h.log()
    assert lo["n"] == lh["n"] > 0
    assert lh["episode_return"] == pytest.approx(lo["episode_return"], rel=1e-6)


def test_config3_shard_of_2pow20(oracle, hip):
    """configs[2]: 2^20 envs sharded 131 072 per GPU — one rank's shard (rank 5) with its global offset."""
    n, rank, seed = 131072, 5, 3
    h = hip.DroneVec(n, seed=seed, task=0, device="cuda:0", env_offset=rank * n)
    h.reset(seed)
    run_steps(h, 300)
    check_against_sampled_oracle(oracle, h, 0, seed, 300, base_offset=rank * n)


def test_config4_262144_waypoint_wind(oracle, hip):
    """configs[3]: 262 144 envs, waypoint tracking with wind."""
    n, seed = 262144, 11
    h = hip.DroneVec(n, seed=seed, task=1, device="cuda:0")
    h.reset(seed)
    run_steps(h, 400)
    check_against_sampled_oracle(oracle, h, 1, seed, 400)
    assert h.get_state(0, 4096)["score_count"].sum() >= 0


def test_config5_fused_128_at_2pow20(oracle, hip):
    """configs[4]: 2^20 envs, fused 128-step rollout with the device policy — on one GPU here."""
    n, seed = 1 << 20, 17
    h = hip.DroneVec(n, seed=seed, task=0, device="cuda:0")
    h.reset(seed)
    h.rollout(128)
    h.rollout(128)
    # oracle: two fused windows on sampled blocks (reward sums are per window: compare the last)
    for start, count in sampled_blocks(n):
        o = oracle.OracleVec(count, seed=seed, cfg=oracle.default_config(0, env_offset=start), threads=4)
        o.reset(seed)
        o.rollout(128)
        o.rollout(128)
        assert_state_equal(o.get_state(), h.get_state(start, count), f"fused envs [{start},{start + count})")
        assert_bits_equal(o.rewards, to_np(h.rewards)[start:start + count], "fused reward sums")
        assert_bits_equal(o.truncations, to_np(h.truncations)[start:start + count], "fused truncation flags")
        assert_bits_equal(o.observations, to_np(h.observations)[start:start + count], "fused obs")


def _digest(h):
    st = h.get_state()
    crcs = [zlib.crc32(np.ascontiguousarray(st[f]).tobytes()) for f in st.dtype.names]
    crcs.append(zlib.crc32(np.ascontiguousarray(to_np(h.observations)).tobytes()))
    return zlib.crc32(np.array(crcs, np.uint32).tobytes())


def test_full_size_properties_2pow20(hip):
    """2^20 envs: determinism, fused == stepped, two half shards == whole — by checksum of checksums."""
    n, seed = 1 << 20, 23

    def fresh(count=n, off=0):
        v = hip.DroneVec(count, seed=seed, task=1, device="cuda:0", env_offset=off, horizon=200)
        v.reset(seed)
        return v

    a = fresh()
    run_steps(a, 64)
    da = _digest(a)
    b = fresh()
    b.rollout(64)
    assert _digest(b) == da, "fused rollout != 64 single steps at 2^20"
    c = fresh()
    run_steps(c, 64)
    assert _digest(c) == da, "not deterministic run to run"
    sa = a.get_state()
    lo, hi = fresh(n // 2, 0), fresh(n // 2, n // 2)
    lo.rollout(64)
    hi.rollout(64)
    assert_state_equal(sa[: n // 2], lo.get_state(), "lower half shard")
    assert_state_equal(sa[n // 2:], hi.get_state(), "upper half shard")


def test_eight_million_envs_index_arithmetic(oracle, hip):
    """2^23 envs (8x the metric's size; 1.2 GB of state planes): 32-bit index and
    byte-offset arithmetic in the kernels, checked on sampled blocks incl. the last."""
    n, seed = 1 << 23, 29
    h = hip.DroneVec(n, seed=seed, task=1, device="cuda:0", horizon=20)
    h.reset(seed)
    run_steps(h, 12)
    h.rollout(30)
    check_against_sampled_oracle(oracle, h, 1, seed, 12, fused=30, horizon=20)


@pytest.mark.parametrize("task", [0, 1, 2])
def test_soak_20000_steps(oracle, hip, task):
    """4096 envs x 20 000 steps (≈170 episodes per env) in fused windows of 1000:
    state, log sums and outputs stay bit-identical to the oracle the whole way."""
    n, seed = 4096, 2718
    over = dict(horizon=257, collision_radius=0.5) if task == 2 else dict(horizon=257)
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(task, **over), threads=16)
    h = hip.DroneVec(n, seed=seed, cfg=hip.default_config(task, **over), device="cuda:0")
    o.reset(seed)
    h.reset(seed)
    for w in range(20):
        o.rollout(1000)
        h.rollout(1000)
        assert_state_equal(o.get_state(), h.get_state(), f"task {task} window {w}")
        assert_bits_equal(o.rewards, h.rewards, f"task {task} window {w} reward sums")
    assert_bits_equal(o.observations, h.observations, "final obs")
    assert o.gstep == h.gstep == 20000
    st = o.get_state()
    assert st["episode"].min() > 50
    lo, lh = o.log(), h.log()
    assert lo["n"] == lh["n"]
    for k in lo:
        assert lh[k] == pytest.approx(lo[k], rel=1e-6, abs=1e-7), k

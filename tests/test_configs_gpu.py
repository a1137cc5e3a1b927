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


def sampled_blocks(n, width=256, align=1):
    """Five blocks of `width` envs: the first, one that straddles a workgroup boundary, two inside, the last.
    `align`: block starts are multiples of it (swarm task: whole swarms)."""
    starts = sorted({0, 255, n // 3, n // 2 - width // 2, n - width})
    return [(s // align * align, min(width, n - s // align * align)) for s in starts if s >= 0]


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


# =====================================================================
# Round 4: every handle bench.py times, with the library's OWN choices for that size (state layout,
# sweep order, line widening, action-load hint, packed RK4), driven through exactly the paths the bench times on it, and
# compared with the oracle on sampled blocks. The list is bench.py's (TIMED_HANDLES), not a copy.
# =====================================================================
import bench  # noqa: E402  (module level of bench.py imports nothing heavy)

TASK_IDS = {"hover": 0, "waypoint": 1, "swarm": 2, "race": 3}
# what the library is expected to pick for the sizes the headline rests on; a changed heuristic must be noticed here
EXPECTED_VARIANT = {
    ("hover", 1 << 20): dict(task=0, compact=0, mem=0, dt=1, order=1, line_complete=0, packed_rk4=0, bytes=262),
    ("hover", 1 << 22): dict(task=0, compact=0, mem=2, dt=1, order=8, line_complete=1, packed_rk4=0, bytes=262),   # non-temporal STATE loads, plain sweep
    ("hover", 1 << 23): dict(task=0, compact=0, mem=1, dt=1, order=6, line_complete=1, packed_rk4=0, bytes=262),   # reversed sweep, streamed action rows
    ("hover", 65536): dict(task=0, compact=0, mem=0, dt=0, order=1, line_complete=0, packed_rk4=1, bytes=278),
    ("hover", 131072): dict(task=0, compact=0, mem=0, dt=0, order=1, line_complete=0, packed_rk4=0, bytes=278),
    ("hover", 1 << 19): dict(task=0, compact=0, mem=0, dt=1, order=1, line_complete=0, packed_rk4=0, bytes=262),
    ("waypoint", 262144): dict(task=1, compact=0, mem=0, dt=0, order=1, line_complete=0, packed_rk4=0, bytes=310),
}


def drive_and_compare(oracle, hip, task, n, paths, seed, steps, env_offset=0, **over):
    """reset -> `steps` per-step launches (ring of two action buffers, rebound like bench.py does) -> fused 128-step
    rollout -> step_many K = 8 / 32 with caller-staged actions, whichever of them `paths` names; the oracle replays the
    same calls on five sampled 256-env blocks (global env ids) and every output and the state must match bit for bit."""
    import torch

    h = hip.DroneVec(n, seed=seed, cfg=hip.default_config(task, env_offset=env_offset, **over), device="cuda:0")
    align = int(over.get("agents_per_env", 8)) if task == 2 else 1
    blocks = sampled_blocks(n, align=align)
    orc = []
    for start, count in blocks:
        o = oracle.OracleVec(count, seed=seed, cfg=oracle.default_config(task, env_offset=env_offset + start, **over), threads=4)
        o.reset(seed)
        orc.append(o)
    h.reset(seed)
    text, var = h.variant

    def compare(what, obs, rew, term, trunc):
        for (start, count), o in zip(blocks, orc):
            sl = slice(start, start + count)
            assert_bits_equal(o.observations, to_np(obs[sl]), f"{text}: {what}: obs [{start},{start + count})")
            assert_bits_equal(o.rewards, to_np(rew[sl]), f"{text}: {what}: rewards")
            assert_bits_equal(o.terminals, to_np(term[sl]), f"{text}: {what}: terminals")
            assert_bits_equal(o.truncations, to_np(trunc[sl]), f"{text}: {what}: truncations")
            assert_state_equal(o.get_state(), h.get_state(start, count), f"{text}: {what}: state [{start},{start + count})")

    if "step" in paths:
        ring = [torch.empty_like(h.actions) for _ in range(2)]
        for t in range(steps):
            a = ring[t & 1]
            h.fill_random_actions(out=a)  # the step counter's policy draw, into the buffer about to be bound
            h.bind_actions(a)
            h.step()
        for o in orc:
            for _ in range(steps):
                o.fill_random_actions()
                o.step()
        compare(f"{steps} per-step launches", h.observations, h.rewards, h.terminals, h.truncations)
    if "rollout" in paths:
        h.rollout(128)
        for o in orc:
            o.rollout(128)
        compare("fused 128-step rollout", h.observations, h.rewards, h.terminals, h.truncations)
    for K in (8, 32):
        if f"many{K}" not in paths:
            continue
        bufs = h.alloc_step_many(K)
        g0 = h.gstep
        for k in range(K):
            h.fill_random_actions(gstep=g0 + k, out=bufs.actions[k])
        h.step_many(bufs)
        torch.cuda.synchronize()
        want = [o.step_many(K, None) for o in orc]  # the policy's draws for steps g0 .. g0 + K - 1 = the staged blocks
        for (start, count), o, (wo, wr, wt, wu, _) in zip(blocks, orc, want):
            sl = slice(start, start + count)
            assert_bits_equal(wo, to_np(bufs.observations[:, sl]), f"{text}: step_many K={K}: obs")
            assert_bits_equal(wr, to_np(bufs.rewards[:, sl]), f"{text}: step_many K={K}: rewards")
            assert_bits_equal(wt, to_np(bufs.terminals[:, sl]), f"{text}: step_many K={K}: terminals")
            assert_bits_equal(wu, to_np(bufs.truncations[:, sl]), f"{text}: step_many K={K}: truncations")
            assert_state_equal(o.get_state(), h.get_state(start, count), f"{text}: step_many K={K}: state")
        del bufs
    ended = sum(int(o.get_state()["episode"].sum()) for o in orc)
    h.close()
    return var, ended


@pytest.mark.parametrize("task_name,n,paths", bench.TIMED_HANDLES, ids=[f"{t}-{n}" for t, n, _ in bench.TIMED_HANDLES])
def test_every_handle_bench_times_matches_the_oracle(oracle, hip, task_name, n, paths):
    steps = 300 if n <= (1 << 20) else 100 if n <= (1 << 22) else 40
    # beyond 2^20 envs the run is too short for crashes under the default horizon: a short one brings the episode-end
    # path (log-plane read-modify-write widened to whole lines at these footprints) into the comparison; the kernel
    # instantiation does not depend on it
    over = dict(horizon=25) if n > (1 << 20) else {}
    var, ended = drive_and_compare(oracle, hip, TASK_IDS[task_name], n, paths, seed=41, steps=steps, **over)
    want = EXPECTED_VARIANT.get((task_name, n))
    # the footprint table's entry, exactly (round 6: the online measurement of round 5 re-derived the table in 12 of 12 logged cases
    # and is opt-in now, DRONE_AUTOTUNE=1 — tests/test_robustness_gpu.py follows a measuring handle)
    if want:
        assert var == want, f"a {task_name} handle of {n} envs now picks {var}; bench.py's figures and DESIGN.md assume {want}"
    assert ended > 0, "no episode ended in the sampled blocks: the episode-end path went unchecked"


def test_bench_shards_with_their_global_offsets(oracle, hip):
    """N = 8: rank 5's 131 072-env shard of the 2^20 run, per-step + fused, with its env_offset (the RNG is keyed on the
    global id; the shard's own variant choices)."""
    drive_and_compare(oracle, hip, 0, 131072, ("step", "rollout"), seed=5, steps=200, env_offset=5 * 131072)


def test_swarm_at_bench_size_derived_target_by_default(oracle, hip):
    """Swarm task (bench.py --task swarm) at 2^19 envs: the derived-target layout is the default there, collisions on
    (collision_radius 0.5: a few per cent of the swarms collide within the run), per-step + fused rollout + step_many."""
    var, ended = drive_and_compare(oracle, hip, 2, 1 << 19, ("step", "rollout", "many8"), seed=77, steps=150, agents_per_env=8, collision_radius=0.5, horizon=120)
    assert var["dt"] == 1 and var["bytes"] == 278 and ended > 0


def test_swarm_2pow20_whole_lines(oracle, hip):
    """Swarm at 2^20 envs (bench.py --task swarm default size): sweep order / line widening as chosen for that footprint."""
    var, _ = drive_and_compare(oracle, hip, 2, 1 << 20, ("step",), seed=78, steps=100, agents_per_env=8, collision_radius=0.5, horizon=60)
    assert var["dt"] == 1


def test_race_at_2pow20(oracle, hip):
    """Race task at 2^20 envs (bench.py --task race): seven planes, gate passes dealt during the episode."""
    var, ended = drive_and_compare(oracle, hip, 3, 1 << 20, ("step", "rollout"), seed=79, steps=200, gate_radius=2.5, horizon=150)
    assert var["bytes"] == 310 and ended > 0


def test_step_many_k8_at_2pow20(oracle, hip):
    """drone_vec_step_many at the metric's size (bench.py --mode many): derived-target layout, K = 8, caller-staged actions."""
    var, _ = drive_and_compare(oracle, hip, 0, 1 << 20, ("many8",), seed=80, steps=0)
    assert var["dt"] == 1


def test_explicit_state_layout_is_honoured_and_round_trips(oracle, hip, tmp_path):
    """DroneConfig.state_layout: the layout as a declared choice. A small handle forced into the derived-target
    layout and a large one forced to keep the target plane run the same trajectories; state rows and checkpoint files
    cross between the two layouts."""
    from drone_amd import abi

    n, seed = 5000, 12
    a = hip.DroneVec(n, seed=seed, cfg=hip.default_config(0, horizon=40, state_layout=abi.LAYOUT_DERIVED_TARGET), device="cuda:0")
    b = hip.DroneVec(n, seed=seed, cfg=hip.default_config(0, horizon=40, state_layout=abi.LAYOUT_TARGET_PLANE), device="cuda:0")
    assert a.variant[1]["dt"] == 1 and a.bytes_per_env_step == 262 and b.variant[1]["dt"] == 0 and b.bytes_per_env_step == 278
    for v in (a, b):
        v.reset(seed)
        for _ in range(70):
            v.fill_random_actions()
            v.step()
    assert_state_equal(a.get_state(), b.get_state(), "derived-target vs target-plane after 70 steps")
    assert_bits_equal(a.observations, b.observations, "observations")
    # rows cross the layouts both ways
    rows = a.get_state()
    b.set_state(rows)
    a.set_state(b.get_state())
    # checkpoint written by one layout, resumed by the other (state_layout is not part of the file's config check)
    path = str(tmp_path / "dt.npz")
    a.save_checkpoint(path)
    c = hip.DroneVec(n, seed=seed, cfg=hip.default_config(0, horizon=40, state_layout=abi.LAYOUT_TARGET_PLANE), device="cuda:0")
    c.reset(seed)
    c.load_checkpoint(path)
    for v in (a, c):
        for _ in range(50):
            v.fill_random_actions()
            v.step()
    assert_state_equal(a.get_state(), c.get_state(), "resumed in the other layout")
    assert_bits_equal(a.observations, c.observations, "observations after the resume")
    # a big handle told to keep the plane does; one that cannot hold the layout refuses it loudly
    big = hip.DroneVec(1 << 19, seed=seed, cfg=hip.default_config(0, state_layout=abi.LAYOUT_TARGET_PLANE), device="cuda:0")
    assert big.variant[1]["dt"] == 0 and big.bytes_per_env_step == 278
    big.close()
    with pytest.raises(RuntimeError, match="state_layout"):
        hip.DroneVec(512, seed=0, cfg=hip.default_config(1, state_layout=abi.LAYOUT_DERIVED_TARGET), device="cuda:0")
    with pytest.raises(RuntimeError, match="state_layout"):
        hip.DroneVec(512, seed=0, cfg=hip.default_config(0, horizon=70000, state_layout=abi.LAYOUT_DERIVED_TARGET), device="cuda:0")

"""Parity tests proper: the HIP path, called through the C-ABI, against the CPU
oracle on the same seeds and action sequences. Bit-exact (tolerance 0) for
state, observations, rewards and flags; the north-star's bound is 1e-5
relative over 1000 steps. Parity is with THIS REPO'S oracle — upstream parity
is unpinned."""
import numpy as np
import pytest

from drone_amd import abi
from helpers import assert_bits_equal, assert_outputs_equal, assert_state_equal, max_rel_state_error, to_np

pytestmark = pytest.mark.gpu


def make_pair(oracle, hip, n, seed, task=0, device=None, **over):
    ocfg = oracle.default_config(task, **over)
    hcfg = hip.default_config(task, **over)
    o = oracle.OracleVec(n, seed=seed, cfg=ocfg, threads=8)
    h = hip.DroneVec(n, seed=seed, cfg=hcfg, device=device)
    o.reset(seed)
    h.reset(seed)
    return o, h


def set_actions(h, a):
    if h.torch_device is None:
        h.actions[:] = a
    else:
        import torch

        h.actions.copy_(torch.from_numpy(a))


@pytest.mark.parametrize("task", [0, 1])
def test_reset_matches(oracle, hip, task):
    o, h = make_pair(oracle, hip, 1000, 7, task)
    assert_state_equal(o.get_state(), h.get_state(), "reset state")
    assert_outputs_equal(o, h, "reset")


@pytest.mark.parametrize("task,n,horizon", [(0, 1024, 1024), (1, 1024, 1024), (0, 333, 50), (1, 777, 64)])
def test_1000_step_random_rollout_bit_exact(oracle, hip, task, n, horizon):
    """BASELINE.json config 1 (1024 envs, random actions) and the truncation path."""
    o, h = make_pair(oracle, hip, n, 2024, task, horizon=horizon)
    terms = truncs = 0
    for t in range(1000):
        o.fill_random_actions()
        set_actions(h, o.actions)
        o.step()
        h.step()
        assert_outputs_equal(o, h, f"step {t}")
        terms += int(o.terminals.sum())
        truncs += int(o.truncations.sum())
        if t % 100 == 99:
            assert_state_equal(o.get_state(), h.get_state(), f"state@{t}")
    so, sh = o.get_state(), h.get_state()
    assert_state_equal(so, sh, "final state")
    assert max_rel_state_error(so, sh) <= 1e-5  # north-star bound; actual 0
    if horizon < 100:
        assert truncs > 0
    else:
        assert terms > 0
    lo, lh = o.log(), h.log()
    assert lo["n"] == lh["n"] and lo["n"] > 0
    for k in lo:
        assert lh[k] == pytest.approx(lo[k], rel=1e-6, abs=1e-7), k


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 257, 4097])
def test_ragged_sizes(oracle, hip, n):
    o, h = make_pair(oracle, hip, n, 5, 1, horizon=20)
    for t in range(60):
        o.fill_random_actions()
        set_actions(h, o.actions)
        o.step()
        h.step()
        assert_outputs_equal(o, h, f"n={n} step {t}")
    assert_state_equal(o.get_state(), h.get_state(), f"n={n}")


def test_device_random_policy_matches_oracle(oracle, hip):
    o, h = make_pair(oracle, hip, 1500, 99, 0, env_offset=12345)
    for g in (0, 1, 77, 2**31 + 5):
        a = o.fill_random_actions(gstep=g).copy()
        b = h.fill_random_actions(gstep=g)
        assert_bits_equal(a, b, f"random actions gstep={g}")
    assert np.all(a >= -1.0) and np.all(a < 1.0)


@pytest.mark.parametrize("task", [0, 1])
def test_fused_rollout_equals_stepping(oracle, hip, task):
    """SPEC.md §9 / BASELINE.json config 5 at test size."""
    o, h = make_pair(oracle, hip, 2000, 11, task, horizon=100)
    for T in (1, 128, 37):
        o.rollout(T)
        h.rollout(T)
        assert o.gstep == h.gstep
        assert_outputs_equal(o, h, f"rollout T={T}")
        assert_state_equal(o.get_state(), h.get_state(), f"rollout T={T}")
    # and the fused kernel against the per-step kernel on the device itself
    h2 = hip.DroneVec(2000, seed=11, cfg=hip.default_config(task, horizon=100))
    h2.reset(11)
    for _ in range(1 + 128 + 37):
        h2.fill_random_actions()
        h2.step()
    assert_state_equal(h.get_state(), h2.get_state(), "fused vs stepped on device")


def test_shard_invariance(oracle, hip):
    """Envs [0,N) as one vec == two vecs with env_offset (multi-GPU sharding)."""
    n, half = 1024, 512
    whole = hip.DroneVec(n, seed=3, cfg=hip.default_config(1))
    lo = hip.DroneVec(half, seed=3, cfg=hip.default_config(1, env_offset=0))
    hi = hip.DroneVec(half, seed=3, cfg=hip.default_config(1, env_offset=half))
    for v in (whole, lo, hi):
        v.reset(3)
        v.rollout(300)
    sw = whole.get_state()
    assert_state_equal(sw[:half], lo.get_state(), "lower shard")
    assert_state_equal(sw[half:], hi.get_state(), "upper shard")
    assert_bits_equal(whole.observations[half:], hi.observations, "upper shard obs")


def test_edge_states_nan_and_bounds(oracle, hip):
    """Injected states: on the bound, just outside, NaN position, huge velocity,
    zero quaternion, tick at the horizon."""
    n = 256
    o, h = make_pair(oracle, hip, n, 21, 0)
    rows = o.get_state()
    b = 5.0
    rows["pos"][0] = (b, 0, 0)                      # exactly on the bound
    rows["pos"][1] = (b + 0.5, 0, 0)                # outside: terminal on this step
    rows["pos"][2] = (np.nan, 0, 0)
    rows["vel"][3] = (1e30, -1e30, 1e30)
    rows["quat"][4] = (0, 0, 0, 0)                  # 1/sqrt(0) -> inf -> NaN -> oob
    rows["tick"][5] = 1023                          # truncates on this step
    rows["omega"][6] = (1e6, -1e6, 1e6)
    rows["rpm"][7] = (-5, 1e9, 0, 21702)
    rows["pos"][8] = (0, 0, -b)
    rows["vel"][8] = (0, 0, -20)
    # subnormal operands: both sides must keep them (gfx950 float_denorm_mode_32 = 3, x86 without FTZ/DAZ)
    rows["vel"][9] = (1e-40, -1e-41, 3e-39)
    rows["omega"][9] = (2e-39, -1e-45, 1e-38)
    rows["omega"][12] = (1e-20, 1e-20, -1e-20)   # products underflow into the subnormal range
    rows["rpm"][13] = (1e-20, 1e-19, 0.0, 1e-30)
    o.set_state(rows)
    h.set_state(rows)
    assert_state_equal(o.get_state(), h.get_state(), "after set_state")
    acts = np.zeros((n, 4), np.float32)
    acts[10] = (np.nan, 2.0, -3.0, np.inf)
    acts[11] = (1.0, -1.0, 1.0, -1.0)
    for t in range(3):
        o.actions[:] = acts
        set_actions(h, acts)
        o.step()
        h.step()
        assert_outputs_equal(o, h, f"edge step {t}")
        assert_state_equal(o.get_state(), h.get_state(), f"edge state {t}")
        if t == 0:
            assert o.terminals[1] == 1 and o.terminals[2] == 1 and o.truncations[5] == 1
    assert np.isfinite(to_np(h.observations)[:10]).all()


def test_done_list_compaction(oracle, hip):
    n = 3000
    o, h = make_pair(oracle, hip, n, 8, 0, horizon=40, compact_done=1)
    seen = 0
    for t in range(120):
        o.fill_random_actions()
        set_actions(h, o.actions)
        o.step()
        h.step()
        want = np.flatnonzero(o.terminals | o.truncations).astype(np.uint32)
        got = np.sort(h.done_list())
        assert_bits_equal(want, got, f"done ids step {t}")
        seen += len(want)
    assert seen > n


def test_device_buffers_torch(oracle, hip):
    """Zero-copy mode: torch tensors in HBM, launches on torch's stream."""
    import torch

    o, h = make_pair(oracle, hip, 5000, 31, 1, device="cuda:0")
    for t in range(200):
        o.fill_random_actions()
        h.fill_random_actions()
        assert_bits_equal(o.actions, h.actions, "device actions")
        o.step()
        h.step()
    torch.cuda.synchronize()
    assert_outputs_equal(o, h, "device-buffer step")
    assert_state_equal(o.get_state(), h.get_state(), "device-buffer state")
    # action ring via bind_actions
    ring = [torch.zeros_like(h.actions) for _ in range(3)]
    for k, r in enumerate(ring):
        h.fill_random_actions(gstep=h.gstep + k, out=r)
    for k, r in enumerate(ring):
        o.fill_random_actions()
        o.step()
        h.bind_actions(r)
        h.step()
    torch.cuda.synchronize()
    assert_outputs_equal(o, h, "ring step")


def test_init_failures_are_loud(hip):
    cfg = hip.default_config(0)
    cfg.struct_size = 4
    with pytest.raises(RuntimeError, match="struct_size"):
        hip.DroneVec(16, cfg=cfg)
    with pytest.raises(RuntimeError, match="not available"):
        hip.DroneVec(16, cfg=hip.default_config(0, device=63))
    with pytest.raises(RuntimeError):
        hip.DroneVec(0)


def test_lds_constants_variant(oracle, hip, tmp_path):
    """The north-star's "constants staged in LDS" build (-DDRONE_PARAMS_IN_LDS=1)
    of the same kernels passes the same bit-exact check (the default build passes
    them through the kernarg segment; DESIGN.md "Constants")."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = str(tmp_path / "libdrone_hip_lds.so")
    subprocess.run(["make", "-s", "-C", os.path.join(root, "drone_amd", "csrc"), "-B", f"OUT={lib}", "EXTRA=-DDRONE_PARAMS_IN_LDS=1"],
                   check=True, capture_output=True)
    fns = hip.load_variant(lib)
    for task in (0, 1, 2, 3):
        n = 3008 if task == 2 else 3001
        o = oracle.OracleVec(n, seed=6, cfg=oracle.default_config(task, horizon=90, collision_radius=0.5), threads=8)
        h = hip.DroneVec(n, seed=6, cfg=hip.default_config(task, horizon=90, collision_radius=0.5), fns=fns)
        o.reset(6)
        h.reset(6)
        for t in range(200):
            o.fill_random_actions()
            set_actions(h, o.actions)
            o.step()
            h.step()
        o.rollout(50)
        h.rollout(50)
        assert_outputs_equal(o, h, "lds variant")
        assert_state_equal(o.get_state(), h.get_state(), "lds variant state")


def test_non_default_physics_config(oracle, hip):
    """A heavier airframe, 3 RK4 substeps, other step size, bounds and reward weights."""
    over = dict(horizon=77, substeps=3, dt=0.02, mass=0.8, arm=0.18, ixx=4.0e-3, iyy=5.0e-3, izz=8.0e-3, k_thrust=2.0e-8,
                k_torque=3.0e-10, k_drag=0.03, k_ang_damp=2.0e-5, gravity=3.7, max_rpm=12000.0, motor_tau=0.08, max_vel=12.0,
                max_omega=25.0, bound=8.0, spawn_extent=1.5, target_extent=2.5, tilt_init=0.3, waypoint_radius=1.5,
                wind_theta=1.0, wind_sigma=2.0, wind_max=4.0, c_omega=3e-4, c_action=0.05, crash_penalty=2.0,
                progress_scale=2.0, waypoint_bonus=0.5, env_offset=4000000000)
    for task in (0, 1):
        o, h = make_pair(oracle, hip, 2049, 5150, task, **over)
        for t in range(300):
            o.fill_random_actions()
            set_actions(h, o.actions)
            o.step()
            h.step()
        assert_outputs_equal(o, h, "non-default cfg")
        assert_state_equal(o.get_state(), h.get_state(), "non-default cfg state")
        assert o.get_state()["episode"].sum() > 0


def test_unaligned_flag_buffers_take_the_byte_path(oracle, hip):
    """terminals / truncations handed over at odd addresses: the kernels fall back
    from 16-B packed flag stores to per-lane byte stores; results are the same."""
    import ctypes as C

    import torch

    n, seed = 1000, 13
    cfg = hip.default_config(0, horizon=30, buffer_kind=abi.BUFFERS_DEVICE, device=0)
    dev = torch.device("cuda:0")
    obs = torch.zeros((n, abi.OBS_DIM), dtype=torch.float32, device=dev)
    act = torch.zeros((n, abi.ACT_DIM), dtype=torch.float32, device=dev)
    rew = torch.zeros(n, dtype=torch.float32, device=dev)
    flags = torch.full((2, n + 64), 7, dtype=torch.uint8, device=dev)
    term, trunc = flags[0, 1:n + 1], flags[1, 3:n + 3]
    assert term.data_ptr() % 2 == 1 and trunc.data_ptr() % 2 == 1
    f = hip.load_variant(hip.LIB_PATH)
    h = f["drone_vec_init"](obs.data_ptr(), act.data_ptr(), rew.data_ptr(), term.data_ptr(), trunc.data_ptr(), n, seed, C.byref(cfg))
    assert h, f["drone_last_error"]()
    o = oracle.OracleVec(n, seed=seed, cfg=oracle.default_config(0, horizon=30))
    f["drone_vec_reset"](h, seed)
    o.reset(seed)
    for t in range(70):
        o.fill_random_actions()
        act.copy_(torch.from_numpy(o.actions))
        o.step()
        f["drone_vec_step"](h)
        f["drone_vec_sync"](h)
        assert_bits_equal(o.terminals, term, f"unaligned terminals {t}")
        assert_bits_equal(o.truncations, trunc, f"unaligned truncations {t}")
    assert_bits_equal(o.observations, obs, "obs")
    # guard bytes around the slices untouched
    fl = flags.cpu().numpy()
    assert fl[0, 0] == 7 and fl[0, n + 1] == 7 and fl[1, 2] == 7 and fl[1, n + 3] == 7
    f["drone_vec_close"](h)


def test_pipelined_gather_on_gpu(oracle, hip):
    """bind_outputs + PipelinedGather on the device (RCCL group of one rank): the
    gathered tensors of step k equal the oracle's outputs of step k while step k+1
    is already writing the other output set."""
    import os

    import torch
    import torch.distributed as dist

    from drone_amd.dist import PipelinedGather

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        n, seed = 40000, 77
        o, h = make_pair(oracle, hip, n, seed, 1, device="cuda:0", horizon=60)
        pg = PipelinedGather(h, n, abi.OBS_DIM)
        pending = []
        for t in range(12):
            o.fill_random_actions()
            h.fill_random_actions()
            (obs, rew, term, trunc), ev = pg.step()
            o.step()
            pending.append((obs, rew, term, trunc, ev, o.observations.copy(), o.rewards.copy(), o.terminals.copy(), o.truncations.copy()))
            if len(pending) == 2:  # consume one step behind, as an overlapped consumer would
                obs_, rew_, term_, trunc_, ev_, eo, er, et, eu = pending.pop(0)
                torch.cuda.current_stream().wait_event(ev_)
                assert_bits_equal(eo, obs_, f"gathered obs {t}")
                assert_bits_equal(er, rew_, f"gathered rew {t}")
                assert_bits_equal(et, term_, f"gathered term {t}")
                assert_bits_equal(eu, trunc_, f"gathered trunc {t}")
        torch.cuda.synchronize()
        assert_state_equal(o.get_state(), h.get_state(), "state after pipelined steps")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("zero_copy", ["0", "1"])
def test_host_buffer_transports(oracle, hip, zero_copy, monkeypatch):
    """Host-buffer mode, both transports: mirror copies (DRONE_HOST_ZEROCOPY=0) and
    the kernel accessing the caller's registered numpy buffers over PCIe (=1)."""
    monkeypatch.setenv("DRONE_HOST_ZEROCOPY", zero_copy)
    o, h = make_pair(oracle, hip, 5000, 41, 1, horizon=50)
    assert h.host_transport == ("zero-copy" if zero_copy == "1" else "mirror")  # the binding's own buffers own their pages
    for t in range(150):
        o.fill_random_actions()
        h.actions[:] = o.actions
        o.step()
        h.step()
        assert_outputs_equal(o, h, f"host transport {zero_copy} step {t}")
    o.rollout(40)
    h.rollout(40)
    assert_outputs_equal(o, h, "host transport rollout")
    # rebinding to fresh (unregistered) numpy buffers must keep working (falls back to the mirror transport)
    new_act = np.zeros_like(h.actions)
    new_out = (np.zeros_like(h.observations), np.zeros_like(h.rewards), np.zeros_like(h.terminals), np.zeros_like(h.truncations))
    h.bind_actions(new_act)
    h.bind_outputs(*new_out)
    for t in range(5):
        o.fill_random_actions()
        new_act[:] = o.actions
        o.step()
        h.step()
    assert_outputs_equal(o, h, "after rebinding")
    assert h.host_transport == "mirror"
    assert_state_equal(o.get_state(), h.get_state(), "host transport state")


@pytest.mark.parametrize("agents,n,device", [(8, 4096, None), (64, 4096, None), (2, 3000, "cuda:0"), (1, 777, None), (16, 65536, "cuda:0")])
def test_swarm_task_bit_exact(oracle, hip, agents, n, device):
    """Task 2 (SPEC.md §10): agents coupled through a nearest-neighbour term found with
    lane shuffles inside the wave; collisions forced to be frequent."""
    over = dict(agents_per_env=agents, collision_radius=0.6, proximity_radius=1.5, horizon=90, env_offset=64 * 5)
    o, h = make_pair(oracle, hip, n, 606, 2, device=device, **over)
    assert to_np(h.observations).shape == (n, 24)
    assert_outputs_equal(o, h, "swarm reset")
    crashes = 0
    for t in range(250):
        o.fill_random_actions()
        if device is None:
            set_actions(h, o.actions)
        else:
            h.fill_random_actions()
        o.step()
        h.step()
        if t % 10 == 0 or t > 240:
            assert_outputs_equal(o, h, f"swarm step {t}")
        crashes += int(o.terminals.sum())
    assert_state_equal(o.get_state(), h.get_state(), "swarm state")
    assert crashes > 0
    o.rollout(60)
    h.rollout(60)
    assert_outputs_equal(o, h, "swarm fused rollout")
    assert_state_equal(o.get_state(), h.get_state(), "swarm state after fused rollout")
    lo, lh = o.log(), h.log()
    assert lo["n"] == lh["n"] > 0


def test_swarm_rejects_bad_grouping(hip):
    with pytest.raises(RuntimeError, match="multiples of agents_per_env"):
        hip.DroneVec(100, cfg=hip.default_config(2, agents_per_env=8))
    with pytest.raises(RuntimeError, match="power of two"):
        hip.DroneVec(96, cfg=hip.default_config(2, agents_per_env=6))


def test_random_configs_random_sizes(oracle, hip):
    """Thirty random physical configurations x random shard sizes x the three tasks,
    per-step and fused, through the C-ABI."""
    import os

    rng = np.random.default_rng(int(os.environ.get("DRONE_FUZZ_SEED", "2025")))
    for trial in range(int(os.environ.get("DRONE_FUZZ_TRIALS", "30"))):
        task = trial % 4
        A = int(2 ** rng.integers(0, 7)) if task == 2 else 1
        n = int(rng.integers(1, 40)) * A * int(rng.integers(1, 30))
        over = dict(
            horizon=int(rng.integers(5, 150)), substeps=int(rng.integers(1, 4)), dt=float(rng.uniform(0.002, 0.03)),
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
            agents_per_env=A, collision_radius=float(rng.uniform(0.05, 1.0)), proximity_radius=float(rng.uniform(0.3, 3.0)),
            c_proximity=float(rng.uniform(0, 2)), gate_radius=float(rng.uniform(0.2, 3.0)), env_offset=int(rng.integers(0, 2**24)) * 64,
            compact_done=int(rng.integers(0, 2)))
        seed = int(rng.integers(0, 2**63))
        o, h = make_pair(oracle, hip, n, seed, task, **over)
        for t in range(60):
            o.fill_random_actions()
            set_actions(h, o.actions)
            o.step()
            h.step()
            if over["compact_done"] and t % 7 == 0:
                want = np.flatnonzero(o.terminals | o.truncations).astype(np.uint32)
                assert_bits_equal(want, np.sort(h.done_list()), f"trial {trial} done ids step {t}")
        assert_outputs_equal(o, h, f"trial {trial} (task {task}, n {n}, A {A})")
        o.rollout(25)
        h.rollout(25)
        assert_outputs_equal(o, h, f"trial {trial} fused")
        assert_state_equal(o.get_state(), h.get_state(), f"trial {trial} state")
        h.close()
        o.close()


@pytest.mark.parametrize("device", [None, "cuda:0"])
def test_race_task_bit_exact_with_forced_gate_passes(oracle, hip, device):
    """Task 3 (SPEC.md §11): random-policy steps plus rounds in which every drone is steered at
    its gate (through it, past its rim, or not quite reaching it), per-step and fused."""
    n, seed = 3000, 808
    o, h = make_pair(oracle, hip, n, seed, 3, device=device, gate_radius=1.0, horizon=400)
    assert to_np(h.observations).shape == (n, 24)
    assert_outputs_equal(o, h, "race reset")
    rng = np.random.default_rng(11)
    passes = 0
    for rnd in range(10):
        st = o.get_state()
        nrm, c = st["wind"], st["target"]
        side = np.cross(nrm, np.array([0.3, -0.5, 0.8], np.float32)).astype(np.float32)
        st["pos"] = (c - rng.uniform(0.0, 0.08, (n, 1)).astype(np.float32) * nrm + rng.uniform(0, 1.6, (n, 1)).astype(np.float32) * side).astype(np.float32)
        st["vel"] = (rng.uniform(2, 8, (n, 1)).astype(np.float32) * nrm).astype(np.float32)
        o.set_state(st)
        h.set_state(st)
        before = st["score_count"].copy()
        for t in range(4):
            o.fill_random_actions()
            if device is None:
                set_actions(h, o.actions)
            else:
                h.fill_random_actions()
            o.step()
            h.step()
            assert_outputs_equal(o, h, f"race round {rnd} step {t}")
        so = o.get_state()
        assert_state_equal(so, h.get_state(), f"race round {rnd}")
        passes += int((so["score_count"] > before).sum())
        if rnd % 3 == 2:
            o.rollout(20)
            h.rollout(20)
            assert_outputs_equal(o, h, f"race fused after round {rnd}")
    assert n < passes < 10 * n
    lo, lh = o.log(), h.log()
    assert lo["n"] == lh["n"]

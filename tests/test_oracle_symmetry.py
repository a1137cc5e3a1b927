"""Physics symmetries of the oracle — checks that need no second implementation, only the geometry of an X quadrotor.

Rotors (SPEC.md §3; positions read off the torque formulas): 0 at (+x, +y), 1 at (-x, +y), 2 at (-x, -y), 3 at (+x, -y),
spinning alternately. The airframe therefore maps onto itself under
  * a half turn about body z:            rotors 0<->2, 1<->3 (same spin directions), and
  * a mirror in the body x-z plane:      rotors 0<->3, 1<->2 (spin directions flip, as a mirror flips any yaw torque),
and the world is symmetric under any rotation about the vertical. If the transformed initial state and commands are
stepped, the result must be the transformed trajectory. As it turns out the oracle's arithmetic respects these
symmetries almost to the bit: body-frame quantities (angular rate, rotor speeds) come out EXACTLY transformed over 60
steps — sign flips are exact and the operations pair up — and world-frame ones within 1e-6 (2e-6 for a rotation by an
angle whose sine and cosine are not exact). The tests ask for exactly that. A wrong sign in a torque row, a cross
product, the quaternion kinematics or the frame of the angular rate shows as 1e-2 ... 1e+1 in the angular rate after
ONE step (tried: polar instead of axial transformation of the rate 1.7e+1; the other mirror's rotor permutation 1.2e-2;
no permutation 1.5e+0).
"""
import numpy as np
import pytest

N, STEPS, SEED = 128, 60, 5
# no episode ends, no clamps in the way, and symmetric inertia is NOT assumed: ixx != iyy keeps the tests honest
OVER = dict(bound=1.0e4, horizon=1 << 30, max_vel=1.0e4, max_omega=1.0e4, ixx=0.0041, iyy=0.0057)


def qmul(a, b):
    w1, x1, y1, z1 = a.T
    w2, x2, y2, z2 = b.T
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], 1).astype(np.float32)


def run(oracle, rows, actions):
    """Step the given start rows under the given per-step actions; returns the rows after every step."""
    v = oracle.OracleVec(N, seed=SEED, cfg=oracle.default_config(0, **OVER), threads=2)
    v.reset(SEED)
    v.set_state(rows)
    out = []
    for a in actions:
        v.actions[:] = a
        v.step()
        assert not v.terminals.any() and not v.truncations.any()
        out.append(v.get_state())
    return out


def start(oracle):
    v = oracle.OracleVec(N, seed=SEED, cfg=oracle.default_config(0, **OVER), threads=2)
    v.reset(SEED)
    rng = np.random.default_rng(1)
    for _ in range(15):  # tumble a little first: non-trivial attitude, rates and rotor speeds
        v.actions[:] = rng.uniform(-1, 1, (N, 4)).astype(np.float32)
        v.step()
    rows = v.get_state()
    acts = [rng.uniform(-1, 1, (N, 4)).astype(np.float32) for _ in range(STEPS)]
    return rows, acts


def close(a, b, what, tol):
    err = np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.maximum(1.0, np.abs(b.astype(np.float64))))
    assert err <= tol, f"{what}: {err:.3e}"


def same_attitude(qa, qb, what, tol):
    """q and -q are one attitude."""
    s = np.sign(np.sum(qa.astype(np.float64) * qb.astype(np.float64), axis=1, keepdims=True))
    close(qa * s.astype(np.float32), qb, what, tol)


def test_half_turn_about_body_z(oracle):
    rows, acts = start(oracle)
    base = run(oracle, rows, acts)
    perm = [2, 3, 0, 1]
    half = np.tile(np.array([[0, 0, 0, 1]], np.float32), (N, 1))  # 180 degrees about z
    t = rows.copy()
    t["quat"] = qmul(rows["quat"], half)       # the body turned about its own z axis
    t["omega"] = rows["omega"] * np.array([-1, -1, 1], np.float32)
    t["rpm"] = rows["rpm"][:, perm]
    got = run(oracle, t, [a[:, perm] for a in acts])
    for k in (0, 9, STEPS - 1):
        tol = 5e-6
        close(got[k]["pos"], base[k]["pos"], f"pos after {k + 1} steps", tol)
        close(got[k]["vel"], base[k]["vel"], f"vel after {k + 1} steps", tol)
        assert np.array_equal(got[k]["omega"], base[k]["omega"] * np.array([-1, -1, 1], np.float32)), f"omega after {k + 1} steps"
        assert np.array_equal(got[k]["rpm"], base[k]["rpm"][:, perm]), f"rpm after {k + 1} steps"
        same_attitude(got[k]["quat"], qmul(base[k]["quat"], half), f"attitude after {k + 1} steps", tol)
    # and the control: WITHOUT permuting the commands the turned drone flies somewhere else
    wrong = run(oracle, t, acts)
    assert np.max(np.abs(wrong[-1]["pos"] - base[-1]["pos"])) > 1e-2


def test_mirror_in_the_body_xz_plane(oracle):
    rows, acts = start(oracle)
    base = run(oracle, rows, acts)
    perm = [3, 2, 1, 0]
    my = np.array([1, -1, 1], np.float32)            # polar vectors: y -> -y
    mo = np.array([-1, 1, -1], np.float32)           # axial vectors (angular rate): the other two components flip
    mq = np.array([1, -1, 1, -1], np.float32)        # (w, x, y, z) -> (w, -x, y, -z): the mirrored rotation
    t = rows.copy()
    for f in ("pos", "vel", "target"):
        t[f] = rows[f] * my
    t["quat"] = rows["quat"] * mq
    t["omega"] = rows["omega"] * mo
    t["rpm"] = rows["rpm"][:, perm]
    got = run(oracle, t, [a[:, perm] for a in acts])
    for k in (0, 9, STEPS - 1):
        tol = 5e-6
        close(got[k]["pos"], base[k]["pos"] * my, f"pos after {k + 1} steps", tol)
        close(got[k]["vel"], base[k]["vel"] * my, f"vel after {k + 1} steps", tol)
        assert np.array_equal(got[k]["omega"], base[k]["omega"] * mo), f"omega after {k + 1} steps"
        assert np.array_equal(got[k]["rpm"], base[k]["rpm"][:, perm]), f"rpm after {k + 1} steps"
        same_attitude(got[k]["quat"], base[k]["quat"] * mq, f"attitude after {k + 1} steps", tol)
        close(got[k]["ep_return"], base[k]["ep_return"], f"return after {k + 1} steps", 1e-4)  # the hover reward sees only mirror-invariant quantities


@pytest.mark.parametrize("angle", [np.pi / 2, 0.7])
def test_rotation_of_the_world_about_the_vertical(oracle, angle):
    rows, acts = start(oracle)
    base = run(oracle, rows, acts)
    c, s = np.float32(np.cos(angle)), np.float32(np.sin(angle))
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)
    qz = np.tile(np.array([[np.cos(angle / 2), 0, 0, np.sin(angle / 2)]], np.float32), (N, 1))
    t = rows.copy()
    for f in ("pos", "vel", "target"):
        t[f] = rows[f] @ R.T
    t["quat"] = qmul(qz, rows["quat"])  # the world turned: rotation applied on the left; body-frame rates and rotors unchanged
    got = run(oracle, t, acts)
    for k in (0, 9, STEPS - 1):
        tol = 1e-5
        close(got[k]["pos"], base[k]["pos"] @ R.T, f"pos after {k + 1} steps", tol)
        close(got[k]["vel"], base[k]["vel"] @ R.T, f"vel after {k + 1} steps", tol)
        close(got[k]["omega"], base[k]["omega"], f"omega after {k + 1} steps", 1e-6)
        assert np.array_equal(got[k]["rpm"], base[k]["rpm"]), f"rpm after {k + 1} steps"
        same_attitude(got[k]["quat"], qmul(qz, base[k]["quat"]), f"attitude after {k + 1} steps", tol)
        close(got[k]["ep_return"], base[k]["ep_return"], f"return after {k + 1} steps", 1e-4)


def test_translation_of_drone_and_target(oracle):
    """Moving the drone and its target by the same vector changes nothing the dynamics can see: velocities, attitudes,
    rates and rotor speeds stay bit for bit, positions move by the vector (to the rounding of a larger coordinate)."""
    rows, acts = start(oracle)
    base = run(oracle, rows, acts)
    d = np.array([2.0, -4.0, 1.0], np.float32)
    t = rows.copy()
    t["pos"] = rows["pos"] + d
    t["target"] = rows["target"] + d
    got = run(oracle, t, acts)
    for k in (0, 9, STEPS - 1):
        for f in ("vel", "quat", "omega", "rpm"):
            assert np.array_equal(got[k][f], base[k][f]), f"{f} after {k + 1} steps"
        close(got[k]["pos"], base[k]["pos"] + d, f"pos after {k + 1} steps", 2e-6)
        close(got[k]["ep_return"], base[k]["ep_return"], f"return after {k + 1} steps", 1e-4)


def test_doubling_every_inertial_and_aerodynamic_constant(oracle):
    """Dimensional analysis: mass, the three inertias, thrust, rotor-torque, drag and angular-damping coefficients all
    times s leave every acceleration unchanged. With s = 2 every product in the step scales exactly, so the trajectory
    must come out bit for bit (a constant applied on one side of an equation only would break it)."""
    rows, acts = start(oracle)

    def fly(scale):
        cfg = oracle.default_config(0, **OVER)
        for f in ("mass", "ixx", "iyy", "izz", "k_thrust", "k_torque", "k_drag", "k_ang_damp"):
            setattr(cfg, f, getattr(cfg, f) * scale)
        v = oracle.OracleVec(N, seed=SEED, cfg=cfg, threads=2)
        v.reset(SEED)
        v.set_state(rows)
        for a in acts:
            v.actions[:] = a
            v.step()
        return v.get_state(), v.rewards.copy()

    (a, ra), (b, rb), (c, rc) = fly(1.0), fly(2.0), fly(0.5)
    for f in ("pos", "vel", "quat", "omega", "rpm", "ep_return"):
        assert np.array_equal(a[f], b[f]), f"{f}: constants x 2"
        assert np.array_equal(a[f], c[f]), f"{f}: constants x 0.5"
    assert np.array_equal(ra, rb) and np.array_equal(ra, rc)
    # the control: scaling the mass alone does change the flight
    cfg = oracle.default_config(0, **OVER)
    cfg.mass *= 2.0
    v = oracle.OracleVec(N, seed=SEED, cfg=cfg, threads=2)
    v.reset(SEED)
    v.set_state(rows)
    for x in acts:
        v.actions[:] = x
        v.step()
    assert np.max(np.abs(v.get_state()["pos"] - a["pos"])) > 1e-2

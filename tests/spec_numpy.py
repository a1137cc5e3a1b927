"""A third, independent statement of SPEC.md §4–§7 in numpy float64 (vectorised
over envs, no resets, no RNG). Used to catch transcription errors common to the
two C implementations: float32 results must track this to rounding error."""
import numpy as np


def derived(c):
    d = dict(c)
    d["h"] = c["dt"] / c["substeps"]
    d["arm_xy"] = c["arm"] * 0.70710678
    d["hover_rpm"] = np.sqrt(c["mass"] * c["gravity"] / (4 * c["k_thrust"]))
    return d


def rot(q):
    w, x, y, z = q.T
    R = np.empty((len(q), 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def deriv(c, S, cmd, wind):
    p, v, q, o, r = S
    f = c["k_thrust"] * r * r
    T = f.sum(1)
    tx = c["arm_xy"] * ((f[:, 0] + f[:, 1]) - (f[:, 2] + f[:, 3]))
    ty = c["arm_xy"] * ((f[:, 1] + f[:, 2]) - (f[:, 0] + f[:, 3]))
    r2 = r * r
    tz = c["k_torque"] * ((r2[:, 0] + r2[:, 2]) - (r2[:, 1] + r2[:, 3]))
    zb = rot(q)[:, :, 2]
    dv = (T / c["mass"])[:, None] * zb - np.array([0, 0, c["gravity"]]) - (c["k_drag"] / c["mass"]) * (v - wind)
    I = np.array([c["ixx"], c["iyy"], c["izz"]])
    tau = np.stack([tx, ty, tz], 1)
    do = (tau - np.cross(o, I * o) - c["k_ang_damp"] * o) / I
    w, x, y, z = q.T
    ox, oy, oz = o.T
    dq = 0.5 * np.stack([-x * ox - y * oy - z * oz, w * ox + y * oz - z * oy, w * oy + z * ox - x * oz, w * oz + x * oy - y * ox], 1)
    dr = (cmd - r) / c["motor_tau"]
    return [v.copy(), dv, dq, do, dr]


def rk4(c, S, cmd, wind):
    h = c["h"]
    add = lambda A, k, s: [a + s * b for a, b in zip(A, k)]
    k1 = deriv(c, S, cmd, wind)
    k2 = deriv(c, add(S, k1, h / 2), cmd, wind)
    k3 = deriv(c, add(S, k2, h / 2), cmd, wind)
    k4 = deriv(c, add(S, k3, h), cmd, wind)
    return [s + h / 6 * (a + 2 * b + 2 * d + e) for s, a, b, d, e in zip(S, k1, k2, k3, k4)]


def step(c, S, actions, wind):
    a = np.clip(actions, -1, 1)
    cmd = 0.5 * c["max_rpm"] * (a + 1)
    for _ in range(c["substeps"]):
        S = rk4(c, S, cmd, wind)
    p, v, q, o, r = S
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    v = np.clip(v, -c["max_vel"], c["max_vel"])
    o = np.clip(o, -c["max_omega"], c["max_omega"])
    r = np.clip(r, 0, c["max_rpm"])
    return [p, v, q, o, r]


def hover_reward(c, S, target, actions):
    p, v, q, o, r = S
    a = np.clip(actions, -1, 1)
    dist = np.linalg.norm(target - p, axis=1)
    return 1 - dist * (0.5 / c["bound"]) - (c["c_omega"] * (o * o).sum(1) + c["c_action"] * (a * a).sum(1))


def obs(c, S, target):
    p, v, q, o, r = S
    R = rot(q)
    body = lambda u: np.einsum("nij,ni->nj", R, u)  # R^T u
    return np.concatenate([body(v) / c["max_vel"], o / c["max_omega"], q, r / c["max_rpm"],
                           body(target - p) * (0.5 / c["bound"]), p / c["bound"]], 1)

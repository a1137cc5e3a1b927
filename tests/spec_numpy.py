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


def deriv(c, S, r, wind):
    """Rigid-body derivative at rotor speeds r (the rotors are not part of the RK4 system: their first-order
    lag toward the held command is integrated exactly, SPEC.md §4)."""
    p, v, q, o = S
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
    return [v.copy(), dv, dq, do]


def rk4(c, S5, cmd, wind):
    h = c["h"]
    S, r0 = S5[:4], S5[4]
    rotor = lambda t: cmd + (r0 - cmd) * np.exp(-t / c["motor_tau"])
    add = lambda A, k, s: [a + s * b for a, b in zip(A, k)]
    k1 = deriv(c, S, rotor(0.0), wind)
    k2 = deriv(c, add(S, k1, h / 2), rotor(h / 2), wind)
    k3 = deriv(c, add(S, k2, h / 2), rotor(h / 2), wind)
    k4 = deriv(c, add(S, k3, h), rotor(h), wind)
    return [s + h / 6 * (a + 2 * b + 2 * d + e) for s, a, b, d, e in zip(S, k1, k2, k3, k4)] + [rotor(h)]


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


# =====================================================================
# Round 2: the WHOLE env step (SPEC.md §2, §5, §6, §7, §11) restated independently:
# integer counter RNG in numpy uint64 (masked to 32 bits), wind, reward, episode
# end, reset draws, waypoint / gate dealing, log sums — float64 throughout.
# Used one step ahead (tests/test_oracle_independent.py): every step starts from
# the oracle's float32 state, so rounding error never accumulates and every
# branch — resets included — is compared at ~1e-6.
# =====================================================================
_M = np.uint64(0xFFFFFFFF)


def _u(x):
    return np.asarray(x, dtype=np.uint64) & _M


def hash32(x):
    x = _u(x)
    x = x ^ (x >> np.uint64(16))
    x = (x * np.uint64(0x7FEB352D)) & _M
    x = x ^ (x >> np.uint64(15))
    x = (x * np.uint64(0x846CA68B)) & _M
    x = x ^ (x >> np.uint64(16))
    return x


def stream_key(seed, stream):
    lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    return int(hash32(lo ^ int(hash32(hi ^ ((0x9E3779B9 * (stream + 1)) & 0xFFFFFFFF)))))


def rng_base(key, env, ctr):
    return hash32((hash32(_u(key) ^ _u(env)) + _u(ctr) * np.uint64(0x9E3779B9)) & _M)


def rng_draw(base, d):
    return hash32((_u(base) + _u(d) * np.uint64(0x85EBCA6B)) & _M)


def sym(u):
    return 2.0 * ((_u(u) >> np.uint64(8)).astype(np.float64) * 2.0 ** -24) - 1.0


RESET, ACTION, WIND, WAYPOINT = 0, 1, 2, 3


def random_actions(seed, env, gstep):
    k = hash32(_u(stream_key(seed, ACTION)) ^ _u(env))
    g = _u(gstep)
    h0 = hash32((k + ((np.uint64(2) * g) & _M) * np.uint64(0x9E3779B9)) & _M)
    h1 = hash32((k + ((np.uint64(2) * g + np.uint64(1)) & _M) * np.uint64(0x9E3779B9)) & _M)
    s16 = lambda h: (h.astype(np.int64) - 32768) / 32768.0
    return np.stack([s16(h0 & np.uint64(0xFFFF)), s16(h0 >> np.uint64(16)), s16(h1 & np.uint64(0xFFFF)), s16(h1 >> np.uint64(16))], 1)


def unit(e):
    return e / np.sqrt((e * e).sum(1, keepdims=True) + 1e-12)


def reset_draws(c, seed, env, episode, task):
    """SPEC.md §6 (+ §11 for the first gate normal): the fresh state of episode `episode` of global env `env`."""
    b = rng_base(stream_key(seed, RESET), env, episode)
    draws = [rng_draw(b, k) for k in range(5)]  # nine 16-bit values from five draws, low half first
    halves = [(draws[j // 2] >> np.uint64(16)) if j & 1 else (draws[j // 2] & np.uint64(0xFFFF)) for j in range(9)]
    u = [(h.astype(np.int64) - 32768) / 32768.0 for h in halves]
    pos = c["spawn_extent"] * np.stack(u[0:3], 1)
    tgt = c["target_extent"] * np.stack(u[3:6], 1)
    t = c["tilt_init"] * np.stack(u[6:9], 1)
    quat = np.concatenate([np.ones((len(pos), 1)), t], 1)
    n2 = (quat * quat).sum(1, keepdims=True)
    s1 = 1.5 - 0.5 * n2          # SPEC v5: two Newton steps of 1/sqrt(n2) about 1 (not the exact unit vector)
    s2 = 1.5 - 0.5 * (n2 * s1 * s1)
    quat = quat * (s1 * s2)
    n = len(pos)
    out = {"pos": pos, "vel": np.zeros((n, 3)), "quat": quat, "omega": np.zeros((n, 3)), "rpm": np.full((n, 4), c["hover_rpm"]),
           "target": tgt, "wind": np.zeros((n, 3))}
    if task == 3:
        out["wind"] = unit(tgt - pos)
    return out


def neighbour(c, pos):
    """SPEC.md §10: nearest neighbour of every agent among the A consecutive agents of its swarm, scanning
    d = 1 .. A-1 places round the swarm and keeping the first strict minimum. Returns (d2, e, gap) with `gap` the
    distance in d2 to the runner-up (a near-tie may be decided differently in float32)."""
    A = int(c["agents_per_env"])
    n = len(pos)
    P = pos.reshape(n // A, A, 3)
    best = np.full((n // A, A), (4.0 * c["bound"]) ** 2)
    second = np.full((n // A, A), np.inf)
    e_best = np.zeros((n // A, A, 3))
    for d in range(1, A):
        e = np.roll(P, -d, axis=1) - P          # p_{(i+d) mod A} - p_i
        d2 = (e * e).sum(-1)
        take = d2 < best
        second = np.where(take, best, np.minimum(second, d2))
        e_best = np.where(take[..., None], e, e_best)
        best = np.where(take, d2, best)
    return best.reshape(n), e_best.reshape(n, 3), (second - best).reshape(n)


def full_obs(c, S, target, aux, task):
    o = obs(c, S, target)
    if task == 2:
        p, v, q, om, r = S
        d2, e, _ = neighbour(c, p)
        eb = np.einsum("nij,ni->nj", rot(q), e) * (0.5 / c["bound"])
        o = np.concatenate([o, eb, (d2 / c["bound"] ** 2)[:, None]], 1)
    if task == 3:
        p, v, q, om, r = S
        R = rot(q)
        nb = np.einsum("nij,ni->nj", R, aux)
        d = ((p - target) * aux).sum(1, keepdims=True) / c["bound"]
        o = np.concatenate([o, nb, d], 1)
    return o


def env_step(c, seed, task, st, actions, gstep, env_ids):
    """One SPEC.md §5 step for tasks 0, 1, 3 from the state rows `st` (a dict of float64 / int arrays).
    Returns the new rows, (reward, terminal, truncation), the observation, and `margin`: how far the
    closest discontinuity (box wall, hover / waypoint radius, gate plane, ring rim) was from deciding
    differently — the caller ignores envs whose margin is at rounding level."""
    n = len(env_ids)
    a = np.clip(actions, -1, 1)
    wind = st["wind"].copy()
    pos0 = st["pos"].copy()
    tgt = st["target"].copy()
    margin = np.full(n, np.inf)
    if task == 1:
        b = rng_base(stream_key(seed, WIND), env_ids, gstep)
        decay = 1.0 - c["wind_theta"] * c["dt"]
        gain = c["wind_sigma"] * np.sqrt(c["dt"]) / np.sqrt(21845.0)
        for k in range(3):
            u = rng_draw(b, k)
            s = (u & np.uint64(255)) + ((u >> np.uint64(8)) & np.uint64(255)) + ((u >> np.uint64(16)) & np.uint64(255)) + (u >> np.uint64(24))
            wind[:, k] = np.clip(decay * wind[:, k] + gain * (s.astype(np.float64) - 510.0), -c["wind_max"], c["wind_max"])
    prev_dist = np.linalg.norm(tgt - pos0, axis=1)
    S = [st["pos"], st["vel"], st["quat"], st["omega"], st["rpm"]]
    S = step(c, S, a, wind if task == 1 else np.zeros((n, 3)))
    p, v, q, o, r = S
    tick = st["tick"] + 1
    dist = np.linalg.norm(tgt - p, axis=1)
    oob = (np.abs(p) > c["bound"]).any(1) | ~np.isfinite(p).all(1)
    margin = np.minimum(margin, np.abs(np.abs(p) - c["bound"]).min(1))
    trunc = ~oob & (tick >= c["horizon"])
    pen = c["c_omega"] * (o * o).sum(1) + c["c_action"] * (a * a).sum(1)
    score = st["score_count"].copy()
    aux = wind.copy() if task == 1 else st["wind"].copy()
    episode = st["episode"].copy()
    if task in (0, 2):
        rew = 1.0 - dist * (0.5 / c["bound"]) - pen
        score = score + (dist < c["hover_radius"])
        margin = np.minimum(margin, np.abs(dist - c["hover_radius"]))
        if task == 2:  # §10: collisions and the proximity penalty, on the post-integration positions of every agent
            nn_d2, _, gap = neighbour(c, p)
            coll_r2 = c["collision_radius"] ** 2
            oob = oob | (nn_d2 < coll_r2)
            trunc = ~oob & (tick >= c["horizon"])
            margin = np.minimum(margin, np.abs(nn_d2 - coll_r2))
            rew = rew - c["c_proximity"] * np.maximum(0.0, 1.0 - nn_d2 / c["proximity_radius"] ** 2)
    elif task == 1:
        rew = c["progress_scale"] * (prev_dist - dist) - pen
        hit = ~oob & (dist < c["waypoint_radius"])
        margin = np.minimum(margin, np.abs(dist - c["waypoint_radius"]))
        rew = rew + hit * c["waypoint_bonus"]
        score = score + hit
        b = rng_base(stream_key(seed, WAYPOINT), env_ids, episode)
        new_t = c["target_extent"] * np.stack([sym(rng_draw(b, 3 * score + k)) for k in range(3)], 1)
        tgt = np.where(hit[:, None], new_t, tgt)
    else:
        rew = c["progress_scale"] * (prev_dist - dist) - pen
        nrm = st["wind"]
        s0 = (nrm * (pos0 - tgt)).sum(1)
        s1 = (nrm * (p - tgt)).sum(1)
        cross = ~oob & (s0 < 0) & (s1 >= 0)
        margin = np.minimum(margin, np.minimum(np.abs(s0), np.abs(s1)))
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where(cross, s0 / (s0 - s1), 0.0)
        x = pos0 + t[:, None] * (p - pos0)
        m2 = ((x - tgt) ** 2).sum(1)
        gate_r2 = c["gate_radius"] ** 2
        margin = np.where(cross, np.minimum(margin, np.abs(m2 - gate_r2)), margin)
        passed = cross & (m2 < gate_r2)
        rew = rew + passed * c["waypoint_bonus"]
        score = score + passed
        b = rng_base(stream_key(seed, WAYPOINT), env_ids, episode)
        new_c = c["target_extent"] * np.stack([sym(rng_draw(b, 3 * score + k)) for k in range(3)], 1)
        new_n = unit(new_c - tgt)
        aux = np.where(passed[:, None], new_n, aux)
        tgt = np.where(passed[:, None], new_c, tgt)
    rew = rew - oob * c["crash_penalty"]
    ep_return = st["ep_return"] + rew
    done = oob | trunc
    logs = {k: st[k].copy() for k in ("perf_sum", "score_sum", "ret_sum", "len_sum", "n_sum", "oob_sum")}
    if task in (0, 2):
        sc = score.astype(np.float64)   # SPEC v5: the count; vec_log divides by the steps flown
        perf = sc
    else:
        sc = score.astype(np.float64)
        perf = np.where(score >= 8, 1.0, score * 0.125)
    logs["perf_sum"] += done * perf
    logs["score_sum"] += done * sc
    logs["ret_sum"] += done * ep_return
    logs["len_sum"] += done * tick
    logs["n_sum"] += done * 1.0
    logs["oob_sum"] += (done & oob) * 1.0
    episode = episode + done
    fresh = reset_draws(c, seed, env_ids, episode, task)
    new = {"pos": p, "vel": v, "quat": q, "omega": o, "rpm": r, "target": tgt, "wind": aux}
    for k in new:
        new[k] = np.where(done[:, None], fresh[k], new[k])
    new["tick"] = np.where(done, 0, tick)
    new["score_count"] = np.where(done, 0, score)
    new["ep_return"] = np.where(done, 0.0, ep_return)
    new["episode"] = episode
    new.update(logs)
    S2 = [new[k] for k in ("pos", "vel", "quat", "omega", "rpm")]
    if task == 2:
        # an agent's reset moves its neighbours' observation too; a near-tie between two neighbours, before or after
        # the resets, may pick the other one in float32: such swarms are excluded through the margin
        A = int(c["agents_per_env"])
        _, _, gap_after = neighbour(c, new["pos"])
        tie = np.minimum(gap, gap_after)
        margin = np.minimum(margin, tie)
        margin = np.repeat(margin.reshape(-1, A).min(1), A)  # one borderline agent makes its whole swarm borderline
    return new, (rew, oob, trunc), full_obs(c, S2, new["target"], new["wind"], task), margin

"""Free-running drift of the float32 oracle against the independent float64 statement of SPEC.md
(tests/spec_numpy.py): both start from the same reset state and see the same action sequence for `steps` steps and
are NEVER re-synchronised — what the north-star calls "max relative state error over 1000 steps", measured against
something that is not a transliteration of the oracle. Envs cannot end (huge box, horizon beyond the run), so the only
discontinuities are the task events (hover radius, waypoint hits, gate passes); an env whose float64 run came within
`eps` of deciding one differently is dropped from the statistics from that step on (its trajectory legitimately forks).

Error of a field = |f32 - f64| / max(|f64|, scale): relative for large values, absolute in units of the field's natural
scale near zero (a velocity component crossing 0 has no meaningful relative error)."""
import numpy as np

import spec_numpy as sn

FIELDS = ("pos", "vel", "quat", "omega", "rpm")


def cfgdict(cfg):
    return {k: (float(v) if isinstance(v, float) else v) for k, v in cfg.as_dict().items()}


def measure(oracle, task, n=256, steps=1000, seed=77, marks=(1, 10, 100, 300, 1000), eps=1e-4, policy_scale=1.0, **extra):
    over = dict(bound=1.0e4, horizon=1 << 30, max_vel=1.0e4, max_omega=1.0e4)  # no episode end, no velocity clamps in the way
    over.update(extra)
    cfg = oracle.default_config(task, **over)
    v = oracle.OracleVec(n, seed=seed, cfg=cfg, threads=4)
    v.reset(seed)
    c = sn.derived(cfgdict(cfg))
    env_ids = np.arange(n, dtype=np.uint64)
    rows = v.get_state()
    st = {f: rows[f].astype(np.float64) for f in ("pos", "vel", "quat", "omega", "rpm", "target", "wind", "ep_return", "perf_sum", "score_sum", "ret_sum", "len_sum", "n_sum", "oob_sum")}
    st.update({f: rows[f].astype(np.int64) for f in ("tick", "episode", "score_count")})
    scale = {"pos": 1.0, "vel": 1.0, "quat": 1.0, "omega": 1.0, "rpm": float(cfg.max_rpm)}
    alive = np.ones(n, bool)
    out = {}
    for t in range(1, steps + 1):
        v.fill_random_actions()
        if policy_scale != 1.0:  # gentler commands around hover: a flight regime instead of a tumble
            v.actions[:] = v.actions * policy_scale
        a32 = v.actions.copy()
        gstep = v.gstep
        v.step()
        st, (rew, oob, trunc), _, margin = sn.env_step(c, seed, task, st, a32.astype(np.float64), gstep, env_ids)
        assert not oob.any() and not trunc.any() and not v.terminals.any() and not v.truncations.any(), "an env ended: enlarge the box / horizon"
        alive &= margin > eps
        if t in marks:
            rows = v.get_state()
            err = {}
            for f in FIELDS:
                a, b = rows[f].astype(np.float64)[alive], st[f][alive]
                err[f] = np.abs(a - b) / np.maximum(np.abs(b), scale[f])
            worst = np.max(np.stack([e.reshape(len(e), -1).max(1) for e in err.values()]), axis=0)  # per env
            out[t] = {"alive": int(alive.sum()), "max": float(worst.max()), "p99": float(np.quantile(worst, 0.99)), "median": float(np.median(worst)),
                      "by_field_max": {f: float(e.max()) for f, e in err.items()},
                      "events_agree": bool(np.array_equal(rows["score_count"][alive], st["score_count"][alive]))}
    return out

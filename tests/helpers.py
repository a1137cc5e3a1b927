"""Shared helpers for the parity tests: drive the oracle and the HIP path with
the same calls and compare bit for bit."""
import numpy as np

from drone_amd import abi

STATE_FIELDS = abi.state_row_dtype().names


def usable_cores():
    """Threads worth giving the oracle on THIS box: the affinity mask and the cgroup quota, not the machine's core count
    (os.cpu_count() reports the host's; OpenMP teams larger than the quota spend their time in barriers)."""
    import os

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def to_np(x):
    return x.cpu().numpy() if type(x).__module__.startswith("torch") else np.asarray(x)


def assert_bits_equal(a, b, what):
    a = np.ascontiguousarray(to_np(a))
    b = np.ascontiguousarray(to_np(b))
    assert a.shape == b.shape and a.dtype == b.dtype, f"{what}: shape/dtype {a.shape}{a.dtype} vs {b.shape}{b.dtype}"
    if a.tobytes() != b.tobytes():
        if a.dtype == np.float32:
            # bitwise, except that any NaN equals any NaN: x86 and gfx950 generate
            # different default-NaN sign/payload bits and the spec does not pin them
            ne = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
        else:
            ne = a != b
        bad = np.argwhere(ne)
        if len(bad) == 0:
            return
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {a.size} elements differ; first at {i}: {a[i]!r} vs {b[i]!r}")


def assert_state_equal(sa, sb, what):
    for f in STATE_FIELDS:
        assert_bits_equal(sa[f], sb[f], f"{what}.{f}")


def assert_outputs_equal(o, h, what):
    assert_bits_equal(o.observations, h.observations, what + ".observations")
    assert_bits_equal(o.rewards, h.rewards, what + ".rewards")
    assert_bits_equal(o.terminals, h.terminals, what + ".terminals")
    assert_bits_equal(o.truncations, h.truncations, what + ".truncations")


def max_rel_state_error(sa, sb):
    """The north-star's figure of merit (<= 1e-5); we expect exactly 0."""
    worst = 0.0
    for f in ("pos", "vel", "quat", "omega", "rpm"):
        a = sa[f].astype(np.float64)
        b = sb[f].astype(np.float64)
        denom = np.maximum(np.abs(a), 1e-6)
        worst = max(worst, float(np.max(np.abs(a - b) / denom)))
    return worst

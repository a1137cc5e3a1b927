"""The hook that turns "parity unpinned" into work the hour upstream is mounted — and nothing more.

The reference snapshot has no simulator source: `/root/reference/.gitmodules:1-3` names an un-pinned `pufferlib` submodule
whose directory is empty, `/root/reference/.gitignore:14` hints at an older Cython shim (`simulator/cy_env.c`), also absent.
While that is so this test SKIPS with the reason. The day a human mounts `pufferlib/` (or `simulator/`), it FAILS, loudly, with
the to-do list: which upstream files were found, the constants and the observation width they state next to SPEC.md section 1
and DRONE_OBS_DIM, and the command that would build `oracle/_ref` from them with plain gcc. It builds no mock upstream, no
stand-in header, and checks nothing about the oracle itself: oracle work stays frozen until there is something to pin it to.
"""
import os
import re

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# what SPEC.md section 1 fixes, by the names an upstream C env would most plausibly use for them
SPEC_CONSTANTS = {"mass": 0.027, "arm": 0.0397, "ixx": 1.4e-5, "iyy": 1.4e-5, "izz": 2.17e-5, "k_thrust": 3.16e-10, "k_torque": 7.94e-12, "k_drag": 0.0027,
                  "gravity": 9.81, "max_rpm": 21702.0, "motor_tau": 0.05, "max_vel": 20.0, "max_omega": 50.0, "dt": 0.01, "bound": 5.0, "horizon": 1024}


def upstream_sources():
    """C / Cython / header files under the reference's submodule or its older `simulator/` directory that mention a drone."""
    found = []
    for top in ("pufferlib", "simulator"):
        base = os.path.join(REF, top)
        if not os.path.isdir(base):
            continue
        for d, _, files in os.walk(base):
            for f in files:
                if f.endswith((".h", ".c", ".pyx", ".pxd", ".cpp")) and ("drone" in f.lower() or "drone" in d.lower()):
                    found.append(os.path.join(d, f))
    return sorted(found)


def test_upstream_step_is_pinned_or_visibly_absent():
    if not os.path.isdir(REF):
        pytest.skip(f"{REF} does not exist on this box (the GPU box never has it): parity stays UNPINNED (SPEC.md header)")
    src = upstream_sources()
    if not src:
        empty = [t for t in ("pufferlib", "simulator") if os.path.isdir(os.path.join(REF, t)) and not os.listdir(os.path.join(REF, t))]
        pytest.skip(f"no drone env source under {REF} (empty: {empty or 'no such directories'}; .gitmodules:1-3 names an un-pinned pufferlib "
                    "submodule): parity stays UNPINNED — oracle/ restates this repo's SPEC.md, not tensaur/drone")
    # ---- upstream is here: say exactly what has to happen now ----
    report = [f"UPSTREAM SOURCE FOUND ({len(src)} files) — parity can and must be pinned now:"] + [f"  {p}" for p in src[:20]]
    headers = [p for p in src if p.endswith(".h")]
    text = ""
    for p in (headers or src)[:8]:
        with open(p, errors="replace") as fh:
            text += fh.read()
    report.append("constants stated upstream next to SPEC.md section 1 (name: upstream literal(s) | SPEC):")
    for name, spec in SPEC_CONSTANTS.items():
        hits = re.findall(r"(?i)\b" + re.escape(name) + r"\b\s*(?:=|\s)\s*\(?\s*(-?[0-9.]+(?:[eE][-+]?[0-9]+)?)f?", text)
        report.append(f"  {name}: {', '.join(sorted(set(hits))) or 'not found by name'} | {spec}")
    obs = re.findall(r"(?i)(?:obs\w*|observation\w*)\s*(?:\[|=|\s)\s*\(?\s*([0-9]+)", text)
    report.append(f"observation width candidates upstream: {sorted(set(obs)) or 'none found'} | DRONE_OBS_DIM = 20 (24 for tasks 2, 3) in include/drone_vec.h")
    report += [
        "to do, in order:",
        "  1. read upstream's c_step / c_reset / compute_observations and list every difference from SPEC.md sections 4-7 (integrator, reward, reset, obs)",
        "  2. make SPEC.md state upstream's choices; bump SPEC_VERSION (drone_amd/binding.py) and regenerate tests/golden with tests/golden/make_golden.py",
        "  3. build the real step as the reference oracle, outputs only under oracle/_ref/ (git-ignored, NOT gpurun-ignored), e.g.:",
        f"       gcc -O2 -fPIC -shared -ffp-contract=off -I{os.path.dirname((headers or src)[0])} <a ten-line driver that loops c_step over N envs> -lm -o {ROOT}/oracle/_ref/libdrone_ref.so",
        "     commit the recipe as oracle/Makefile target `_ref`; never copy upstream sources into this repo",
        "  4. check oracle/drone_oracle.h against oracle/_ref on seeded random-action rollouts (1000 steps, <= 1e-5 relative: BASELINE.json north_star), then the HIP path against both",
        "  5. drop 'PARITY UNPINNED' from SPEC.md, oracle/*.h, oracle/pyoracle.py, DESIGN.md, README.md; INTEGRATION.md: import-test the binding inside PufferLib",
    ]
    pytest.fail("\n".join(report))

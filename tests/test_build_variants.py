"""The compile-time knobs of drone_kernels.hip that tools and A/B logs refer to must keep BUILDING (gfx950 cross-compile, no
GPU needed): round 4 found -DDRONE_PARAMS_GLOBAL=1 broken since the packed RK4 went in. Compiled in parallel, device code
only, output discarded; the variants' RESULTS are checked where they are used (tests/test_parity_gpu.py builds and runs the
LDS-constants variant, tools/ab_step.py times the others)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "drone_amd", "csrc")
BASE = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
        "-mllvm", "-amdgpu-kernarg-preload-count=12", "--cuda-device-only", "-c", "drone_kernels.hip", "-o", "/dev/null"]
VARIANTS = ["-DDRONE_PARAMS_GLOBAL=1", "-DDRONE_STAMPS=1", "-DDRONE_EARLY_ARGS=0", "-DDRONE_EARLY_ARGS=3", "-DDRONE_SCALAR_RESET=1", "-DDRONE_STEP_TILES=2 -DDRONE_STEP_WAVE_OUTPUTS=1",
            "-DDRONE_PK_RK4=0 -DDRONE_CARRY_ROTOR=0"]


def test_every_documented_knob_still_compiles():
    procs = [(v, subprocess.Popen(BASE + v.split(), cwd=SRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)) for v in VARIANTS]
    failed = []
    for v, p in procs:
        so, se = p.communicate(timeout=900)
        if p.returncode != 0:
            failed.append(f"{v}: {se[-600:]}")
    assert not failed, "\n".join(failed)

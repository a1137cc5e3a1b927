"""Two checks of drone_kernels.hip that need hipcc but no GPU (gfx950 cross-compile).

1. The compile-time variants a test, a tool or a fallback build still selects must keep BUILDING (round 4 found one broken
   for two rounds). Their RESULTS are checked where they are used: tests/test_parity_gpu.py builds and runs the LDS-constants
   variant on the GPU box, tools/stamps.py the stamped one, tools/r06_flake.sh the build without kernarg preloading.
2. The ISA of every shipped kernel instantiation is the MEASURED one: round 6 pruned the measured-dead
   knobs out of the source and re-landed the peer-store stop word as instantiations of their own, under the rule that not one
   instruction and not one descriptor field of the kernels the round-5 profiles measured may move. tests/golden/
   isa_shipped_r05.json is tools/isa_digest.py's digest of the round-5 library's listing (sha256 of each kernel's
   instruction stream with labels renumbered, its descriptor, register counts), isa_r05_no_preload.json the same sources built
   without kernarg preloading (-DDRONE_EARLY_ARGS=0: round 5's build B). New kernels may appear; none may change or go.
   A deliberate kernel change regenerates the golden file — and owes TUNING.md a measurement.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "drone_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
COMMON = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "--cuda-device-only"]
PRELOAD = ["-mllvm", "-amdgpu-kernarg-preload-count=12", "-DDRONE_EARLY_ARGS=2"]  # make PRELOAD=... EXTRA=-DDRONE_EARLY_ARGS=2: the address-forming arguments arrive in SGPRs with the wave
NO_PRELOAD = ["-DDRONE_EARLY_ARGS=0"]
VARIANTS = [("-DDRONE_STAMPS=1", []), ("-DDRONE_PARAMS_IN_LDS=1", []), ("-DDRONE_PK_RK4=0", []), ("-DDRONE_STAMPS=1", PRELOAD)]
# The two builds whose ISA is pinned: round 5's sources compiled the same two ways (tools/isa_digest.py on `hipcc -S`). Round 6
# ships the one WITHOUT kernarg preloading (Makefile PRELOAD says why): against round 5's shipped library that changes the 48
# per-step instantiations and three one-wave helpers — the fused rollout (8), step_many (32), reset (4) and fill kernels are
# byte-identical in both — and profiles/r06_* measures it.
PINNED = {"isa_r05_no_preload.json": [], "isa_shipped_r05.json": PRELOAD}


def test_every_variant_somebody_selects_still_compiles_and_the_shipped_isa_is_the_measured_one(tmp_path):
    procs = [(v, subprocess.Popen(COMMON + pre + v.split() + ["-c", "drone_kernels.hip", "-o", "/dev/null"], cwd=SRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
             for v, pre in VARIANTS]
    listings = {g: (str(tmp_path / (g + ".s")), subprocess.Popen(COMMON + flags + ["-S", "-o", str(tmp_path / (g + ".s")), "drone_kernels.hip"], cwd=SRC, stdout=subprocess.PIPE,
                                                                stderr=subprocess.PIPE, text=True)) for g, flags in PINNED.items()}
    failed = []
    for v, p in procs:
        so, se = p.communicate(timeout=900)
        if p.returncode != 0:
            failed.append(f"{v}: {se[-600:]}")
    assert not failed, "\n".join(failed)

    import isa_digest

    for golden, (listing, p) in listings.items():
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, se[-800:]
        now = isa_digest.digest(listing)
        want = json.load(open(os.path.join(ROOT, "tests", "golden", golden)))["kernels"]
        gone = sorted(set(want) - set(now))
        # drone_flag_wait_kernel only ever runs for handles in a peer-store exchange: round 6 gave it the stop word to raise
        peer_only = {"drone_flag_wait_kernel"}
        changed = sorted(k for k in want if k in now and now[k] != want[k] and k not in peer_only)
        assert not gone, f"{golden}: kernels of the measured build that no longer exist: {gone}"
        assert not changed, f"{golden}: kernels whose ISA or descriptor differs from round 5's: " + "; ".join(
            f"{k}: " + ", ".join(f"{f} {want[k][f]} -> {now[k][f]}" for f in want[k] if want[k][f] != now[k][f] and not f.endswith("sha256")) for k in changed[:8])
        assert len(want) == 96
        if not PINNED[golden]:  # the shipped build: no kernel is entered through a preload trampoline
            assert all(k["kernarg_preload"] == 0 for k in now.values())
        # what is new must be named as such: the peer instantiations (PEER = true, the stop word; task x compact x layout for the step
        # kernel — no load hints —, task x packed for the rollout, task for the reset) and nothing else
        extra = sorted(set(now) - set(want))
        assert all("_peer_kernel<" in k for k in extra) and len(extra) == 12 + 8 + 4, extra
        assert now["drone_flag_wait_kernel"]["instructions"] < 120

#!/usr/bin/env python3
"""gpurun_out/<round>_prof/ (written on the GPU box by tools/<round>_profiles.sh) -> profiles/<round>_*: per profile directory the
drone kernels' rows of rocprofv3's kernel_stats.csv, and a summary.json that names the BUILD (git revision + sha256 of the
libdrone_hip.so that ran) and every kernel by its demangled AND mangled name (looked up in drone_amd/csrc/drone_kernels.s);
then profiles/traffic_latest.json and profiles/rollout_valu.json from the same run. tests/test_profiles.py checks that
what is committed is consistent (half of profiles/r03_* came from another build).
   python tools/collect_round.py [r05]        (round 4's set was made by this file under its old name, collect_r04.py)"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r05"
SRC = os.path.join(ROOT, "gpurun_out", f"{ROUND}_prof")
CXXFILT = "c++filt"  # binutils; the ROCm image ships no llvm-cxxfilt


def mangled_names():
    """{demangled: mangled} of every kernel in the ISA listing of the current sources (make asm)."""
    s_path = os.path.join(ROOT, "drone_amd", "csrc", "drone_kernels.s")
    subprocess.run(["make", "-s", "-C", os.path.dirname(s_path), "asm"], check=True, capture_output=True)
    names = sorted(set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", open(s_path).read(), re.M)))
    dem = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    return dict(zip(dem, names))


def norm(name):
    """rocprofv3 and llvm-cxxfilt print the same demangling up to a leading 'void ' and spacing."""
    return re.sub(r"\s+", "", name.replace("void ", "", 1) if name.startswith("void ") else name)


def main():
    build = json.load(open(os.path.join(SRC, "build.json")))
    table = {norm(d): (d, m) for d, m in mangled_names().items()}
    made = []
    for d in sorted(os.listdir(SRC)):
        src = os.path.join(SRC, d)
        summ_path = os.path.join(src, "summary.json")
        if not os.path.isfile(summ_path):
            continue
        summ = json.load(open(summ_path))
        dst = os.path.join(ROOT, "profiles", f"{ROUND}_{d}")
        os.makedirs(dst, exist_ok=True)
        rows, header = [], None
        # ONE file: gpurun merges every call's output into the same gpurun_out/ tree and rocprofv3 names its files by pid, so the
        # files of earlier evidence runs lie beside the newest one (the first round-5 sets carried an older run's rows as well)
        stats = sorted(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
        if len(stats) > 1:
            print(f"{d}: {len(stats)} kernel_stats.csv files, taking the newest ({os.path.basename(stats[-1])})", file=sys.stderr)
        for f in stats[-1:]:
            with open(f) as fh:
                rd = csv.reader(fh)
                header = next(rd)
                rows += [r for r in rd if "drone" in r[0]]
        with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as fh:
            w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(header)
            w.writerows(rows)
        kernels = []
        for r in rows:
            key = norm(r[0])
            if key not in table:
                sys.exit(f"{d}: kernel {r[0]!r} of the profile is not in drone_kernels.s — the profile is from another build")
            kernels.append({"name": r[0], "mangled": table[key][1], "calls": int(r[1]), "avg_us": float(r[3]) / 1e3, "min_us": float(r[5]) / 1e3, "max_us": float(r[6]) / 1e3})
        out = {"profile": f"{ROUND}_{d}", "build": build, "kernels": kernels,
               "traffic": {k: v for k, v in summ.get("traffic", {}).items() if "drone" in k},
               "kernel_trace_avg_us": {k: v for k, v in summ.get("kernel_trace_avg_us", {}).items() if "drone" in k}}
        json.dump(out, open(os.path.join(dst, "summary.json"), "w"), indent=1)
        line = os.path.join(src, "bench_line_under_rocprof.json")
        if os.path.isfile(line) and os.path.getsize(line):
            shutil.copy(line, os.path.join(dst, "bench_line_under_rocprof.json"))
        made.append(d)
    # SQ counter passes (tools/pmc_pass.sh): sq_<profile>[_<set>] -> profiles/<round>_<profile>/sq_counters.json; several passes over one
    # workload (the SQ block has eight counters per pass) are merged, each counter named once
    merged = {}
    for d in sorted(os.listdir(SRC)):
        pmc = os.path.join(SRC, d, "pmc_avg.json")
        if not (d.startswith("sq_") and os.path.isfile(pmc)):
            continue
        target = d[3:]
        for suffix in ("_a", "_b", "_c"):
            if target.endswith(suffix):
                target = target[:-2]
        target = {"step_65536": "step_hover_65536"}.get(target, target)
        for kernel, vals in json.load(open(pmc)).items():
            merged.setdefault(target, {}).setdefault(kernel, {}).update(vals)
    for target, counters in merged.items():
        dst = os.path.join(ROOT, "profiles", f"{ROUND}_{target}")
        os.makedirs(dst, exist_ok=True)
        json.dump({"build": build, "counters": counters}, open(os.path.join(dst, "sq_counters.json"), "w"), indent=1)
    for name in ("bench_default", "bench_driver_window", "bench_force_dist_one_rank"):
        p = os.path.join(SRC, name + ".json")
        lines = [l for l in open(p).read().splitlines() if l.startswith("{")] if os.path.isfile(p) else []
        if lines:
            open(os.path.join(ROOT, "profiles", f"{ROUND}_{name}.json"), "w").write(lines[-1] + "\n")
    for name in ("wg_census.txt", "wg_census_oldest_first.txt", "rollout_clock.txt"):  # diagnostic builds of the same sources
        p = os.path.join(SRC, name)
        if os.path.isfile(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(ROOT, "profiles", f"{ROUND}_{name}"))
    # traffic_latest.json: PMC bytes per launch of the per-step kernel by task:envs, all from this run
    traffic = {}
    for key, d in (("hover:4194304", "step_hover_4194304"), ("hover:1048576", "step_hover"), ("hover:65536", "step_hover_65536"), ("hover:131072", "step_hover_131072"),
                   ("waypoint:262144", "step_waypoint_262144")):
        try:
            s = json.load(open(os.path.join(ROOT, "profiles", f"{ROUND}_{d}", "summary.json")))
        except OSError:
            continue
        name = next((k for k in s["traffic"] if "step_kernel" in k), None)
        if not name or s["traffic"][name]["hbm_bytes_per_launch"] is None:
            continue
        t = s["traffic"][name]
        k = next(x for x in s["kernels"] if x["name"] == name)
        traffic[key] = {"hbm_bytes_per_launch": t["hbm_bytes_per_launch"], "read_bytes": t["read_bytes_corrected"], "write_bytes": t["write_bytes"],
                        "rocprof_kernel_avg_us": k["avg_us"], "kernel": name, "mangled": k["mangled"],
                        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, KiB units, FETCH_SIZE doubled (gfx950 wide coalesced reads count at half), averaged over dispatches",
                        "source": f"profiles/{ROUND}_{d}/summary.json", "build": build}
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
    # rollout_valu.json: VALU instructions per wave-step from the SQ pass + ISA mix of the same sources
    try:
        c = json.load(open(os.path.join(SRC, "sq_rollout_hover_a" if os.path.isdir(os.path.join(SRC, "sq_rollout_hover_a")) else "sq_rollout_hover", "pmc_avg.json")))
        k = next(v for n, v in c.items() if "rollout" in n)
        per = k["SQ_INSTS_VALU"] / k["SQ_WAVES"] / 128.0
        note = (f"SQ_INSTS_VALU {k['SQ_INSTS_VALU']:.4g} / {k['SQ_WAVES']:.0f} waves / 128 steps (profiles/{ROUND}_rollout_hover/sq_counters.json, build {build.get('git_head', '?')[:12]}); "
                "issue roof: 1.03 ns per wave64 VALU per SIMD, the best f32 rate measured (v_mul_f32, 8 waves/SIMD; v_fma_f32 1.17-1.22 ns) in profiles/micro_valu_issue.txt")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rollout_flops.py"), "hover", "drone_rollout_kernelILi0ELb0ELb0EE", f"{per:.1f}", "1.03", note], check=True)
    except (OSError, StopIteration, KeyError) as exc:
        print("rollout_valu.json not refreshed:", exc)
    print("profiles written for:", made)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""gpurun_out/r06_flake/<tag>/{box.txt,runs.jsonl,...} (tools/r06_flake.sh, one tag per gpurun call = per box) ->
profiles/r06_flake/summary.json + a table on stdout: failures / runs per (build, workload) cell, per box, the whole-suite
runs, the fault texts seen, and an upper bound on the failure rate of each build's eight-process runs (exact one-sided 95 %
Clopper-Pearson bound; with zero failures in n runs that is 1 - 0.05^(1/n), the "rule of three" 3/n).

    python3 tools/r06_flake_summary.py            # writes profiles/r06_flake/summary.json, copies every failure's text beside it
"""
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r06_flake")
DST = os.path.join(ROOT, "profiles", "r06_flake")
BUILDS = {"A": "as shipped by round 5 (kernarg preload, -DDRONE_EARLY_ARGS=2)", "B": "-DDRONE_EARLY_ARGS=0 without -mllvm -amdgpu-kernarg-preload-count=12",
          "C": "A + profiles/r05_ab/stop_word_withdrawn.patch (round 5's stop word in every instantiation)",
          "D": "round 6's source (pruned; stop word as PEER instantiations only) built WITH kernarg preloading",
          "E": "round 6 as shipped: the same source without kernarg preloading (make default)"}


def upper95(fails, n):
    """one-sided 95 % Clopper-Pearson upper bound on a failure probability, by bisection on the binomial tail"""
    if n == 0:
        return None
    if fails >= n:
        return 1.0
    from math import comb

    def cdf(p):  # P[X <= fails]
        return sum(comb(n, k) * p ** k * (1 - p) ** (n - k) for k in range(fails + 1))

    lo, hi = 0.0, 1.0
    for _ in range(60):
        mid = (lo + hi) / 2
        lo, hi = (mid, hi) if cdf(mid) > 0.05 else (lo, mid)
    return round(hi, 4)


def main():
    os.makedirs(DST, exist_ok=True)
    runs, boxes = [], {}
    for tag_dir in sorted(glob.glob(os.path.join(SRC, "*"))):
        tag = os.path.basename(tag_dir)
        p = os.path.join(tag_dir, "runs.jsonl")
        if not os.path.isfile(p):
            continue
        for line in open(p):
            line = line.strip()
            if line:
                try:
                    runs.append(json.loads(line))
                except ValueError:
                    print("unparseable line in", p, ":", line[:120], file=sys.stderr)
        box = open(os.path.join(tag_dir, "box.txt")).read() if os.path.isfile(os.path.join(tag_dir, "box.txt")) else ""
        boxes[tag] = {"box_txt": box.strip().splitlines()[:14]}
        for f in sorted(glob.glob(os.path.join(tag_dir, "fail_*.txt")) + glob.glob(os.path.join(tag_dir, "soft_*.txt")) + glob.glob(os.path.join(tag_dir, "*.dmesg"))):
            shutil.copy(f, os.path.join(DST, f"{tag}_{os.path.basename(f)}"))
        for f in sorted(glob.glob(os.path.join(tag_dir, "suite_*.txt"))):
            text = open(f, errors="replace").read()
            if " failed" in text.splitlines()[-1] if text.strip() else True:  # a suite with failures (or no summary line): keep its whole output
                shutil.copy(f, os.path.join(DST, f"{tag}_{os.path.basename(f)}"))
            else:
                open(os.path.join(DST, f"{tag}_{os.path.basename(f)}"), "w").write("\n".join(text.splitlines()[-3:]) + "\n")
    cells, suites, per_build, faults = {}, {}, {}, {}
    for r in runs:
        failed = r["rc"] != 0
        if r["workload"] == "suite":
            s = suites.setdefault(r["build"], {"runs": 0, "failed": 0, "detail": []})
            s["runs"] += 1
            s["failed"] += failed
            text = ""
            try:  # pytest's own summary line (the record's is whatever came last on stdout: RCCL's banner)
                f = os.path.join(SRC, r["tag"], f"suite_{r['build']}_{r['rep']}.txt")
                text = next((l.strip(" =") for l in reversed(open(f, errors="replace").read().splitlines()) if re.search(r"\d+ passed|\d+ failed", l)), "")
            except OSError:
                pass
            s["detail"].append({"tag": r["tag"], "rc": r["rc"], "seconds": r["seconds"], "summary": text or r.get("summary", "")})
        else:
            c = cells.setdefault(r["build"], {}).setdefault(r["workload"], {"runs": 0, "failed": 0, "soft": 0, "seconds": 0, "by_box": {}})
            c["runs"] += 1
            c["failed"] += failed
            m = re.search(r"(?:(\d+) failed, )?(\d+) passed", str(r.get("summary", "")))
            if m:  # a pytest session of several cases (peer_small / peer_file): cases counted as well as sessions
                c["cases_failed"] = c.get("cases_failed", 0) + int(m.group(1) or 0)
                c["cases"] = c.get("cases", 0) + int(m.group(1) or 0) + int(m.group(2))
            if r.get("mismatches"):
                c["torn_batches"] = c.get("torn_batches", 0) + 1
            if r["workload"] == "peer_stress":
                c["handshake_rounds"] = c.get("handshake_rounds", 0) + 3000
            c["seconds"] += r["seconds"]
            soft = (not failed) and r["workload"] == "bench8" and not str(r.get("optional", "")).startswith("ok")
            c["soft"] += soft
            b = c["by_box"].setdefault(r["tag"], [0, 0])
            b[0] += failed
            b[1] += 1
            t = per_build.setdefault(r["build"], [0, 0])
            t[0] += failed
            t[1] += 1
        for f in r.get("faults", []):
            if failed or "HSA" in f or "fault" in f or "ILLEGAL" in f:
                faults.setdefault(r["build"], {}).setdefault(f.rsplit(" x", 1)[0], []).append(f"{r['tag']}/{r['workload']}/{r['rep']}")
    for b in cells.values():
        for c in b.values():
            c["avg_seconds"] = round(c.pop("seconds") / max(1, c["runs"]), 1)
    def describe(b):
        lib, _, env = b.partition("+")
        return BUILDS.get(lib, lib) + (f"; run with {env}" if env else "")

    out = {"builds": {b: describe(b) for b in sorted(set(per_build) | set(suites))}, "boxes": boxes,
           "eight_process_cells": cells, "whole_gpu_suite": suites,
           "totals_over_cells": {b: {"failed": f, "runs": n, "failure_rate_upper_bound_95": upper95(f, n)} for b, (f, n) in sorted(per_build.items())},
           "fault_texts": faults,
           "note": "every repetition is a fresh child process (a pytest session of its own, or bench.py --gpus 8 itself); eight processes share the one GPU of the "
                   "box; `soft` = the bench line arrived with rc 0 but an optional exchange record lost its budget"}
    json.dump(out, open(os.path.join(DST, "summary.json"), "w"), indent=1, sort_keys=True)
    print(f"{len(runs)} runs on {len(boxes)} box(es)")
    wl = sorted({w for b in cells.values() for w in b})
    print(f"{'build':<28} " + "  ".join(f"{w:>11}" for w in wl) + "   |  all cells   <=95%   | suites")
    for b in sorted(set(cells) | set(suites)):
        row = "  ".join(f"{cells.get(b, {}).get(w, {}).get('failed', 0)}/{cells.get(b, {}).get(w, {}).get('runs', 0):<3}".rjust(11) for w in wl)
        f, n = per_build.get(b, (0, 0))
        s = suites.get(b, {"failed": 0, "runs": 0})
        print(f"{b:<28} {row}   |  {f}/{n:<4}  {str(upper95(f, n)):>8}   | {s['failed']}/{s['runs']} failed")
    for b in sorted(cells):
        for w, c in sorted(cells[b].items()):
            if "cases" in c or "torn_batches" in c or "handshake_rounds" in c:
                print(f"  {b} / {w}: " + ", ".join(f"{k} {c[k]}" for k in ("cases_failed", "cases", "torn_batches", "handshake_rounds") if k in c))
    for b, d in faults.items():
        for text, where in d.items():
            print(f"  build {b}: '{text}' in {len(where)} run(s): {where[:6]}")


if __name__ == "__main__":
    main()

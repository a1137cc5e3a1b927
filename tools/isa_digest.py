#!/usr/bin/env python3
"""Digest of the gfx950 ISA of every kernel in a `hipcc -S --cuda-device-only` listing (drone_amd/csrc: `make asm`).

For each `.amdhsa_kernel`: the sha256 of its instruction stream (labels renumbered per function, comments and the symbol's
own name dropped) and of its kernel descriptor (registers, LDS, kernarg size, preload length ...). Two listings with equal
digests launch the same machine code. Used by tests/test_isa_frozen.py to hold the round-6 source pruning to "not one
instruction of a shipped kernel moved" (VERDICT r5 item 4), and by hand to see which instantiations a change touches:

    python3 tools/isa_digest.py drone_amd/csrc/drone_kernels.s > new.json
    python3 tools/isa_digest.py --diff tests/golden/isa_frozen.json new.json
"""
import hashlib
import json
import re
import subprocess
import sys

CXXFILT = "c++filt"  # binutils (the ROCm image ships no llvm-cxxfilt)


def canonical(names):
    """mangled kernel symbols -> `kernel<template arguments>` without namespaces or the parameter list: the key survives a
    change of the kernel's ARGUMENTS (which the descriptor's kernarg size shows) but not of what it is instantiated with"""
    out = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    res = {}
    for m, d in zip(names, out):
        d = re.sub(r"^void ", "", d).replace("drone::(anonymous namespace)::", "").replace("drone::", "")
        depth, cut = 0, len(d)
        for i, ch in enumerate(d):  # the parameter list opens at the first '(' outside the template brackets
            if ch == "<": depth += 1
            elif ch == ">": depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        d = d[:cut]
        # round 6: the reset / step / rollout kernels gained a trailing template argument, PEER (the stop word of the peer-store
        # exchange). PEER = false IS the kernel of the rounds before — keyed as it was, so that its ISA can be held to the measured
        # one; PEER = true is a kernel of its own, keyed `..._peer_kernel<...>`
        arity = {"drone_reset_kernel": 1, "drone_rollout_kernel": 2, "drone_step_kernel": 4}
        fam = d.split("<")[0]
        if fam in arity and d.endswith(">"):
            args = [a.strip() for a in d[len(fam) + 1:-1].split(",")]
            if len(args) == arity[fam] + 1:
                d = (fam if args[-1] == "false" else fam.replace("_kernel", "_peer_kernel")) + "<" + ", ".join(args[:-1]) + ">"
        res[m] = d
    return res


def describe(desc):
    """the kernel descriptor: its hash without the kernarg size (an argument dropped from a kernel's signature changes nothing
    else), and the fields worth reading in a diff"""
    d = dict((l.split(None, 1) + [""])[:2] for l in desc)
    rest = "\n".join(f"{k} {v}" for k, v in sorted(d.items()) if k != ".amdhsa_kernarg_size")
    pick = {"kernarg_size": ".amdhsa_kernarg_size", "vgprs": ".amdhsa_next_free_vgpr", "sgprs": ".amdhsa_next_free_sgpr", "lds": ".amdhsa_group_segment_fixed_size",
            "scratch": ".amdhsa_private_segment_fixed_size", "kernarg_preload": ".amdhsa_user_sgpr_kernarg_preload_length"}
    return {"descriptor_sha256": hashlib.sha256(rest.encode()).hexdigest(), **{k: int(d.get(f, "0") or 0) for k, f in pick.items()}}


def digest(path):
    kernels = {}
    name, body, desc, in_body, in_desc = None, [], [], False, False
    label = re.compile(r"\.LBB\d+_")
    for raw in open(path, errors="replace"):
        line = raw.split(";", 1)[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"\s*\.type\s+(\S+),@function", line)
        if m:
            name, body, desc, in_body = m.group(1), [], [], False
            continue
        if name and line.strip() == name + ":":
            in_body = True
            continue
        if in_body and re.match(r"\.Lfunc_end\d+:", line.strip()):
            in_body = False
            if desc:  # a kernel (device functions have no descriptor: all of ours are inlined)
                kernels[name] = {"isa_sha256": hashlib.sha256("\n".join(body).encode()).hexdigest(), **describe(desc),
                                 "instructions": sum(1 for l in body if l.startswith("\t") and not l.lstrip().startswith("."))}
            name = None
            continue
        if in_body:
            s = line.strip()
            if s.startswith(".amdhsa_kernel"):
                in_desc = True
                continue
            if s.startswith(".end_amdhsa_kernel"):
                in_desc = False
                continue
            if in_desc:
                desc.append(s)
                continue
            if s.startswith((".section", ".p2align 8", ".protected", ".globl", ".weak")) and not body:
                continue
            if s.startswith(".section"):  # the descriptor's detour through .rodata and back
                continue
            body.append(label.sub(".LBB_", line).replace(name, "@SELF"))
    names = canonical(sorted(kernels))
    assert len(set(names.values())) == len(names), "two kernels share a canonical name"
    return {names[k]: v for k, v in kernels.items()}


def diff(a, b):
    A, B = json.load(open(a)), json.load(open(b))
    A, B = A.get("kernels", A), B.get("kernels", B)
    rc = 0
    for k in sorted(set(A) | set(B)):
        if k not in B:
            print("gone   ", k); rc = 1
        elif k not in A:
            print("new    ", k)
        elif A[k] != B[k]:
            print("CHANGED", k, A[k]["instructions"], "->", B[k]["instructions"]); rc = 1
    print(f"{len(A)} kernels before, {len(B)} after, {'identical where both exist' if rc == 0 else 'DIFFERENT'}")
    return rc


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--diff":
        sys.exit(diff(sys.argv[2], sys.argv[3]))
    json.dump({"kernels": digest(sys.argv[1])}, sys.stdout, indent=0, sort_keys=True)
    print()

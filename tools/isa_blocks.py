#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in drone_kernels.s (make -C drone_amd/csrc asm), with the
branch structure, to price the fused rollout's loop statically:
   python tools/isa_blocks.py drone_amd/csrc/drone_kernels.s rollout_kernelILi0"""
import collections
import re
import sys

text = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(text) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
end = next(i for i in range(start, len(text)) if text[i].strip().startswith("s_endpgm"))
blocks = collections.OrderedDict()
cur = "entry"
blocks[cur] = []
for l in text[start + 1:end + 1]:
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        cur = m.group(1)
        blocks[cur] = []
        continue
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    blocks[cur].append(t)
    if t.startswith(("s_cbranch", "s_branch")):  # fall-through code after a branch is its own block
        cur = cur.split("+")[0] + "+" + str(sum(1 for k in blocks if k.split("+")[0] == cur.split("+")[0]))
        blocks[cur] = []


def kind(op):
    if op.startswith(("v_fma", "v_fmac", "v_pk_fma")):
        return "fma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "mem"


tot = collections.Counter()
for name, ins in blocks.items():
    c = collections.Counter(kind(i.split()[0]) for i in ins)
    br = [i for i in ins if i.startswith(("s_cbranch", "s_branch"))]
    print(f"{name:12s} n={len(ins):4d} valu={c['valu'] + c['fma']:4d} (fma {c['fma']:3d}) salu={c['salu']:3d} mem={c['mem']:3d}  " + "; ".join(b.replace("s_cbranch_", "").replace("s_branch", "br") for b in br))
    tot += c
print("total", dict(tot))

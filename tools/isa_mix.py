#!/usr/bin/env python3
"""Instruction mix of a kernel in drone_kernels.s (make -C drone_amd/csrc asm):
   python tools/isa_mix.py drone_amd/csrc/drone_kernels.s rollout_kernelILi0"""
import collections
import re
import sys

text = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(text) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
end = next(i for i in range(start, len(text)) if text[i].strip().startswith("s_endpgm"))
ins = []
for l in text[start + 1:end + 1]:
    l = l.strip()
    if not l or l.startswith((".", ";")) or l.endswith(":"):
        continue
    ins.append(l.split()[0])
c = collections.Counter(ins)
groups = collections.Counter()
for k, v in c.items():
    g = ("v_pk" if k.startswith("v_pk_") else "v_fma" if k.startswith(("v_fma", "v_fmac")) else
         "v_mul_f32" if k.startswith("v_mul_f32") else "v_add/sub_f32" if k.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32")) else
         "v_other" if k.startswith("v_") else "salu" if k.startswith("s_") else "mem" if k.startswith(("global_", "ds_", "buffer_", "flat_", "scratch_")) else "other")
    groups[g] += v
print(pat, "static instructions:", len(ins), dict(groups))
print([(k, v) for k, v in c.most_common(40)])

#!/usr/bin/env python3
"""Price one env-step of the fused rollout kernel from its ISA (make -C drone_amd/csrc asm) and the measured
SQ_INSTS_VALU: writes profiles/rollout_valu.json, which bench.py reads for the config-5 roofline.
  flop = 2 per v_fma/v_fmac, 1 per f32 add / sub / mul, 0 for everything else (integer hash, cvt, compares, selects,
  med3, moves). Blocks inside the step loop count once; the episode-end blocks count with the measured share of
  wave-steps that execute them, derived from the counter: share = (measured - always) / (episode-end VALU).
  usage: tools/rollout_flops.py <task name> <kernel pattern> <measured VALU per wave-step> <ns per VALU issue> <source note>"""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
task, pat, measured, ns_issue, note = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4]), sys.argv[5]
text = open(os.path.join(ROOT, "drone_amd", "csrc", "drone_kernels.s")).read().split("\n")
start = next(i for i, l in enumerate(text) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
end = next(i for i in range(start, len(text)) if text[i].strip().startswith(".Lfunc_end"))
blocks, cur = collections.OrderedDict(), "entry"
blocks[cur] = []
for l in text[start + 1:end]:
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        cur = m.group(1)
        blocks[cur] = []
        continue
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    blocks[cur].append(t)
    if t.startswith(("s_cbranch", "s_branch")):
        cur = cur.split("+")[0] + "+" + str(sum(1 for k in blocks if k.split("+")[0] == cur.split("+")[0]))
        blocks[cur] = []


def mix(ins):
    valu = fma = addmul = 0
    for i in ins:
        op = i.split()[0]
        if not op.startswith("v_"):
            continue
        valu += 1
        if op.startswith(("v_fma_f32", "v_fmac_f32")):
            fma += 1
        elif op.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32")):
            addmul += 1
    return valu, fma, addmul


names = list(blocks)
# the step loop: from the block after the loop header's back-edge target to the block that branches back
back = None
for n, ins in blocks.items():
    for i in ins:
        m = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)", i) or re.match(r"s_branch (\.LBB\d+_\d+)", i)
        if m and names.index(m.group(1).split("+")[0]) < names.index(n) and len(blocks[m.group(1)]) <= 12:
            back = (m.group(1), n)  # outer loop: its header is a short bookkeeping block
lo, hi = names.index(back[0]), names.index(back[1])
always = ends = [0, 0, 0]
always, ends = [0, 0, 0], [0, 0, 0]
for n in names[lo:hi + 1]:
    v = mix(blocks[n])
    # blocks entered through an execz-guard are the episode-end path (reset + log fold)
    prev = names[names.index(n) - 1]
    guarded = "+" in n and any(i.startswith("s_cbranch_execz") for i in blocks[prev][-1:])
    tgt = ends if guarded else always
    for k in range(3):
        tgt[k] += v[k]
share = max(0.0, min(1.0, (measured - always[0]) / ends[0])) if ends[0] else 0.0
flop = 2 * (always[1] + share * ends[1]) + (always[2] + share * ends[2])
fma_share = (always[1] + share * ends[1]) / (always[0] + share * ends[0])
out_path = os.path.join(ROOT, "profiles", "rollout_valu.json")
try:
    out = json.load(open(out_path))
except (OSError, ValueError):
    out = {}
out[task] = {"valu_per_wave_step": measured, "static_valu_always": always[0], "static_valu_episode_end": ends[0], "episode_end_share_of_wave_steps": round(share, 3),
             "flop_per_env_step": round(flop, 1), "fma_share": round(fma_share, 3), "ns_per_valu_issue": ns_issue, "source": note}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out[task]))

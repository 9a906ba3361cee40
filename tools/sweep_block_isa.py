#!/usr/bin/env python3
"""Instruction mix of the pipelined contact-row block of the sweep loop (pgs_dv) of one kernel, from a -save-temps .s file
(tools/check_dpp_masks.py leaves them in scratch/isa).  The block is found by its 16 friction clamps (v_med3_f32).
  python tools/sweep_block_isa.py scratch/isa/rex_step_base-hip-amdgcn-amd-amdhsa-gfx950.s ILi16ELb0E [--dump]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
m = re.search(r"^(\S*kernel%s\S*):" % re.escape(sys.argv[2]), s, re.M)
k = s[m.start():s.index(".Lfunc_end", m.start())]
lines = [l.strip().split(";")[0].strip() for l in k.split("\n")]
lines = [l for l in lines if l]
idx = [i for i, l in enumerate(lines) if l.startswith("v_med3_f32")]
cl = []
for i in idx:
    if cl and i - cl[-1][-1] < 60:
        cl[-1].append(i)
    else:
        cl.append([i])
for c in cl:
    if len(c) != 16:
        continue
    a = max([i for i in range(c[0] - 170, c[0]) if lines[i].endswith(":")] or [c[0] - 170])
    b = min([i for i in range(c[-1], c[-1] + 80) if lines[i].endswith(":") or lines[i].startswith(("s_branch", "s_cbranch"))] or [c[-1] + 40])
    seg = [l for l in lines[a:b] if not l.endswith(":")]
    cc = Counter(x.split()[0] for x in seg)
    print(f"rows block at {a}..{b}: {len(seg)} instructions;", ", ".join(f"{n} {k_}" for k_, n in cc.most_common(12)))
    if "--dump" in sys.argv:
        print("\n".join(seg))

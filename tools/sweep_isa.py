#!/usr/bin/env python3
"""Instruction mix of the lane-parallel sweep loop (the loop with 24 row_half_mirror adds) in scratch/isa/*.s
(run tools/kstat.sh first).  usage: tools/sweep_isa.py [EPW=4] [ARM=0]"""
import collections
import re
import sys

epw = sys.argv[1] if len(sys.argv) > 1 else "4"
arm = sys.argv[2] if len(sys.argv) > 2 else "0"
s = open('scratch/isa/rex_step_%s-hip-amdgcn-amd-amdhsa-gfx950.s' % ('arm' if arm == '1' else 'base')).read()
m = re.search(r'^(\S*rex_step_kernelILi%sELb%sE\S*):' % (epw, arm), s, re.M)
k = s[m.start():s.index('.Lfunc_end', m.start())].split('\n')
lab = {}
for i, l in enumerate(k):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        lab[mm.group(1)] = i
pat = 'row_half_mirror' if int(epw) <= 8 else 'quad_perm:[2,3,0,1]'
best = None
for i, l in enumerate(k):
    mm = re.search(r's_cbranch\w*\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in lab and lab[mm.group(1)] < i:
        a = lab[mm.group(1)]
        n = sum(1 for x in k[a:i] if pat in x and 'v_add_f32' in x)
        if n == 24 and (best is None or i - a < best[1] - best[0]):
            best = (a, i)
a, i = best
seg = [l.strip() for l in k[a:i + 1] if l.strip() and not l.strip().startswith(';')]
c = collections.Counter(x.split()[0] for x in seg if not x.endswith(':'))
print(len(seg), "instructions in the sweep loop")
print(c.most_common(30))

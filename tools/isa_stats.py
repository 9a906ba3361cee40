#!/usr/bin/env python3
"""Instruction histogram + loop summary of one kernel in a hipcc -save-temps .s file."""
import re
import sys
from collections import Counter

path, kname = sys.argv[1], sys.argv[2]
s = open(path).read()
m = re.search(r'^(\S*%s\S*):' % re.escape(kname), s, re.M)
start = m.start()
end = s.index('.Lfunc_end', start)
k = s[start:end]
lines = [l.strip() for l in k.split('\n') if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')
         or re.match(r'^\.LBB\S+:', l.strip() or '')]
lines = [l for l in lines if l]
print("total instr lines", len(lines))
c = Counter(l.split()[0] for l in lines if not l.endswith(':'))
for mn, n in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print(n, mn)
labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(':')}
for i, l in enumerate(lines):
    mm = re.match(r's_cbranch_\w+\s+(\S+)', l) or re.match(r's_branch\s+(\S+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        body = lines[labels[mm.group(1)]:i]
        cc = Counter(x.split()[0] for x in body if not x.endswith(':'))
        tot = sum(cc.values())
        print("LOOP", mm.group(1), "instrs", tot, "scratch", sum(v for k2, v in cc.items() if 'scratch' in k2),
              "ds", sum(v for k2, v in cc.items() if k2.startswith('ds_')),
              "accvgpr", sum(v for k2, v in cc.items() if 'accvgpr' in k2),
              "valu", sum(v for k2, v in cc.items() if k2.startswith('v_')),
              "pk", sum(v for k2, v in cc.items() if k2.startswith('v_pk')),
              "salu", sum(v for k2, v in cc.items() if k2.startswith('s_')))

#!/bin/bash
# compile rexsim.hip with -save-temps into scratch/isa and print loop/instruction stats of rex_step_kernel
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch/isa && cd scratch/isa
hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps -Rpass-analysis=kernel-resource-usage -c ../../rex_gym_amd/csrc/rexsim.hip -o /dev/null 2>&1 | grep -A9 "rex_step_kernel" | grep -E "error|VGPRs:|AGPRs|Scratch|Spill" || true
cd ../..
python - <<'PY'
import re
from collections import Counter
s=open('scratch/isa/rexsim-hip-amdgcn-amd-amdhsa-gfx950.s').read()
start=s.index('_ZN3rex15rex_step_kernelENS_6DevCfgEPfPKfS3_S1_S1_PhS1_:')
end=s.index('.Lfunc_end', start)
k=s[start:end].split('\n')
allb=[x.strip() for x in k if x.strip() and not x.strip().startswith(';') and not x.strip().startswith('.')]
print("kernel total instrs", len(allb))
lab={}
for i,l in enumerate(k):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: lab[m.group(1)]=i
for i,l in enumerate(k):
    m=re.search(r'\bs_c?branch\w*\s+(\.LBB\d+_\d+)',l)
    if m and m.group(1) in lab and lab[m.group(1)]<i:
        a=lab[m.group(1)]
        body=[x.strip() for x in k[a:i] if x.strip() and not x.strip().startswith(';') and not x.strip().startswith('.')]
        if len(body)<100: continue
        cc=Counter(x.split()[0] for x in body)
        print(m.group(1),"n",len(body),"ds_read",sum(v for q,v in cc.items() if q.startswith('ds_read')),"ds_write",sum(v for q,v in cc.items() if q.startswith('ds_write')),"scratch",sum(v for q,v in cc.items() if 'scratch' in q),"acc",sum(v for q,v in cc.items() if 'accvgpr' in q),"pk",sum(v for q,v in cc.items() if q.startswith('v_pk')), "valu",sum(v for q,v in cc.items() if q.startswith('v_')), "waitcnt", cc.get('s_waitcnt',0))
PY

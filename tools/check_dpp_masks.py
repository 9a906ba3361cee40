#!/usr/bin/env python3
"""Developer check: no DPP lane shift of the row finishing (row_shr:1, rex_device.h physics_substep) sits under a lane mask.

A DPP move reads 0 from a lane the EXEC mask has switched off, and the compiler is free to turn `cond ? f(dpp(x)) : 0`
into a branch around the move (it did: the couplings of consecutive contact rows came out wrong for the lanes whose
neighbour failed the condition, and the sweeps converged more slowly -- same fixed point, 53 sweeps instead of 42).
Compiles every variant group with -save-temps into scratch/isa and scans each kernel: between the first and the last
row_shr:1 of a kernel there must be no instruction that writes EXEC.
  python tools/check_dpp_masks.py [group ...]
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rex_gym_amd", "csrc")
OUT = os.path.join(ROOT, "scratch", "isa")
os.makedirs(OUT, exist_ok=True)
groups = sys.argv[1:] or ["step_base", "step_arm", "step_mixed_base", "step_mixed_arm", "step_body", "settle_base", "settle_arm"]


def compile_group(g):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-save-temps", "-I", CSRC, "-c",
                    os.path.join(CSRC, f"rex_{g}.hip"), "-o", os.devnull], cwd=OUT, check=True, stderr=subprocess.DEVNULL)
    return g


with ThreadPoolExecutor(4) as ex:
    list(ex.map(compile_group, groups))
bad = 0
for g in groups:
    s = open(os.path.join(OUT, f"rex_{g}-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    for m in re.finditer(r"^(_ZN3rex\S*kernel\S*):", s, re.M):
        body = s[m.start():s.index(".Lfunc_end", m.start())]
        lines = [l.strip().split(";")[0].strip() for l in body.split("\n")]
        lines = [l for l in lines if l]
        idx = [i for i, l in enumerate(lines) if "row_shr:1" in l]
        if not idx:
            continue
        masked = [l for l in lines[idx[0]:idx[-1]] if re.match(r"s_\w+\s+exec", l) or "saveexec" in l]
        print(f"{g:16s} {m.group(1)[:60]:60s} shifts {len(idx):3d}  exec writes among them: {len(masked)}")
        bad += len(masked)
sys.exit(1 if bad else 0)

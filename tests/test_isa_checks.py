"""Checks on the compiled kernels that need no GPU: hipcc cross-compiles gfx950 here (half a minute for the three variant
groups compiled below; tools/check_dpp_masks.py alone covers all seven).

1. The row finishing of a substep (rex_device.h, physics_substep) hands rows between neighbouring lanes with DPP shifts.
   A DPP move reads 0 from a lane the EXEC mask has switched off, and the compiler is free to put `cond ? f(dpp(x)) : 0`
   behind a branch -- it did once (round 3): every parity test still passed within tolerance, the sweeps just converged
   more slowly.  No EXEC write may sit among those shifts.
2. The shape of the sweep loop's contact-row block, which is what a step costs (a lone wave pays 5
   cycles per instruction of it): its instruction count, no row slice read back from AGPRs, no per-row register copies.
   A compiler or source change that loses one of the three shows up here instead of as a few per cent on the GPU.
"""
import os
import re
import shutil
import subprocess
import sys
from collections import Counter

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = ("step_base", "step_arm", "step_mixed_arm")
pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")


@pytest.fixture(scope="module")
def compiled():
    env = dict(os.environ)
    env["PATH"] = env.get("PATH", "") + ":/opt/rocm/bin"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_masks.py"), *GROUPS],
                       capture_output=True, text=True, env=env, timeout=900)
    return r


def test_no_dpp_shift_of_the_row_finishing_sits_under_a_lane_mask(compiled):
    assert compiled.returncode == 0, compiled.stdout + compiled.stderr
    lines = [l for l in compiled.stdout.splitlines() if "shifts" in l]
    assert len(lines) >= 9, compiled.stdout            # 3 groups x (4 / 8 / 16 envs per wave)
    assert all(l.rstrip().endswith(": 0") for l in lines), compiled.stdout


def _contact_blocks(group, kernel):
    """Instruction mix of every contact-row block (found by its 16 friction clamps) of one kernel of a compiled group."""
    s = open(os.path.join(ROOT, "scratch", "isa", f"rex_{group}-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    m = re.search(r"^(\S*kernel%s\S*):" % re.escape(kernel), s, re.M)
    body = s[m.start():s.index(".Lfunc_end", m.start())]
    lines = [l.strip().split(";")[0].strip() for l in body.split("\n")]
    lines = [l for l in lines if l]
    clamps = [i for i, l in enumerate(lines) if l.startswith("v_med3_f32")]
    clusters = []
    for i in clamps:
        if clusters and i - clusters[-1][-1] < 60:
            clusters[-1].append(i)
        else:
            clusters.append([i])
    out = []
    for c in clusters:
        if len(c) != 16:
            continue
        a = max([i for i in range(c[0] - 170, c[0]) if lines[i].endswith(":")] or [c[0] - 170])
        b = min([i for i in range(c[-1], c[-1] + 80) if lines[i].endswith(":") or lines[i].startswith(("s_branch", "s_cbranch"))] or [c[-1] + 40])
        out.append(Counter(l.split()[0] for l in lines[a:b] if not l.endswith(":")))
    return out


@pytest.mark.parametrize("epw", [4, 8, 16])
def test_contact_row_block_of_the_base_kernels_keeps_its_shape(compiled, epw):
    assert compiled.returncode == 0, compiled.stdout + compiled.stderr
    blocks = _contact_blocks("step_base", f"ILi{epw}ELb0ELb0ELb0E")
    assert len(blocks) == 2, "the base kernels run their sweeps in pairs (two impulse sets): two row blocks"
    for mix in blocks:
        n = sum(mix.values())
        assert n <= 325, (n, mix)                        # 24 rows x 13 instructions + the first row's sum (measured: 316-318)
        assert mix.get("v_accvgpr_read_b32", 0) == 0, mix    # the row slices live in VGPRs
        assert mix.get("v_mov_b32_e32", 0) <= 6, mix         # no impulse copies at the end of a sweep
        assert mix.get("v_add_f32_dpp", 0) == (72 if epw <= 8 else 48), mix   # 3 (8 lanes per env) or 2 (4 lanes) steps per group sum


def test_fused_actor_kernels_run_their_layers_on_the_matrix_cores_with_the_reads_a_chunk_ahead():
    """The `_pol` instantiations (rex_step_policy / rex_step_segment_policy, csrc/rex_policy.h), compiled here for the base group:
    every kernel carries the actor twice (weights in LDS / streamed) x two ReLU layers of 4x4x1 16-block MFMAs; the weights and
    activations arrive as b128 reads; the DPP sums of the mean layer -- like every cross-lane instruction of the step -- do not sit
    among EXEC writes of the row finishing (the tool's first check) and the kernels do not spill to scratch."""
    env = dict(os.environ)
    env["PATH"] = env.get("PATH", "") + ":/opt/rocm/bin"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_masks.py"), "--pol", "step_base"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    s = open(os.path.join(ROOT, "scratch", "isa_pol", "rex_step_base-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    kernels = list(re.finditer(r"^(_ZN3rex15rex_step_kernel\S*):", s, re.M))
    assert len(kernels) == 3                                           # 4 / 8 / 16 envs per wave
    for m, epw in zip(kernels, (4, 8, 16)):
        assert f"ILi{epw}E" in m.group(1) and m.group(1).count("Lb1E") == 2      # <EPW, base, ..., SEG, POLICY>
        body = s[m.start():s.index(".Lfunc_end", m.start())]
        mfma = len(re.findall(r"v_mfma_f32_4x4x1_16b_f32", body))
        # per layer call: KC quads x 2 chunks consumed per loop trip x 8 MFMAs per quad and env group; 2 layers x (LDS, streamed)
        assert mfma >= 2 * 2 * 8 * (epw // 4), (epw, mfma)
        assert len(re.findall(r"ds_read_b128", body)) >= 16 and len(re.findall(r"global_load_dwordx4", body)) >= 8, epw
        meta = re.search(r"\.name:\s*%s\n(?:.*\n){0,40}?\s*\.private_segment_fixed_size:\s*(\d+)" % re.escape(m.group(1)), s)
        if meta:
            assert int(meta.group(1)) == 0, (epw, meta.group(1))

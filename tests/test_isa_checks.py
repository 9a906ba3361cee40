"""Checks on the compiled kernels that need no GPU: hipcc cross-compiles gfx950 here.

The row finishing of a substep (rex_device.h, physics_substep) hands rows between neighbouring lanes with DPP shifts.  A DPP
move reads 0 from a lane the EXEC mask has switched off, and the compiler is free to put `cond ? f(dpp(x)) : 0` behind a
branch -- it did once (round 3): every parity test still passed within tolerance, the sweeps just converged more slowly.
tools/check_dpp_masks.py scans the ISA of a variant group for EXEC writes among those shifts; this test runs it on three of
the seven groups (half a minute; the tool alone covers them all).
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_dpp_shift_of_the_row_finishing_sits_under_a_lane_mask():
    env = dict(os.environ)
    env["PATH"] = env.get("PATH", "") + ":/opt/rocm/bin"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_masks.py"), "step_base", "step_arm", "step_mixed_arm"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if "shifts" in l]
    assert len(lines) >= 9, r.stdout            # 3 groups x (4 / 8 / 16 envs per wave)
    assert all(l.rstrip().endswith(": 0") for l in lines), r.stdout

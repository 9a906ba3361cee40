#!/usr/bin/env python3
"""One-step error of the HIP path along an fp64 oracle rollout (developer diagnostic, needs a GPU): every control step the
kernel steps once from the oracle's state and the joint angles are compared -- separates the per-step error of the
kernels from its amplification by the contact dynamics.   python tools/onestep_error.py [clock=1.5] [n=1024] [steps=200]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import orclib
import parity_window as pw
from helpers import numeric_to_product_state, product_state_to_numeric
from rex_gym_amd import RexBatchEnv

clock = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
env = RexBatchEnv(n, task="walk", signal_type="ik", seed=23, gait_clock_scale=clock)
orc = pw.oracle_for(env)
orc.o.lib.orc_set_threads(16)
env.reset(); orc.reset()
rng = np.random.RandomState(23)
for k in range(steps):
    a = rng.uniform(-0.4, 0.4, (n, 2)).astype(np.float32)
    st = orc.get_state()
    env.state.copy_(numeric_to_product_state(st, torch, env.state.device))
    _, _, _, info = env.step(torch.as_tensor(a, device="cuda"))
    _, _, _, ocmd = orc.step(a)
    ps, os_ = product_state_to_numeric(env.state), orc.get_state()
    eq = np.abs(ps[13:25] - os_[13:25]).max(0)
    ev = np.abs(ps[25:37] - os_[25:37]).max(0)
    ec = np.abs(info["action"].cpu().numpy() - ocmd).max(1)
    if eq.max() > 1e-4 or k in (86, 87, 88, 173, 174, 175):
        w = orclib
        print(f"  step {k}: BEFORE oracle flags {int(st[w.S_FLAGS, 0])} lastt {st[w.S_LASTT, 0]} steps {st[w.S_STEPS, 0]} phi {st[w.S_PHI, 0]:.6f} | AFTER oracle flags {int(os_[w.S_FLAGS, 0])} lastt {os_[w.S_LASTT, 0]} "
              f"phi {os_[w.S_PHI, 0]:.6f} | AFTER hip flags {int(ps[w.S_FLAGS, 0])} lastt {ps[w.S_LASTT, 0]} phi {ps[w.S_PHI, 0]:.6f}")
        print("   cmd hip", np.round(info["action"].cpu().numpy()[0], 5).tolist())
        print("   cmd orc", np.round(ocmd[0], 5).tolist(), "action", a[0].tolist())
    if k % 10 == 0 or eq.max() > 1e-4:
        print(f"step {k:3d}: one-step |dq| median {np.median(eq):.1e} p99 {np.percentile(eq, 99):.1e} max {eq.max():.1e} (env {eq.argmax()}); "
              f"|dqd| median {np.median(ev):.1e} max {ev.max():.1e}; |dcmd| max {ec.max():.1e}")

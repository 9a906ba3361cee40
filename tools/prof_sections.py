#!/usr/bin/env python3
"""Cycle breakdown of one physics substep inside rex_step_kernel (developer tool, needs a GPU).

Builds the library with -DREX_PROF (clock64() stamps around the sections of physics_substep, accumulated per
workgroup by lane 0) into scratch/librexsim_prof.so, runs walk-IK at N envs and prints cycles per substep for:
leg factorisation, base Cholesky, row finishing, PGS sweeps, back-substitution + integration.
  python tools/prof_sections.py [N=4096] [--arm] [--task=standup|poses|...] [--roll=R] [--body=0|1] [--rebuild]
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.environ.get("REX_PROF_LIB") or os.path.join(ROOT, "scratch", "librexsim_prof.so")   # REX_PROF_LIB: a prebuilt -DREX_PROF library (A/B)
os.makedirs(os.path.dirname(so), exist_ok=True)
import rex_gym_amd.build as b
if not os.path.exists(so) or "--rebuild" in sys.argv:
    b.build(force=True, lib_path=so, defines=["-DREX_PROF"], unity=True)   # one translation unit: the counters are one device global
b.LIB_PATH = so
import torch
from rex_gym_amd import RexBatchEnv, _lib

L = _lib.lib()
L.rex_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.rex_debug_prof2.argtypes = [ctypes.c_void_p, ctypes.c_int]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 4096
mark = "arm" if "--arm" in sys.argv else "base"
task = next((a.split("=")[1] for a in sys.argv if a.startswith("--task=")), "walk")
extra = {}
for a in sys.argv:
    if a.startswith("--roll="):
        extra["base_roll"] = float(a.split("=")[1])          # poses: every env holds this roll (the self-collision regime at -0.74)
    if a.startswith("--body="):
        extra["body_contacts"] = bool(int(a.split("=")[1]))
env = RexBatchEnv(n, check_actions=False, task=task, signal_type="ol" if task == "standup" else "ik", seed=0, auto_reset=True, max_episode_steps=2000, mark=mark,
                  **extra)
env.reset()
_lo = torch.as_tensor(env.action_space.low, device="cuda").minimum(torch.as_tensor(env.action_space.high, device="cuda"))
acts = [torch.rand((n, env.action_dim), device="cuda") * (-2 * _lo) + _lo for _ in range(8)]


def window(steps, label):
    torch.cuda.synchronize()
    L.rex_debug_prof(None, 1)
    L.rex_debug_prof2(None, 1)
    for k in range(steps):
        env.step(acts[k % 8])
    torch.cuda.synchronize()
    out = np.zeros((1024, 10), np.int64)
    L.rex_debug_prof(out.ctypes.data, 0)
    out2 = np.zeros((1024, 16), np.int64)
    L.rex_debug_prof2(out2.ctypes.data, 0)
    out2 = out2[out[:, 4] > 0].astype(float)
    out = out[out[:, 4] > 0].astype(float)
    pgs, sw, tot, fin, sub, legs, chol = (out[:, i] for i in (0, 1, 2, 3, 4, 6, 7))
    rest = tot - pgs - fin - legs - chol
    fb = out[:, 5]
    print(f"{label}: {len(out)} workgroups sampled, sweeps/substep mean {np.mean(sw / sub):.1f} (slowest workgroup "
          f"{np.max(sw / sub):.1f}), cycles/sweep {pgs.sum() / sw.sum():.0f}, joint-limit rows in reach in "
          f"{100 * fb.sum() / sub.sum():.2f} % of the substeps")
    print("  cycles/step (whole kernel, per wave): mean %.0f, slowest workgroup %.0f; inside physics_substep %.0f" %
          (np.mean(out[:, 8] / out[:, 9]), np.max(out[:, 8] / out[:, 9]), np.mean(tot / out[:, 9])))
    print("  cycles/substep: total %.0f (slowest workgroup %.0f) = legs %.0f + base chol %.0f + finish rows %.0f + pgs %.0f "
          "+ back-subst/integrate %.0f" % (np.mean(tot / sub), np.max(tot / sub), np.mean(legs / sub), np.mean(chol / sub),
                                           np.mean(fin / sub), np.mean(pgs / sub), np.mean(rest / sub)))
    if out2[:, 7].sum() > 0:     # start times of the blocks of the last launch (10 ns ticks), its own length from the same counter
        st = out2[:, 7] - out2[:, 7].min()
        dur = out2[:, 6] / out[:, 9]
        print("  last launch: block starts spread over %.1f us (median %.1f, p90 %.1f); a block runs %.1f us (mean; max %.1f); "
              "latest end %.1f us after the first start" % (st.max() / 100, np.median(st) / 100, np.percentile(st, 90) / 100,
                                                          dur.mean() / 100, dur.max() / 100, (st + dur).max() / 100))
    if out2[:, 6].sum() > 0:
        print("  clock64() ticks per 10 ns tick of the constant 100 MHz counter over the kernel: %.2f  (a clock64 tick = %.3f ns)" %
              (out[:, 8].sum() / out2[:, 6].sum(), 10.0 * out2[:, 6].sum() / out[:, 8].sum()))
    if out2[:, 4].sum() > 0:
        print("  step = load + command (planner, IK) %.0f + substeps (motors, physics, observation ring) %.0f + reward / done / reset / observation / stores %.0f" %
              tuple(np.mean(out2[:, k] / out[:, 9]) for k in (3, 4, 5)))
    if out2[:, 1].sum() > 0:     # lane-group kernels: inside the pgs section
        su, lp, hb = (np.mean(out2[:, k] / sub) for k in range(3))
        print("  pgs = row couplings %.0f + sweep set-up %.0f + sweep loop %.0f (%.0f per sweep) + hand-back %.0f" %
              (np.mean(pgs / sub) - su - lp - hb, su, lp, out2[:, 1].sum() / sw.sum(), hb))
    # the workgroup that ends the launch: its own sections, and what its sweeps are made of (thread 0's sweeps: an env that has
    # converged no longer counts its wave's rows)
    w = int(np.argmax(tot / sub))
    print("  SLOWEST workgroup, cycles/substep: total %.0f = legs %.0f + base chol %.0f + finish rows %.0f + pgs %.0f + back-subst/integrate %.0f; "
          "%.1f sweeps/substep" % (tot[w] / sub[w], legs[w] / sub[w], chol[w] / sub[w], fin[w] / sub[w], pgs[w] / sub[w], rest[w] / sub[w], sw[w] / sub[w]))
    if out2[w, 10:14].sum() > 0:
        nsw = max(out2[w, 14], 1.0)
        c = out2[w, 10:14] / nsw
        print("    per sweep: %.1f link-box rows + %.1f joint-limit rows + 24 toe rows; cycles: limit rows %.0f, link-box normals %.0f, toe rows %.0f, "
              "link-box friction %.0f (sum %.0f of %.0f per sweep)" % (out2[w, 8] / nsw, out2[w, 9] / nsw, c[0], c[1], c[2], c[3], c.sum(), out2[w, 1] / nsw))
        nsa = max(out2[:, 14].sum(), 1.0)
        ca = out2[:, 10:14].sum(0) / nsa
        print("    all workgroups, per sweep: %.1f link-box rows + %.1f joint-limit rows; cycles: limit rows %.0f, link-box normals %.0f, toe rows %.0f, "
              "link-box friction %.0f" % (out2[:, 8].sum() / nsa, out2[:, 9].sum() / nsa, ca[0], ca[1], ca[2], ca[3]))


window(20, "first 20 steps after reset")
for k in range(1500):
    env.step(acts[k % 8])
window(100, "steady state (after 1500 steps)")

#!/usr/bin/env python3
"""Which legs have a toe point in reach, env by env and step by step (developer tool, needs a GPU; -DREX_PROF build).

The toe block of the sweep runs all 24 rows whatever is in reach; a wave could skip the rows of a leg that none of its envs has in
reach.  This census says how often that would happen: with the envs grouped as they are (neighbouring indices), and grouped by their
leg mask of the previous step (what a per-step or per-segment regrouping could do), for 4 and 16 envs per wave.
  python tools/leg_census.py [N=4096] [--task=walk|gallop|turn] [--signal=ik|ol]
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.environ.get("REX_PROF_LIB") or os.path.join(ROOT, "scratch", "librexsim_prof.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
import rex_gym_amd.build as b
if not os.path.exists(so) or "--rebuild" in sys.argv:
    b.build(force=True, lib_path=so, defines=["-DREX_PROF"], unity=True)
b.LIB_PATH = so
import torch
from rex_gym_amd import RexBatchEnv, _lib

L = _lib.lib()
L.rex_debug_legmask.argtypes = [ctypes.c_void_p, ctypes.c_int]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 4096
task = next((a.split("=")[1] for a in sys.argv if a.startswith("--task=")), "walk")
signal = next((a.split("=")[1] for a in sys.argv if a.startswith("--signal=")), "ik")
env = RexBatchEnv(n, check_actions=False, task=task, signal_type=signal, seed=0, auto_reset=True, max_episode_steps=2000)
env.reset()
lo = torch.as_tensor(np.minimum(env.action_space.low, env.action_space.high), device="cuda", dtype=torch.float32)
hi = torch.as_tensor(np.maximum(env.action_space.low, env.action_space.high), device="cuda", dtype=torch.float32)
g = torch.Generator(device="cuda"); g.manual_seed(1)
act = lambda: torch.rand((n, env.action_dim), device="cuda", generator=g) * (hi - lo) + lo
for k in range(1500):
    env.step(act())
    if k % 50 == 0 and k < 1300:
        env.reset(torch.randperm(n, device="cuda", generator=g)[: n // 32].to(torch.int32))
torch.cuda.synchronize()
buf = np.zeros(n, np.uint32)
L.rex_debug_legmask(buf.ctypes.data, n)
T = 300
masks = np.zeros((T, n), np.uint8)      # leg mask (4 bits) of the OR over the step's substeps
for t in range(T):
    env.step(act())
    torch.cuda.synchronize()
    L.rex_debug_legmask(buf.ctypes.data, n)
    pts = (buf >> 8) & 0xFF
    masks[t] = sum((((pts >> (2 * l)) & 3) != 0).astype(np.uint8) << l for l in range(4))
pop = np.array([bin(m).count("1") for m in range(16)])
print(f"{task}-{signal}, {n} envs, {T} steps: legs with a toe point in reach per env and step: mean {pop[masks].mean():.2f}; "
      "share of (env, step) by count 0..4: " + " ".join(f"{(pop[masks] == c).mean():.3f}" for c in range(5)))
for lag in (1, 5, 25, 50):
    print(f"  leg mask unchanged after {lag:2d} steps: {(masks[lag:] == masks[:-lag]).mean():.3f}")


def union_legs(order, epw, t):
    m = masks[t][order]
    m = m[: (n // epw) * epw].reshape(-1, epw)
    u = np.bitwise_or.reduce(m, axis=1)
    return pop[u].mean()


for epw in (4, 16):
    asis = np.mean([union_legs(np.arange(n), epw, t) for t in range(50, T)])
    by_prev = np.mean([union_legs(np.argsort(masks[t - 1], kind="stable"), epw, t) for t in range(50, T)])
    by_25 = np.mean([union_legs(np.argsort(masks[t - (t % 25) - 1], kind="stable"), epw, t) for t in range(50, T)])
    by_100 = np.mean([union_legs(np.argsort(masks[max(t - (t % 100) - 1, 0)], kind="stable"), epw, t) for t in range(100, T)])
    print(f"  {epw:2d} envs per wave: legs some env of the wave has in reach (of 4): as the envs lie {asis:.2f}; sorted by the previous step's mask {by_prev:.2f}; "
          f"sorted once per 25 steps {by_25:.2f}; once per 100 steps {by_100:.2f}")

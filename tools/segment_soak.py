#!/usr/bin/env python3
"""Long bit-identity run of rex_step_segment against rex_step on the GPU: the same seed and the same actions (some far outside the
Box, every 7th step), auto-reset, thousands of steps in segments of several lengths; every observation, reward, done and the state
block after every segment are compared with torch.equal.  (tests/test_gpu_parity.py::test_segment_launch_is_bit_identical_to_single_steps
is the short form that runs in the suite.)   python tools/segment_soak.py [STEPS=3000]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from rex_gym_amd import RexBatchEnv

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
cases = [("walk", "ik", "plane", "base", 4096, {}), ("gallop", "ol", "plane", "base", 8192, {}), ("turn", "ik", "random", "base", 4096, {}),
         ("walk", "ik", "plane", "arm", 4096, {}), ("poses", "ik", "plane", "base", 4096, {}), ("mixed", "ik", "plane", "arm", 2048, dict(mass_scale_range=(0.8, 1.2), friction_range=(0.25, 0.625))),
         ("walk", "ik", "plane", "base", 4096, dict(control_latency=0.02, pd_latency=0.003)), ("walk", "ik", "plane", "base", 70000, {})]
for task, signal, terrain, mark, n, kw in cases:
    mk = lambda: RexBatchEnv(n, check_actions=False, task=task, signal_type=signal, terrain_type=terrain, mark=mark, auto_reset=True, max_episode_steps=700, seed=7, **kw)
    one, seg = mk(), mk()
    lo = torch.as_tensor(one.action_space.low, device=one.device); hi = torch.as_tensor(one.action_space.high, device=one.device)
    lo, hi = torch.minimum(lo, hi), torch.maximum(lo, hi)
    assert torch.equal(one.reset(), seg.reset())
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    k, bad, dones, t_one, t_seg = 0, 0, 0, 0.0, 0.0
    lengths = [1, 2, 7, 25, 100, 333]
    while k < steps:
        T = lengths[(k // 50) % len(lengths)]
        a = lo + (hi - lo) * torch.rand((T, n, one.action_dim), device="cuda", generator=g)
        a[::7] = a[::7] * 3.0 - (hi - lo)          # far outside the Box every 7th step of a segment
        torch.cuda.synchronize(); t0 = time.time()
        so, sr, sd, _ = seg.step_segment(a)
        torch.cuda.synchronize(); t1 = time.time()
        for t in range(T):
            oo, orw, od, _ = one.step(a[t])
            bad += int(not (torch.equal(oo, so[t]) and torch.equal(orw, sr[t]) and torch.equal(od, sd[t])))
            dones += int(od.sum())
        torch.cuda.synchronize(); t_one += time.time() - t1; t_seg += t1 - t0
        bad += int(not torch.equal(one.state, seg.state))
        k += T
    print(f"{task}-{signal} {terrain} {mark} {n} {kw}: {k} steps in segments of {lengths}: mismatching steps / states {bad}; episodes ended {dones}; "
          f"finite {bool(torch.isfinite(seg.state[:13]).all())}; segment launches {n * k / t_seg / 1e6:.1f} M env-steps/s, per-step launches (with the comparisons) {n * k / t_one / 1e6:.1f} M", flush=True)
    one.close(); seg.close()

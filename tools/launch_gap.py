#!/usr/bin/env python3
"""Where do the microseconds between two step launches go?  Times N env.step() calls from Python against the same number
of launches issued back to back from C (REX_STEP_REPEAT), and an empty-kernel chain for the floor of the device."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(repeat, n=4096, steps=2000):
    os.environ["REX_STEP_REPEAT"] = str(repeat)
    import importlib
    import rex_gym_amd
    from rex_gym_amd import RexBatchEnv
    env = RexBatchEnv(n, check_actions=False, task="walk", signal_type="ik", seed=0, auto_reset=True, max_episode_steps=2000)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    pool = [torch.rand((n, 2), device="cuda", generator=g) * 0.8 - 0.4 for _ in range(16)]
    env.reset()
    for k in range(1500 // repeat):
        env.step(pool[k % 16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps // repeat):
        env.step(pool[k % 16])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    env.close()
    return dt / (steps // repeat * repeat) * 1e3


if __name__ == "__main__":
    r = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    print(f"REX_STEP_REPEAT={r}: {run(r):.4f} ms per launch")

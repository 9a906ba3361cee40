#!/usr/bin/env python3
"""Per-env solver sweep counts of a large steady-state batch (MI355X): what regrouping envs into waves by their sweep
count can buy.  Prints the histogram, the mean, and the mean over waves of the slowest env's count for the natural and
for the sorted grouping (a wave sweeps until its slowest env has converged)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("REX_REGROUP", "1")
from rex_gym_amd import RexBatchEnv, _lib  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    clock = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    env = RexBatchEnv(n, check_actions=False, task="walk", signal_type="ik", seed=0, auto_reset=True, max_episode_steps=2000, gait_clock_scale=clock)
    env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    pool = [torch.rand((n, 2), device="cuda", generator=g) * 0.8 - 0.4 for _ in range(16)]
    for k in range(1500):
        env.step(pool[k % 16])
    buf = torch.zeros(n, dtype=torch.int32, device="cuda")
    prev = None
    out = {}
    for k in range(4):
        env.step(pool[k % 16])
        _lib.check(env._L.rex_get_sweeps(env._h, buf.data_ptr(), env._stream_ptr()), "rex_get_sweeps")
        sw = buf.cpu().numpy().astype(np.int64)
        if prev is not None:
            out.setdefault("corr_with_previous_step", []).append(float(np.corrcoef(prev, sw)[0, 1]))
        prev = sw
    epw = 16
    nat = sw[: n // epw * epw].reshape(-1, epw).max(1).mean()
    srt = np.sort(sw)[::-1][: n // epw * epw].reshape(-1, epw).max(1).mean()
    byprev = sw[np.argsort(-buf.cpu().numpy())]  # (same array: placeholder for symmetry)
    out.update(envs=n, gait_clock=clock, mean_sweeps_per_step=float(sw.mean()), wave_max_natural=float(nat), wave_max_sorted=float(srt),
               hist=np.bincount(np.minimum(sw // 25, 12)).tolist(), hist_bin_width=25)
    print(json.dumps(out))
    env.close()


if __name__ == "__main__":
    main()

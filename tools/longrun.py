#!/usr/bin/env python3
"""Long run on the GPU: 300 000 steps of the benchmark workload in 50 000-step blocks (throughput drift, episode ends, non-finite state words)."""
import sys, time, torch
sys.path.insert(0, '.')
from rex_gym_amd import RexBatchEnv
n = 4096
env = RexBatchEnv(n, check_actions=False, task="walk", signal_type="ik", auto_reset=True, max_episode_steps=2000, seed=0)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
pool = [torch.rand((n, 2), device="cuda", generator=g) * 0.8 - 0.4 for _ in range(64)]
done_total = 0
for blk in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dsum = torch.zeros((), device="cuda")
    for k in range(50000):
        o, r, d, _ = env.step(pool[k % 64])
        if k % 100 == 0:
            dsum += d.sum()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    bad = int((~torch.isfinite(env.state[:37])).sum()) + int((~torch.isfinite(o)).sum())
    print(f"steps {blk*50000:>6}-{(blk+1)*50000:>6}: {dt/50000*1e3:.4f} ms/step = {n*50000/dt/1e6:.2f} M env-steps/s, episode ends per step (sampled) {float(dsum)/500:.2f}, non-finite words {bad}", flush=True)
env.close()

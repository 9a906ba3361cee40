"""Steady-state and first-steps ms/step of walk-IK (and mark arm with an argument) at a few batch sizes: the quick
check used between kernel changes.  Needs a GPU: python tools/timing.py [arm]"""
import sys, time; sys.path.insert(0,'.')
import torch
from rex_gym_amd import RexBatchEnv
def run(n, steps=300, warm=300, **kw):
    env = RexBatchEnv(n, check_actions=False, seed=0, auto_reset=True, max_episode_steps=2000, **kw)
    env.reset()
    lo=torch.as_tensor(env.action_space.low,device='cuda'); hi=torch.as_tensor(env.action_space.high,device='cuda')
    lo,hi=torch.minimum(lo,hi),torch.maximum(lo,hi)
    acts=[torch.rand((n,env.action_dim),device='cuda')*(hi-lo)+lo for _ in range(8)]
    for k in range(warm): env.step(acts[k%8])
    torch.cuda.synchronize(); t0=time.perf_counter()
    for k in range(steps): env.step(acts[k%8])
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/steps
    env.close()
    print(f"n={n} {kw}: {dt*1e3:.3f} ms/step  {n/dt/1e6:.2f} M env-steps/s", flush=True)
run(4096, warm=20, steps=200)
run(4096, warm=1500, steps=500)
run(65536, warm=100, steps=100)
run(262144, warm=60, steps=60)
if len(sys.argv)>1:
    run(4096, warm=300, steps=300, mark='arm')
    run(16384, warm=100, steps=100, mark='arm')

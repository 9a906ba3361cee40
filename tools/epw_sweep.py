"""Steady-state ms/step over batch size x envs-per-wave (REX_ENVS_PER_WAVE override): the data behind pick_envs_per_wave (rexsim.hip).  Needs a GPU."""
import sys, time, os; sys.path.insert(0,'.')
import torch
from rex_gym_amd import RexBatchEnv
def run(n, epw, steps=150, warm=400):
    os.environ['REX_ENVS_PER_WAVE']=str(epw)
    env = RexBatchEnv(n, check_actions=False, seed=0, auto_reset=True, max_episode_steps=2000)
    env.reset()
    acts=[torch.rand((n,2),device='cuda')*0.8-0.4 for _ in range(8)]
    for k in range(warm): env.step(acts[k%8])
    torch.cuda.synchronize(); t0=time.perf_counter()
    for k in range(steps): env.step(acts[k%8])
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/steps
    env.close()
    print(f"n={n} epw={epw}: {dt*1e3:.3f} ms/step  {n/dt/1e6:.2f} M env-steps/s", flush=True)
for n in (2048, 4096, 8192, 16384, 32768, 65536, 131072):
    for epw in (4, 8, 16, 64):
        if n/epw > 40000: continue
        run(n, epw, steps=100 if n>=32768 else 150, warm=300 if n>=32768 else 500)

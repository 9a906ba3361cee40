"""Soak run: every task / signal / terrain / mark family for 4 000 steps with auto-reset, every 7th action far outside the
Box; reports throughput (including the torch.rand action draws), non-finite counts, done rate, state extrema.
usage (GPU box): python tools/soak.py"""
import sys, time, torch
sys.path.insert(0,'.')
from rex_gym_amd import RexBatchEnv
for task,signal,terrain,mark,n in [("walk","ik","random","base",16384),("walk","ol","plane","arm",8192),("gallop","ik","random","base",16384),
                                   ("gallop","ol","plane","base",16384),("turn","ik","random","arm",8192),("turn","ol","plane","base",16384),
                                   ("standup","ol","plane","base",16384),("poses","ik","plane","base",16384)]:
    env=RexBatchEnv(n,task=task,signal_type=signal,terrain_type=terrain,mark=mark,auto_reset=True,max_episode_steps=1000,seed=7)
    lo=torch.as_tensor(env.action_space.low,device=env.device); hi=torch.as_tensor(env.action_space.high,device=env.device)
    lo,hi=torch.minimum(lo,hi),torch.maximum(lo,hi)
    env.reset(); t0=time.time(); dones=0; bad=0; rsum=0.0
    for k in range(4000):
        a=lo+(hi-lo)*torch.rand((n,env.action_dim),device=env.device)*(3.0 if k%7==0 else 1.0)-(hi-lo)*(1.0 if k%7==0 else 0.0)   # every 7th step: actions far outside the Box
        o,r,d,_=env.step(a)
        if k%200==0:
            bad+=int((~torch.isfinite(o)).sum())+int((~torch.isfinite(r)).sum())+int((~torch.isfinite(env.state[:13])).sum())
        dones+=int(d.sum()) if k%50==0 else 0
    torch.cuda.synchronize()
    st=env.state
    print(task,signal,terrain,mark,n,f"{4000*n/(time.time()-t0)/1e6:.1f} M steps/s nonfinite={bad} done-rate(sampled)={dones/(80*n):.4f} |z| max={float(st[2].abs().max()):.3f} |v| max={float(st[7:13].abs().max()):.1f}",flush=True)
    env.close()

#!/usr/bin/env python3
"""Soak run on the GPU: random actions (some far outside the Box), auto-reset; reports throughput, non-finite counts and state ranges."""
import sys, time, torch
sys.path.insert(0,'.')
from rex_gym_amd import RexBatchEnv
cases=[("poses","ik","plane","base",8192,{}),("poses","ik","plane","arm",4096,{}),("walk","ik","random","base",8192,dict(body_contacts=True)),
       ("gallop","ik","plane","base",8192,dict(body_contacts=True)),("standup","ol","plane","base",4096,dict(body_contacts=True)),
       ("walk","ik","plane","base",4096,dict(on_rack=True)),("turn","ik","plane","arm",2048,dict(on_rack=True))]
for task,signal,terrain,mark,n,kw in cases:
    env=RexBatchEnv(n, check_actions=False,task=task,signal_type=signal,terrain_type=terrain,mark=mark,auto_reset=True,max_episode_steps=1000,seed=7,**kw)
    lo=torch.as_tensor(env.action_space.low,device=env.device); hi=torch.as_tensor(env.action_space.high,device=env.device)
    lo,hi=torch.minimum(lo,hi),torch.maximum(lo,hi)
    env.reset(); t0=time.time(); dones=0; bad=0
    for k in range(3000):
        a=lo+(hi-lo)*torch.rand((n,env.action_dim),device=env.device)*(3.0 if k%7==0 else 1.0)-(hi-lo)*(1.0 if k%7==0 else 0.0)
        o,r,d,_=env.step(a)
        if k%200==0:
            bad+=int((~torch.isfinite(o)).sum())+int((~torch.isfinite(r)).sum())+int((~torch.isfinite(env.state[:13])).sum())
        dones+=int(d.sum()) if k%50==0 else 0
    torch.cuda.synchronize()
    st=env.state
    print(task,signal,terrain,mark,n,kw,f"{3000*n/(time.time()-t0)/1e6:.1f} M steps/s nonfinite={bad} done-rate(sampled)={dones/(60*n):.4f} z in [{float(st[2].min()):.3f}, {float(st[2].max()):.3f}] |v| max={float(st[7:13].abs().max()):.1f}",flush=True)
    env.close()

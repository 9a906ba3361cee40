#!/usr/bin/env python3
"""Play the reference's shipped PPO policies (rex_gym/policies/**/model.ckpt-N) on the batched HIP env and report what
happens -- a behavioural check of the restated physics: the policies were trained on PyBullet.

    python tools/play_shipped_policies.py --policies-root /path/to/rex_gym/policies [--num-envs 256] > report.jsonl

One JSON line per policy: episode return / length statistics over the batch, how episodes ended (fell, env goal, step
limit) and how far the base got.  Policies whose data shard is not shipped (gallop/*, walk/ol) are reported as missing.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rex_gym_amd import RexBatchEnv                                              # noqa: E402
from rex_gym_amd.agents.policy_player import SimplePPOPolicy                     # noqa: E402
from rex_gym_amd.agents.tf_checkpoint import CheckpointError                     # noqa: E402

POLICIES = [   # util/flag_mapper.py:1-10
    ("walk", "ik", "walk/ik/model.ckpt-2000000"), ("walk", "ol", "walk/ol/model.ckpt-4000000"),
    ("gallop", "ik", "gallop/ik/model.ckpt-2000000"), ("gallop", "ol", "gallop/ol/model.ckpt-4000000"),
    ("turn", "ik", "turn/ik/model.ckpt-2000000"), ("turn", "ol", "turn/ol/model.ckpt-2000000"),
    ("standup", "ol", "standup/ol/model.ckpt-2000000"), ("poses", "ik", "poses/model.ckpt-2000000"),
]
F_GOAL, F_BACKWARDS, F_ENV_GOAL = 1, 8, 32      # include/rexsim.h REX_F_*


@torch.no_grad()
def episode(env, act, max_steps):
    obs = env.reset()
    n = obs.shape[0]
    dev = obs.device
    ret = torch.zeros(n, device=dev)
    length = torch.zeros(n, device=dev)
    alive = torch.ones(n, dtype=torch.bool, device=dev)
    end_state = env.state.clone()
    for _ in range(max_steps):
        obs, reward, done, _ = env.step(act(obs))
        ret += torch.where(alive, reward.float(), torch.zeros_like(ret))
        length += alive.float()
        just = alive & done.bool()
        end_state[:, just] = env.state[:, just]
        alive &= ~done.bool()
        if not bool(alive.any()):
            break
    end_state[:, alive] = env.state[:, alive]
    return ret, length, ~alive, end_state


def summarise(env, ret, length, ended, st):
    flags = st[43].view(torch.int32)
    x, y, z = st[0], st[1], st[2]
    target = st[40]
    out = dict(mean_return=float(ret.mean()), mean_length=float(length.mean()), ended_by_itself=float(ended.float().mean()),
               reached_env_goal=float(((flags & F_ENV_GOAL) != 0).float().mean()),
               goal_flag=float(((flags & F_GOAL) != 0).float().mean()),
               fell_or_out=float((ended & ((flags & F_ENV_GOAL) == 0)).float().mean()),
               mean_abs_x=float(x.abs().mean()), mean_abs_y=float(y.abs().mean()), mean_z=float(z.mean()))
    if env.task in ("walk", "gallop"):
        back = (flags & F_BACKWARDS) != 0
        travelled = torch.where(back, x, -x)                                  # forward is -x (rex_gym_env.py:505)
        for name, sel in (("forward", ~back), ("backward", back)):
            if bool(sel.any()):
                out[name] = dict(episodes=int(sel.sum()), mean_travelled_m=float(travelled[sel].mean()),
                                 mean_target_m=float(target[sel].abs().mean()), mean_length=float(length[sel].mean()),
                                 fell=float((ended & sel & ((flags & F_ENV_GOAL) == 0)).float().sum() / sel.sum()))
    return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--policies-root", required=True)
    p.add_argument("--num-envs", type=int, default=256)
    p.add_argument("--max-steps", type=int, default=2500)
    p.add_argument("--gait-clock-scale", type=float, default=1.0,
                   help="wall-clock seconds per simulated second seen by GaitPlanner.loop (gait_planner.py:108-110)")
    a = p.parse_args()
    for task, signal, rel in POLICIES:
        row = dict(env=task, signal=signal, checkpoint=rel, num_envs=a.num_envs, max_steps=a.max_steps,
                   gait_clock_scale=a.gait_clock_scale)
        env = RexBatchEnv(a.num_envs, task=task, signal_type=signal, seed=1, gait_clock_scale=a.gait_clock_scale, check_actions=False)
        try:
            pol = SimplePPOPolicy(env, os.path.join(a.policies_root, rel))
        except (CheckpointError, FileNotFoundError) as e:
            row["missing"] = str(e).split(":")[0]
            print(json.dumps(row), flush=True)
            env.close()
            continue
        row["policy"] = summarise(env, *episode(env, pol.get_action, a.max_steps))
        lo = torch.as_tensor(env.action_space.low, device=env.device).minimum(torch.as_tensor(env.action_space.high, device=env.device))
        hi = torch.as_tensor(env.action_space.low, device=env.device).maximum(torch.as_tensor(env.action_space.high, device=env.device))
        gen = torch.Generator(device=env.device).manual_seed(0)
        rand = lambda obs: lo + (hi - lo) * torch.rand((a.num_envs, lo.numel()), device=env.device, generator=gen)   # noqa: E731
        row["random_actions"] = summarise(env, *episode(env, rand, a.max_steps))
        zero = lambda obs: torch.zeros((a.num_envs, lo.numel()), device=env.device)                                   # noqa: E731
        row["zero_actions"] = summarise(env, *episode(env, zero, a.max_steps))
        print(json.dumps(row), flush=True)
        env.close()


if __name__ == "__main__":
    main()

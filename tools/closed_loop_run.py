#!/usr/bin/env python3
"""A closed-loop rollout for profiling runs: the fused actor (rex_step_segment_policy) on the benchmark workload -- 30 untimed 50-step
launches (the pre-roll), then `--launches` launches of `--segment` steps each; prints the wall-clock and device-timestamp figures of those
launches as one JSON line, to be set next to the rocprofv3 kernel trace of the same process (tools/measure.sh prof_policy:
dispatches 30 .. 30 + launches of the `rex_step_kernel<..., true, true>` instantiation).
  python tools/closed_loop_run.py [--envs 4096] [--segment 25] [--launches 40] [--task walk --signal ik --mark base]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096); ap.add_argument("--segment", type=int, default=25); ap.add_argument("--launches", type=int, default=40)
    ap.add_argument("--task", default="walk"); ap.add_argument("--signal", default="ik"); ap.add_argument("--mark", default="base")
    a = ap.parse_args()
    import torch
    from rex_gym_amd import RexBatchEnv
    from rex_gym_amd.agents.fused_actor import FusedActor
    from rex_gym_amd.agents.ppo import ForwardGaussianPolicy, PPOConfig, StreamingNormalize
    n, T = a.envs, a.segment
    env = RexBatchEnv(n, task=a.task, signal_type=a.signal, mark=a.mark, seed=0, auto_reset=True, max_episode_steps=2000, check_actions=False, range_normalize=True)
    torch.manual_seed(0)
    net = ForwardGaussianPolicy(env.obs_dim, env.action_dim, PPOConfig()).cuda()
    flt = StreamingNormalize((env.obs_dim,), clip=5, device="cuda")
    obs = env.reset()
    flt.update(obs)
    FusedActor(env, net, flt, sample=True, seed=1)
    g = torch.Generator(device="cuda"); g.manual_seed(77)
    for k in range(30):
        o, r, d, info = env.step_segment_policy(50, obs)
        obs = o[-1].clone()
        if k < 26:
            idx = torch.randperm(n, device="cuda", generator=g)[: max(1, n // 32)].to(torch.int32)
            obs[idx.long()] = env.reset(idx)
    torch.cuda.synchronize()
    env.set_timing(3)
    t0 = time.perf_counter()
    for k in range(a.launches):
        o, r, d, info = env.step_segment_policy(T, obs)
        obs = o[-1].clone()
    torch.cuda.synchronize()
    e = time.perf_counter() - t0
    ms = env.step_times_ms(a.launches)
    print(json.dumps({"workload": f"{n} envs {a.task}-{a.signal}/{a.mark}, closed loop (fused 4-200-100-A actor), {T}-step segments", "launches": a.launches,
                      "steps_per_launch": T, "kernel_ms_per_launch_device_timestamps": sum(ms) / len(ms), "kernel_ms_per_step": sum(ms) / len(ms) / T,
                      "wall_ms_per_step": e / (a.launches * T) * 1e3, "env_steps_per_s": n * a.launches * T / e,
                      "envs_per_wave": env._L.rex_envs_per_wave(env._h), "finite": bool(torch.isfinite(o).all())}))


if __name__ == "__main__":
    main()

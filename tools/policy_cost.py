#!/usr/bin/env python3
"""What the actor inside the launch costs: the closed-loop kernels (rex_step_policy / rex_step_segment_policy) against the open-loop
ones (rex_step / rex_step_segment) ON THE SAME TRAJECTORY -- env B replays the actions env A's fused actor took (the twin test of
tests/test_gpu_policy.py shows the two are bit-identical), so both run the same physics and differ by the actor alone.  Kernel
durations are device timestamps (rex_set_timing(3)).

  python tools/policy_cost.py [--envs 4096] [--steps 400] [--task walk --signal ik] [--hidden 200,100]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096); ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--preroll", type=int, default=1500)
    ap.add_argument("--task", default="walk"); ap.add_argument("--signal", default="ik"); ap.add_argument("--mark", default="base")
    ap.add_argument("--hidden", default="200,100"); ap.add_argument("--segment", type=int, default=25)
    a = ap.parse_args()
    import torch
    from rex_gym_amd import RexBatchEnv
    from rex_gym_amd.agents.fused_actor import FusedActor
    from rex_gym_amd.agents.ppo import ForwardGaussianPolicy, PPOConfig, StreamingNormalize
    n, T = a.envs, a.segment
    kw = dict(task=a.task, signal_type=a.signal, mark=a.mark, seed=0, auto_reset=True, max_episode_steps=2000, check_actions=False, range_normalize=True)
    A_, B_ = RexBatchEnv(n, **kw), RexBatchEnv(n, **kw)
    torch.manual_seed(0)
    net = ForwardGaussianPolicy(A_.obs_dim, A_.action_dim, PPOConfig(policy_layers=tuple(int(v) for v in a.hidden.split(",")))).cuda()
    flt = StreamingNormalize((A_.obs_dim,), clip=5, device="cuda")
    obs = A_.reset(); B_.reset()
    flt.update(obs)
    FusedActor(A_, net, flt, sample=True, seed=1)
    # pre-roll (closed loop on A, replayed on B)
    for k in range(a.preroll // 50):
        o, r, d, info = A_.step_segment_policy(50, obs)
        B_.step_segment(info["policy_action"])
        obs = o[-1].clone()
    assert torch.equal(A_.state, B_.state)
    out = {"envs": n, "task": f"{a.task}-{a.signal}/{a.mark}", "hidden": a.hidden, "envs_per_wave": A_._L.rex_envs_per_wave(A_._h)}

    def kernel_ms(env):
        ms = env.step_times_ms(4096)
        env.set_timing(False)
        return sum(ms) / max(len(ms), 1), len(ms)

    # one launch per step
    acts = []
    A_.set_timing(3)
    for k in range(a.steps):
        o, r, d, info = A_.step_policy(obs)
        acts.append(info["policy_action"]); obs = o
    ms_a, cnt = kernel_ms(A_)
    B_.set_timing(3)
    for k in range(a.steps):
        B_.step(acts[k])
    ms_b, _ = kernel_ms(B_)
    assert torch.equal(A_.state, B_.state)
    # the SEG kernel with one step per launch (what the register cost of the segment loop alone is)
    obs_keep = obs.clone()
    acts = []
    for k in range(a.steps):
        o, r, d, info = A_.step_policy(obs)
        acts.append(info["policy_action"]); obs = o
    B_.set_timing(3)
    for k in range(a.steps):
        B_.step_segment(acts[k][None])
    ms_b1, _ = kernel_ms(B_)
    assert torch.equal(A_.state, B_.state)
    out["per_step"] = {"fused_kernel_ms": ms_a, "open_loop_kernel_ms": ms_b, "open_loop_seg_kernel_T1_ms": ms_b1, "launches": cnt,
                       "actor_ms": ms_a - ms_b1}
    # one launch per segment
    for TT in (T, 4 * T):
        segs = max(8, a.steps // TT)
        recs = []
        A_.set_timing(3)
        for k in range(segs):
            o, r, d, info = A_.step_segment_policy(TT, obs)
            recs.append(info["policy_action"]); obs = o[-1].clone()
        ms_a, cnt = kernel_ms(A_)
        B_.set_timing(3)
        for k in range(segs):
            B_.step_segment(recs[k])
        ms_b, _ = kernel_ms(B_)
        assert torch.equal(A_.state, B_.state)
        out[f"segment_{TT}"] = {"fused_kernel_ms_per_step": ms_a / TT, "open_loop_kernel_ms_per_step": ms_b / TT, "launches": cnt,
                                "actor_ms_per_step": (ms_a - ms_b) / TT}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

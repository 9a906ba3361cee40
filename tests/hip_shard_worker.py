"""Worker of tests/test_gpu_parity.py::test_hip_shards_on_two_ranks_reproduce_the_single_process_batch: one rank of a 2-rank job
(torch.distributed.run, gloo; both ranks share the one GPU of the test box).  Every rank steps ITS shard of a global batch on the HIP
path -- RexBatchEnv(n / 2, env_index_base = rank * n / 2) -- through the steps of one rollout segment (in-launch resets included),
all-gathers the segment with sharding.gather_rollout, and rank 0 compares the gathered blocks with a single-process
RexBatchEnv(n) run of the same seeds and actions: bit for bit.  usage: hip_shard_worker.py CASE OUT.json"""
import json
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

CASES = {
    "walk_ik": dict(n=2048, kw=dict(task="walk", signal_type="ik")),
    "mixed_arm": dict(n=2048, kw=dict(task="mixed", signal_type="ik", mark="arm", mass_scale_range=(0.8, 1.2), friction_range=(0.25, 0.625))),
    "walk_ik_policy": dict(n=2048, kw=dict(task="walk", signal_type="ik", range_normalize=True), policy=True),
}


def rollout(case, count, base, T=50):
    from rex_gym_amd import RexBatchEnv
    c = CASES[case]
    n = c["n"]
    env = RexBatchEnv(count, seed=23, env_index_base=base, auto_reset=True, max_episode_steps=17, check_actions=False, **c["kw"])
    obs0 = env.reset()
    if c.get("policy"):       # closed loop: the actor inside the launch, samples keyed by the GLOBAL env index
        from rex_gym_amd.agents.fused_actor import FusedActor
        from rex_gym_amd.agents.ppo import ForwardGaussianPolicy, PPOConfig
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(3)
            net = ForwardGaussianPolicy(env.obs_dim, env.action_dim, PPOConfig()).to(env.device)
            with torch.no_grad():
                net.mean.weight.mul_(10.0)
        FusedActor(env, net, None, sample=True, seed=5)
        o, r, d, info = env.step_segment_policy(T, obs0)
        seg = {"obs": o, "reward": r, "done": d.to(torch.uint8), "action": info["policy_action"]}
    else:                     # one global action table, every rank takes its rows
        g = torch.Generator().manual_seed(11)
        lo, hi = float(min(env.action_space.low.min(), env.action_space.high.min())), float(max(env.action_space.low.max(), env.action_space.high.max()))
        acts = (torch.rand((T, n, env.action_dim), generator=g) * (hi - lo) + lo)[:, base:base + count].to(env.device).contiguous()
        obs = torch.zeros((T, count, env.obs_dim), device=env.device); rew = torch.zeros((T, count), device=env.device)
        done = torch.zeros((T, count), dtype=torch.uint8, device=env.device)
        for t in range(T):
            env.step(acts[t], out=(obs[t], rew[t], done[t]))
        seg = {"obs": obs, "reward": rew, "done": done, "action": acts}
    torch.cuda.synchronize()
    state = env.state.clone()
    env.close()
    return obs0, seg, state


def main():
    case, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from rex_gym_amd.sharding import gather_rollout, shard_from_env
    n = CASES[case]["n"]
    sh = shard_from_env(n)
    assert sh.num_envs == n // world and sh.env_index_base == rank * (n // world)
    obs0, seg, state = rollout(case, sh.num_envs, sh.env_index_base)
    full = gather_rollout(seg)                                        # [T, n, ...] in global env order on every rank
    states = [torch.zeros_like(state).cpu() for _ in range(world)]
    dist.all_gather(states, state.cpu())
    obs0s = [torch.zeros_like(obs0).cpu() for _ in range(world)]
    dist.all_gather(obs0s, obs0.cpu())
    if rank == 0:
        w_obs0, w_seg, w_state = rollout(case, n, 0)
        res = {"case": case, "ranks": world, "resets_in_segment": int(w_seg["done"].sum()),
               "obs0": bool(torch.equal(torch.cat(obs0s, 0), w_obs0.cpu())), "state": bool(torch.equal(torch.cat(states, 1), w_state.cpu()))}
        for k in ("obs", "reward", "done", "action"):
            res[k] = bool(torch.equal(full[k].cpu(), w_seg[k].cpu()))
        with open(out_path, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""N > 1 path on CPU: two gloo ranks, each simulating its shard of one global batch (here on the
oracle, since there is no GPU), then the learner hand-off all-gather.  Checks that (a) shard r of a
2-rank run equals slice r of a 1-rank run (global-index RNG keys), (b) gather_rollout returns the
segment in global env order on every rank."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _rollout(n, base, steps, seed, mixed=False):
    import orclib
    if mixed:   # BASELINE configs[4] shape: task drawn per env from its GLOBAL index, mass / friction drawn per reset
        cfg = orclib.default_config("mixed", "ik", n, seed=seed, env_index_base=base, auto_reset=1, max_episode_steps=6, task_mix=0b111,
                                    action_repeat=6, solver_iterations=60, mass_scale_lo=0.8, mass_scale_hi=1.2, friction_lo=0.25,
                                    friction_hi=0.625)
    else:
        cfg = orclib.default_config("walk", "ik", n, seed=seed, env_index_base=base, auto_reset=1, max_episode_steps=6)
    env = orclib.OracleEnv(cfg, np.float32)
    obs0 = env.reset()
    rng = np.random.RandomState(42)
    acts = rng.uniform(-0.4, 0.4, (steps, 8, 2)).astype(np.float32)     # global action table for 8 envs
    if mixed:
        acts *= 0.025                                                    # inside every task's Box
    seg = {"obs": [], "reward": [], "done": [], "action": []}
    for t in range(steps):
        a = acts[t, base:base + n]
        o, r, d, _ = env.step(a)
        seg["obs"].append(o.copy()); seg["reward"].append(r.copy()); seg["done"].append(d.copy()); seg["action"].append(a)
    return obs0, {k: np.stack(v) for k, v in seg.items()}, env.get_state()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rex_gym_amd.sharding import gather_rollout, shard_from_env
    sh = shard_from_env(8)
    assert sh.num_envs == 4 and sh.env_index_base == 4 * rank
    _, seg, _ = _rollout(sh.num_envs, sh.env_index_base, 8, seed=7)
    full = gather_rollout({k: torch.from_numpy(v.astype(np.float32)) for k, v in seg.items()})
    # a one-step segment: the gathered [T = 1, world * n, ...] tensor would be a VIEW of the cached receive buffer -- the caller
    # must still own what it got when the next gather on the same slot arrives
    from rex_gym_amd.sharding import clear_gather_buffers
    a1 = gather_rollout({"x": torch.full((1, 4, 2), float(rank))})
    keep = a1["x"].clone()
    a2 = gather_rollout({"x": torch.full((1, 4, 2), float(rank) + 10.0)})
    assert torch.equal(a1["x"], keep) and torch.equal(a2["x"], keep + 10.0), "a gathered segment was overwritten by the next gather"
    assert torch.equal(keep[0, :4], torch.zeros(4, 2)) and torch.equal(keep[0, 4:], torch.ones(4, 2))      # global env order
    clear_gather_buffers()
    q.put((rank, {k: v.numpy() for k, v in full.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_match_single_rank_and_gather():
    os.environ["OMP_NUM_THREADS"] = "1"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, HERE)
    _, ref, _ = _rollout(8, 0, 8, seed=7)
    for rank in (0, 1):
        for k in ("obs", "reward", "done", "action"):
            got = results[rank][k]
            assert got.shape[:2] == (8, 8)
            np.testing.assert_array_equal(got, ref[k].astype(np.float32), err_msg=f"rank {rank} {k}")
    assert ref["done"][5].all()   # episode limit 6 hit on every env, auto-reset drew new targets


def test_mixed_task_shards_match_the_single_rank_batch():
    """The mixed-task batch shards like any other: an env's task, its per-reset mass / friction draws and its episode
    draws are keyed by the GLOBAL env index, so the two halves of an 8-env batch stepped as separate shards (what two ranks
    do) reproduce the single batch bit for bit."""
    sys.path.insert(0, HERE)
    full_obs0, full, full_state = _rollout(8, 0, 8, seed=11, mixed=True)
    for rank in (0, 1):
        obs0, seg, state = _rollout(4, 4 * rank, 8, seed=11, mixed=True)
        np.testing.assert_array_equal(obs0, full_obs0[4 * rank:4 * rank + 4])
        for k in ("obs", "reward", "done"):
            np.testing.assert_array_equal(seg[k], full[k][:, 4 * rank:4 * rank + 4], err_msg=k)
        np.testing.assert_array_equal(state, full_state[:, 4 * rank:4 * rank + 4])


def _ppo_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rex_gym_amd.agents import PPOAgent, PPOConfig, train
    from test_agents_ppo import _PointEnv
    n = 16
    cfg = PPOConfig(policy_layers=(16,), value_layers=(16,), update_every=n, update_epochs_policy=3, update_epochs_value=3,
                    policy_lr=1e-3, value_lr=1e-3, max_length=8)
    agent = PPOAgent(n, 1, 1, cfg, device="cpu", seed=2, sync_gradients=True)   # same initial weights on every rank
    # different data on every rank, and episodes that end at different steps on different ranks: the update decision
    # itself has to be collective (PPOAgent.train_if_all_full), or the gradient all-reduces of the ranks mis-pair
    train(_PointEnv(n, seed=100 + rank, fall=0.15), agent, 8 * 6, sync_poll=4)
    flat = torch.cat([p.detach().reshape(-1) for p in agent.net.parameters()])
    q.put((rank, torch.cat([flat, agent.observ_filter.mean.reshape(-1), torch.tensor([agent.penalty])]).numpy(), agent.updates))
    dist.barrier()
    dist.destroy_process_group()


def test_ppo_ranks_stay_in_step_when_gradients_are_averaged():
    """The learner's multi-rank mode: every rank trains on its own shard's episodes, gradients are all-reduced, so the
    weights, the KL penalty and the observation filter stay identical on all ranks, although the ranks' episodes end at
    different times."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29810 + (os.getpid() % 150)
    procs = [ctx.Process(target=_ppo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (w, u) for r, w, u in (q.get(timeout=240) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] >= 2
    np.testing.assert_array_equal(res[0][0], res[1][0])


def _gather_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rex_gym_amd.sharding import gather_rollout, shard_from_env
    sh = shard_from_env(2 * world)                                   # two envs per rank
    g = torch.arange(sh.env_index_base, sh.env_index_base + sh.num_envs, dtype=torch.float32)
    t = torch.arange(3, dtype=torch.float32)
    seg = {"obs": (100 * t[:, None, None] + g[None, :, None] + 0.25 * torch.arange(4)[None, None, :]).contiguous(),     # [T, n, 4]: value = f(step, GLOBAL env, word)
           "done": ((t[:, None] + g[None, :]) % 3 == 0)}                                                                 # bool travels as bytes
    a = gather_rollout(seg, async_op=True, slot=0)                    # two segments in flight, as the bench keeps them
    b = gather_rollout({k: (v + 1000 if v.dtype != torch.bool else ~v) for k, v in seg.items()}, async_op=True, slot=1)
    a, b = a.wait(), b.wait()
    q.put((rank, a["obs"].numpy(), a["done"].numpy(), b["obs"].numpy(), b["done"].numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_rollout_on_four_ranks_keeps_the_global_env_order():
    """The hand-off at the world sizes a SCALE run uses beyond two: four gloo ranks, two segments in flight on two buffer slots; every rank
    receives [T, 4 * n, ...] in global env order, bool blocks included."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 150)
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = np.arange(2 * world, dtype=np.float32)
    t = np.arange(3, dtype=np.float32)
    obs = 100 * t[:, None, None] + g[None, :, None] + 0.25 * np.arange(4, dtype=np.float32)[None, None, :]
    done = (t[:, None] + g[None, :]) % 3 == 0
    for rank, a_obs, a_done, b_obs, b_done in res:
        np.testing.assert_array_equal(a_obs, obs, err_msg=f"rank {rank}")
        np.testing.assert_array_equal(a_done, done)
        np.testing.assert_array_equal(b_obs, obs + 1000)
        np.testing.assert_array_equal(b_done, ~done)

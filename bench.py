#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched Rex walk-IK hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps 2000 --warmup 200
  python bench.py --gpus N --steps K --warmup W            (N > 1: starts its own N ranks, one per GPU, under torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE launch of rex_step_kernel = one env.step() of every env of the shard
(controller + 5 x [motor model + restated stepSimulation with 60 PGS iterations] + reward/done/obs,
with in-launch auto-reset), writing its observation / reward / done into slice t of a rollout segment.
Workload = BASELINE.json configs[1]: 4 096 Rex envs per GPU, walk-IK, flat plane, random actions
U(-0.4, 0.4) drawn afresh for every step -- one device-side draw (one kernel) per 25-step rollout
segment on the launch stream, inside the timed region, a segment ahead of its use -- (weak scaling: 4 096 envs on every GPU, independent shards, no
data-path collective).  With N > 1 every rank all-gathers
its finished 25-step segment to all ranks (RCCL over xGMI: the learner hand-off, the design's only
collective) while the next segment is stepped; the line reports the throughput with it (`value`) and
without it.  `--config 3|4|5` runs BASELINE.json's other configs at their TOTAL sizes divided over the
GPUs (strong scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
PREROLL_STEPS = 1500             # untimed: episodes fall / time out at different steps, so after this many steps the batch
                                 # holds a stationary mix of episode phases whatever --warmup the caller asks for


def algorithmic_bytes(task, signal, terrain, mark, action_dim, obs_dim):
    """ALGORITHMIC bytes per env-step = persistent state words read + written (64 walk / poses / standup,
    62 gallop: no gait phase words, 66 turn: start and target yaw; + 15 for the 6 arm motors of mark 'arm'), the action
    row in, the observation row + reward + done out, and on a heightfield 4 feet x 4 height samples.  Gives the
    survey's 541 (walk-IK), 581 (gallop-OL), 621 (turn-IK on the heightfield pool), 661 (arm walk-IK)."""
    words = {"walk": 64, "gallop": 62, "turn": 66, "poses": 64, "standup": 64, "mixed": 66}[task] + (15 if mark == "arm" else 0)
    return 2 * 4 * words + 4 * action_dim + 4 * obs_dim + 4 + 1 + (64 if terrain != "plane" else 0)


def layout_bytes(state_words, action_dim, obs_dim):
    """this implementation's own traffic: its state words read + written, action, obs, reward, done"""
    return 2 * 4 * state_words + 4 * action_dim + 4 * obs_dim + 4 + 1
ALGO_FLOP_PER_ENV_STEP = 3.0e5   # SURVEY.md 8(d) estimate
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3


def csrc_sha16():
    """A hash of the sources the library is built from (rex_gym_amd/csrc/*, include/rexsim.h, the compile flags): printed next to
    `roofline.traffic`, whose rocprofv3 counter passes cannot be collected from inside this process -- a traffic.json entry measured on
    other sources is visible as a different hash."""
    import hashlib
    from rex_gym_amd import build as hb
    h = hashlib.sha256(" ".join(hb.HIPCC_FLAGS).encode())
    for f in sorted(os.listdir(hb.CSRC)):
        if f.endswith((".h", ".hip")):
            with open(os.path.join(hb.CSRC, f), "rb") as fh:
                h.update(f.encode()); h.update(fh.read())
    with open(os.path.join(ROOT, "include", "rexsim.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


def usable_cores():
    """CPUs this process may actually run on: the affinity mask, capped by the cgroup CPU quota (the GPU boxes show 256
    logical CPUs but grant 16 CPUs of time; 256 OpenMP threads on that quota run 5x slower than 16-32)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:     # cgroup v1
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                cores = min(cores, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return cores


def cpu_baseline(task="walk", signal="ik", mark="base", n=ENVS_PER_GPU, cores=None, max_seconds=20.0):
    """Time the CPU oracle (a restatement, NOT PyBullet) on the host cores: bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orclib
    cores = cores or usable_cores()
    cfg = orclib.default_config(task, signal, n, seed=0, auto_reset=1, max_episode_steps=2000, mark=orclib_mark(mark))
    env = orclib.OracleEnv(cfg, np.float32, mark=mark)
    threads = int(env.o.lib.orc_set_threads(cores))
    env.reset()
    rng = np.random.RandomState(0)
    b = {"walk": {"ik": 0.4, "ol": 0.01}, "gallop": {"ik": 0.4, "ol": 0.3}, "turn": {"ik": 0.01, "ol": 0.01}}.get(task, {}).get(signal, 0.1)
    acts = rng.uniform(-b, b, (8, n, env.action_dim)).astype(np.float32)
    for k in range(3):
        env.step(acts[k])
    t0 = time.perf_counter()
    steps = 0
    while True:
        env.step(acts[steps % 8])
        steps += 1
        dt = time.perf_counter() - t0
        if dt > max_seconds or steps >= 2000:
            break
    env.close()
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{n} envs x {steps} steps {task}-{signal.upper()} ({mark} mark), oracle/rex_oracle.c fp32 build, OpenMP "
                      f"over envs on {threads} thread(s) of the {usable_cores()} CPUs this container is granted "
                      f"({dt:.1f} s); CPU restatement, not PyBullet (pybullet is not installable here).  SURVEY 8(d) baseline (iii), the "
                      "reference's own pure-Python controller (GaitPlanner.loop + Kinematics.solve + 5 x MotorModel.convert_to_torque), "
                      "needs /root/reference and cannot run on the GPU box: 990 control steps/s on one core of the build container "
                      "(profiles/r06_reference_python_controller.json, tools/time_reference_controller.py; round 1: 1 134)"}


def orclib_mark(mark):
    return 1 if mark == "arm" else 0


def joint_rmse_vs_oracle(workload, kwargs, n=256, steps=200, seed=11):
    """Half of BASELINE.json's metric ("joint RMSE vs PyBullet"): PyBullet cannot run here, so this is the HIP path
    against the fp64 oracle -- `n` envs of the bench workload from reset, the same random actions, `steps` control steps
    (1 s of robot time); per env the RMSE over time and joints of (q_hip - q_oracle), median / p99 / max over the envs
    (tests/parity_window.py; the full-size windows of every BASELINE config are in profiles/r06_parity.json).
    The error figures are those of the PRODUCT kernels (the instantiations timed above: no event trace set); a second pass with
    the debug `_trace` instantiations splits the envs by their discrete events and must reproduce the first bit for bit
    (`trace_pass_bit_identical`).  `fp32_tolerance` is the stated tolerance of north_star: next to the whole-batch p99 the
    FLOAT32 FLOOR of the same window (the oracle's own fp32 build against its fp64 build), the share of envs that took every
    discrete decision as the fp64 oracle did, the p99 on them, and the p99 of every env up to its first such divergence."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import parity_window as pw
    from rex_gym_amd import RexBatchEnv
    env = RexBatchEnv(n, device=torch.cuda.current_device(), seed=seed, **kwargs)
    name = f"bench:{workload}"
    rec = pw.window(name, env, steps=steps, seed=seed, threads=usable_cores())
    floor = pw.float32_floor(name, env, steps=steps, seed=seed, threads=usable_cores())
    env.close()
    rec.pop("abs_error_by_step", None)
    ev = rec["events"]
    rec["fp32_tolerance"] = {
        "bar_rad": 1e-3, "p99_rad": rec["p99_rad"], "float32_floor_p99_rad": floor["p99_rad"],
        "stated": "whole-batch p99 <= max(1e-3 rad, 1.2 x the float32 floor of the same window); envs that take the fp64 oracle's "
                  "discrete decisions (contacts, facets, bounds, controller flags) <= 1e-3 rad at p99",
        "within_stated_tolerance": bool(rec["p99_rad"] <= max(1e-3, 1.2 * floor["p99_rad"])),
        # the ABSOLUTE bar of north_star (1e-3 rad), never relaxed by the floor: whole-batch p99 and median against it
        "within_1e-3_rad_absolute": bool(rec["p99_rad"] <= 1e-3), "median_within_1e-3_rad_absolute": bool(rec["median_rad"] <= 1e-3),
        "share_envs_with_the_oracles_event_sequence": ev["share_same_event_sequence"],
        "p99_rad_on_them": ev["joint_rmse_same_events"].get("p99_rad"),
        "p99_rad_until_first_divergence": ev["joint_rmse_until_first_divergence"].get("p99_rad"),
        "float32_floor": floor}
    return rec


def vs_pybullet_record(device):
    """The other half of BASELINE.json's metric against PyBullet itself, as far as the reference holds PyBullet data: its shipped turn / ol and
    standup / ol checkpoints carry recorded episodes of their envs on real PyBullet (tests/golden/make_pybullet_golden.py; observations, not joint
    angles: base roll / pitch and their rates).  The recorded actions replayed on the HIP path (tests/pybullet_replay.py: checker code, after the
    timed region, like the oracle legs)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    import numpy as np
    import pybullet_replay as pr
    eps = pr.load()
    s = pr.summarize(eps, pr.HipReplayer(eps, steps=120, device=device), steps=120, windows=(25, 50, 100))
    st = pr.load_standup()
    shipped = pr.replay_standup_hip(st, device=device)
    low = pr.replay_standup_hip(st, device=device, friction_range=(0.25, 0.25))
    ev = s["event_peaks"]
    return {
        "what": "recorded PyBullet episodes of the reference's RexTurnEnv / RexStandupEnv (its shipped checkpoints' episode memory) replayed on the HIP "
                "path: RMS error of base roll and pitch [rad] over the first 25 / 50 / 100 control steps, gait events on the record's control step, "
                "standup outcome",
        "turn_episodes": s["episodes"], "turn_roll_pitch_rmse_rad": {str(k): v["rp_rmse"] for k, v in s["windows"].items()},
        "turn_roll_pitch_record_rms_rad": {str(k): v["rp_ref_rms"] for k, v in s["windows"].items()},
        "turn_rate_profile_correlation": s["rate_profile_correlation"],
        "turn_gait_events_on_the_records_step": f"{sum(1 for e in ev.values() if e['record'] == e['replay'])} of {len(ev)}",
        "within_1e-3_rad": bool(s["windows"][25]["rp_rmse"] <= 1e-3),
        "standup_episodes": len(st), "standup_record_return_mean": float(np.mean([e["reward"].sum() for e in st])),
        "standup_default_friction": {"fell": int(sum(r[3] is not None for r in shipped)), "return_mean": float(np.mean([r[2].sum() for r in shipped]))},
        "standup_friction_0.25": {"fell": int(sum(r[3] is not None for r in low)), "return_mean": float(np.mean([r[2].sum() for r in low]))},
    }


class Rollout:
    """An env shard stepped the way the timed region steps it: every step writes observation / reward / done straight into slice t
    of a rollout segment [T, n, ...] (RexBatchEnv.step(out=...), no copies); every T steps the finished segment is all-gathered to
    all ranks (the learner hand-off, sharding.gather_rollout) while the next segment fills the other buffer -- the collective runs
    on RCCL's stream, the steps keep the compute stream busy --; the actions of every step are drawn afresh on the device, one
    draw (one kernel) per segment on the launch stream, a segment ahead of its use."""

    def __init__(self, env, n, T, dev, gen):
        import torch
        self.env, self.n, self.T, self.dev, self.gen = env, n, T, dev, gen
        lo = self.lo = torch.as_tensor(env.action_space.low, device=dev).minimum(torch.as_tensor(env.action_space.high, device=dev)).float()
        hi = self.hi = torch.as_tensor(env.action_space.low, device=dev).maximum(torch.as_tensor(env.action_space.high, device=dev)).float()
        self.span = (hi - lo).contiguous()
        self.one_box = bool((lo == lo[0]).all() and (hi == hi[0]).all())   # every dimension of the env's Box has the same bounds (all five envs)
        self.lo0, self.hi0 = float(lo[0]), float(hi[0])
        # (the action pool is drawn inside the env's Box and lives in HBM before the timed region: BatchEnv's per-step Box test -- a
        #  host synchronisation per step -- has nothing to find and is switched off by the caller: check_actions=False)
        self.pool = self.draw_actions(torch.empty((T, n, env.action_dim), device=dev))
        assert bool(((self.pool >= lo) & (self.pool <= hi)).all())     # by construction; checked once on a drawn segment
        self.seg = [dict(obs=torch.zeros((T, n, env.obs_dim), device=dev), reward=torch.zeros((T, n), device=dev),
                         done=torch.zeros((T, n), dtype=torch.uint8, device=dev), action=self.draw_actions(torch.empty_like(self.pool)))
                    for _ in range(2)]
        self.seg_bytes = sum(v.numel() * v.element_size() for v in self.seg[0].values())
        # (slices and their device pointers are taken once, outside the timed loops: nothing but the launch is left per step)
        self.acts = [self.pool[t] for t in range(T)]
        self.sacts = [[s["action"][t] for t in range(T)] for s in self.seg]
        self.outs = [[env.bind_out(s["obs"][t], s["reward"][t], s["done"][t]) for t in range(T)] for s in self.seg]
        self.clock = 0     # steps taken through run() so far: the segments continue across the warm-up / timed / untimed calls

    def draw_actions(self, buf):
        """a rollout segment of fresh actions, uniform over the env's Box, drawn on the launch stream: one kernel"""
        import torch
        if self.one_box:
            return buf.uniform_(self.lo0, self.hi0, generator=self.gen)
        torch.rand(buf.shape, device=self.dev, generator=self.gen, out=buf)
        return buf.mul_(self.span).add_(self.lo)

    def preroll(self, stagger=True, steps=PREROLL_STEPS):
        """untimed, outside --warmup: right after a synchronous reset every env is in the same episode phase and the solver
        converges quickly; the number reported is the steady state a training run sees"""
        import torch
        env, n, T = self.env, self.n, self.T
        env.reset()
        for k in range(steps):
            env.step(self.pool[k % T])
            if k % 50 == 0 and k < steps - 200 and stagger:
                # stagger the episodes: half the envs (those that walk backwards) never fall, and started together they would
                # all run into the 2 000-step cap in the same step, stand up together, fall together ... -- a load that swings
                # by 30 % with a 2 000-step period.  Resetting a random 1 / 32 of the batch every 50 pre-roll steps spreads the
                # episode ages, as a training run that has been going for a while has them.
                idx = torch.randperm(n, device=self.dev, generator=self.gen)[: max(1, n // 32)].to(torch.int32)
                env.reset(idx)

    def run(self, steps, gather, segment_launch=False):
        """segment_launch: the steps of a segment in ONE launch (RexBatchEnv.step_segment / rex_step_segment: the actions of a segment
        are drawn a segment ahead anyway) instead of one launch per step; the same steps, the same results, bit for bit."""
        from rex_gym_amd.sharding import gather_rollout
        T, seg, sacts, outs = self.T, self.seg, self.sacts, self.outs
        pending = [None, None]
        step = self.env.step
        self.launch_steps = []       # env steps of every launch of this call, in order (a segment launch at the edges of the call is shorter than T)
        k, end = self.clock, self.clock + steps
        while k < end:
            b, t = (k // T) & 1, k % T
            if t == 0 and pending[b] is not None:
                pending[b].wait(); pending[b] = None         # this buffer's previous segment has left before it is overwritten
            if segment_launch:
                m = min(T - t, end - k)
                sb = seg[b]
                self.env.step_segment(sb["action"][t:t + m], out=(sb["obs"][t:t + m], sb["reward"][t:t + m], sb["done"][t:t + m]))
                self.launch_steps.append(m)
                k += m; t += m - 1
            else:
                step(sacts[b][t], outs[b][t])
                k += 1
            if t == T - 1:
                if gather:
                    pending[b] = gather_rollout(seg[b], async_op=True, slot=b)
                # fresh actions for every step: the OTHER buffer's next segment is drawn now, behind this
                # segment's last launch (its previous segment was handed over one segment ago)
                if pending[b ^ 1] is not None:
                    pending[b ^ 1].wait(); pending[b ^ 1] = None
                self.draw_actions(seg[b ^ 1]["action"])
        self.clock += steps
        for p in pending:
            if p is not None:
                p.wait()


def north_star_workload(args, dev, local_rank, rank, world, dist, barrier, T):
    """BASELINE.json north_star: "65 536 parallel Rex walk-IK envs at 8 x MI355X" = 8 192 per GPU.  The `value` of the line stays the
    weak-scaling one of configs[1] (4 096 per GPU, so that N = 1 agrees with the single-GPU record); a multi-GPU run reports this
    block next to it: walk-IK, 65 536 / 8 x n_gpus envs in total, the same pre-roll and segment hand-off, with and without the
    all-gather.  Returns the dict rank 0 prints (every rank takes part)."""
    import torch
    from rex_gym_amd import RexBatchEnv
    per_gpu = 65536 // 8
    env = RexBatchEnv(per_gpu, device=local_rank, seed=0, env_index_base=rank * per_gpu, auto_reset=True, max_episode_steps=2000,
                      check_actions=False, task="walk", signal_type="ik")
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321 + rank)
    ro = Rollout(env, per_gpu, T, dev, gen)
    ro.preroll()
    res = {}
    for name, gather in (("with_gather", True), ("without_gather", False)):
        ro.run(args.warmup, gather)
        barrier()
        t0 = time.perf_counter()
        ro.run(args.steps, gather)
        torch.cuda.synchronize(dev)
        e = time.perf_counter() - t0
        barrier()
        t = torch.tensor([e], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res[name] = float(t.item())
    finite = bool(torch.isfinite(ro.seg[0]["obs"]).all().item())
    env.close()
    total = per_gpu * world
    return {"workload": f"walk-IK, flat plane, base mark, {per_gpu} envs per GPU = 65 536 / 8 x {world} GPUs = {total} envs in total (north_star: "
                        f"65 536 at 8 GPUs), same pre-roll, {T}-step segments, fresh actions every step",
            "envs_total": total, "envs_per_gpu": per_gpu, "scaling": "weak (8 192 per GPU)",
            "value": total * args.steps / res["with_gather"], "unit": "env-steps/s", "ms_per_step": res["with_gather"] / args.steps * 1e3,
            "value_without_gather": total * args.steps / res["without_gather"],
            "ms_per_step_without_gather": res["without_gather"] / args.steps * 1e3, "steps": args.steps, "finite": finite}


def closed_loop_workload(n, dev, local_rank, rank, env_kw, steps, T=25, world=1, dist=None, backend="nccl"):
    """The rollout the reference's PPO agents actually run: a POLICY IN THE LOOP -- `action = algo.perform(prevob)` then
    `batch_env.simulate(action)` for every step (agents/tools/simulate.py:57-76, agents/ppo/algorithm.py:105-134) -- on the workload of the
    timed region (same envs, auto-reset, episode cap 2000), behind the wrapper stack the agents act through (RangeNormalize + ClipAction
    folded into the launch).  Policy: the reference's shape (agents/scripts/networks.py:66-110, configs.py:29-34: observ filter -> 200 relu ->
    100 relu -> tanh mean, free logstd -1, Gaussian sample), random initial weights.  Four ways to run it, same envs, same policy:
      torch_policy_per_step   perform() as PyTorch ops (filter, three matmuls, sample) + one rex_step launch per step
      fused_per_step          the actor inside the step launch (rex_step_policy): one launch per step
      fused_segment_T / 4T    rex_step_segment_policy: one launch per T-step (4T-step) rollout segment
    Every variant writes obs / action / mean / reward / done of every step into rollout-segment blocks, as the learner's memory needs them."""
    import torch
    from rex_gym_amd import RexBatchEnv
    from rex_gym_amd.agents.fused_actor import FusedActor
    from rex_gym_amd.agents.ppo import ForwardGaussianPolicy, PPOConfig, StreamingNormalize
    env = RexBatchEnv(n, device=local_rank, seed=0, env_index_base=rank * n, auto_reset=True, max_episode_steps=2000, check_actions=False,
                      range_normalize=True, **env_kw)
    O, A = env.obs_dim, env.action_dim
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(0)
        net = ForwardGaussianPolicy(O, A, PPOConfig()).to(dev)
    flt = StreamingNormalize((O,), center=True, scale=True, clip=5, device=dev)
    obs0 = env.reset()
    flt.update(obs0)
    actor = FusedActor(env, net, flt, sample=True, seed=1)
    gen = torch.Generator(device=dev); gen.manual_seed(77 + rank)
    T4 = 4 * T
    obs = torch.zeros((T4 + 1, n, O), device=dev); act = torch.zeros((T4, n, A), device=dev); mean = torch.zeros((T4, n, A), device=dev)
    rew = torch.zeros((T4, n), device=dev); done = torch.zeros((T4, n), dtype=torch.uint8, device=dev)
    obs[0].copy_(obs0)

    def segment(length):
        env.step_segment_policy(length, obs[0], out=(obs[1:length + 1], rew[:length], done[:length]), action=act[:length], mean=mean[:length])
        obs[0].copy_(obs[length])

    # untimed pre-roll, closed loop: 1 500 steps with staggered resets (as the open-loop pre-roll), the filter following the rollout
    for k in range(PREROLL_STEPS // 50):
        segment(50)
        if k % 4 == 0:
            flt.update(obs[1:51].reshape(-1, O)); actor.sync()
        if k < PREROLL_STEPS // 50 - 4:
            idx = torch.randperm(n, device=dev, generator=gen)[: max(1, n // 32)].to(torch.int32)
            obs[0][idx.long()] = env.reset(idx)
    res = {}

    def timed(name, fn, count, launches):
        fn(max(2, count // 8))
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        fn(count)
        torch.cuda.synchronize(dev)
        e = time.perf_counter() - t0
        if dist is not None:       # every rank rolls out its own shard (no exchange on this path): the job's rate is set by the slowest rank
            t = torch.tensor([e], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t.item())
        res[name] = {"value": n * world * count / e, "ms_per_step": e / count * 1e3, "steps": count, "launches_per_step": launches}

    # (a) perform() in PyTorch + rex_step: what a learner written against env.step() gets (no host synchronisation in the loop)
    sgen = torch.Generator(device=dev); sgen.manual_seed(5)
    outs = [env.bind_out(obs[t + 1], rew[t], done[t]) for t in range(T)]

    def torch_loop(count):
        with torch.no_grad():
            for k in range(count):
                t = k % T
                if t == 0 and k:
                    obs[0].copy_(obs[T])
                m_ = torch.tanh(net.mean(net.policy(flt.transform(obs[t]))))     # (the actor only: the value network is not evaluated during a rollout)
                mean[t].copy_(m_)
                torch.addcmul(m_, torch.exp(net.logstd), torch.randn(m_.shape, device=dev, generator=sgen), out=act[t])
                env.step(act[t], outs[t])
        obs[0].copy_(obs[(count - 1) % T + 1])
    timed("torch_policy_per_step", torch_loop, steps, None)   # (launches per step: the rocprofv3 kernel trace of this command, profiles/)

    # (b) the actor inside the launch, one launch per step
    def fused_loop(count):
        for k in range(count):
            t = k % T
            if t == 0 and k:
                obs[0].copy_(obs[T])
            env.step_policy(obs[t], out=(obs[t + 1], rew[t], done[t]), action=act[t], mean=mean[t])
        obs[0].copy_(obs[(count - 1) % T + 1])
    timed("fused_per_step", fused_loop, steps, 1.0)

    # (c) one launch per rollout segment
    def seg_loop(length):
        def run(count):
            for _ in range(count // length):
                segment(length)
        return run
    timed(f"fused_segment_{T}", seg_loop(T), max(steps, 8 * T) // T * T, 1.0 / T)
    timed(f"fused_segment_{T4}", seg_loop(T4), max(steps, 8 * T4) // T4 * T4, 1.0 / T4)
    finite = bool(torch.isfinite(obs).all().item())
    falls = float(done[:T].float().mean().item())
    env.close()
    return {"what": "closed-loop rollout (policy in the loop: the reference's 4-200-100-A Gaussian MLP actor behind its observ filter, random "
                    "initial weights, Gaussian sample every step; RangeNormalize + ClipAction folded) on the workload of the timed region",
            "unit": "env-steps/s", "envs_total": n * world, "finite": finite, "done_rate_per_step": falls, **res}


# BASELINE.json configs, numbered as SURVEY.md 8(d) numbers them (config 1 is the 1-env CPU plumbing case): what --config N runs.
# total = envs of the WHOLE job (strong scaling: divided over the GPUs); None = 4 096 per GPU (weak scaling, configs[1])
CONFIGS = {
    2: dict(task="walk", signal="ik", terrain="plane", mark="base", mixed=False, total=None),
    3: dict(task="gallop", signal="ol", terrain="plane", mark="base", mixed=False, total=65536),
    4: dict(task="turn", signal="ik", terrain="random", mark="base", mixed=False, total=32768),
    5: dict(task="walk", signal="ik", terrain="plane", mark="arm", mixed=True, total=16384),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS),
                    help="a BASELINE.json config by its SURVEY 8(d) number: 2 = 4 096 walk-IK envs per GPU (the default line, weak "
                         "scaling); 3 / 4 / 5 = 65 536 gallop-OL / 32 768 turn-IK on the heightfield pool / 16 384 mark-arm mixed-task "
                         "envs in TOTAL, divided over --gpus (strong scaling)")
    ap.add_argument("--gather-every", type=int, default=25, help="T: steps per rollout segment; every T steps the segment (obs, action, "
                    "reward, done of the shard) is all-gathered to all ranks -- the learner hand-off, the design's one collective")
    ap.add_argument("--no-gather", action="store_true", help="time the independent shards only (no learner hand-off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--segment-launch", action="store_true", help="the timed region launches once per rollout segment (rex_step_segment) instead of once per step")
    ap.add_argument("--no-segment-launch", action="store_true", help="skip the secondary segment-launch measurement")
    ap.add_argument("--no-walking-workload", action="store_true", help="skip the secondary gait-clock-1.5 measurement (profiling runs)")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip the secondary closed-loop (policy in the loop) measurement")
    ap.add_argument("--no-stagger", action="store_true", help="developer A/B runs only: pre-roll without the staggered resets")
    ap.add_argument("--no-device-timing", action="store_true", help="developer A/B runs only: no device-side kernel timestamps in the timed region")
    # the default line is BASELINE.json configs[1]; the other supported workloads can be timed with these
    ap.add_argument("--task", default="walk", choices=["walk", "gallop", "turn", "poses", "standup"])
    ap.add_argument("--signal", default="ik", choices=["ik", "ol"])
    ap.add_argument("--terrain", default="plane", choices=["plane", "random"])
    ap.add_argument("--mark", default="base", choices=["base", "arm"])
    ap.add_argument("--mixed", action="store_true", help="BASELINE configs[4] shape: env task split over {walk, gallop, turn}-IK "
                                                         "+ per-env mass x U(0.8,1.2) / foot friction 0.5 x U(0.5,1.25)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to smoke-test the "
                                                      "multi-rank path on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1) -- the same
        # command line under torch.distributed.run; rank 0 prints the one JSON line
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         "(or without it: bench.py starts its own ranks)")
    scaling = "weak"
    if args.config is not None:
        c = CONFIGS[args.config]
        args.task, args.signal, args.terrain, args.mark, args.mixed = c["task"], c["signal"], c["terrain"], c["mark"], c["mixed"]
        if c["total"] is not None and args.envs_per_gpu is None:
            if c["total"] % world:
                raise SystemExit(f"config {args.config}: {c['total']} envs do not divide over {world} GPUs")
            args.envs_per_gpu, scaling = c["total"] // world, "strong"
    n = args.envs_per_gpu or ENVS_PER_GPU
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ngpu = torch.cuda.device_count()
        if args.backend == "nccl" and ngpu < world:
            raise SystemExit(f"{world} ranks need {world} GPUs with the nccl backend (found {ngpu})")
        local_rank = local_rank % max(ngpu, 1)
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from rex_gym_amd import RexBatchEnv
    from rex_gym_amd.sharding import gather_rollout
    # (the action pool below is drawn inside the env's Box and lives in HBM before the timed region: BatchEnv's per-step Box
    #  test -- a host synchronisation per step -- has nothing to find and is switched off)
    env_kw = dict(task="mixed" if args.mixed else args.task, signal_type=args.signal, terrain_type=args.terrain, mark=args.mark)
    if args.mixed:
        env_kw.update(mass_scale_range=(0.8, 1.2), friction_range=(0.25, 0.625))
    env = RexBatchEnv(n, device=local_rank, seed=0, env_index_base=rank * n, auto_reset=True, max_episode_steps=2000, check_actions=False,
                      **env_kw)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    T = max(1, args.gather_every)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    ro = Rollout(env, n, T, dev, gen)
    ro.preroll(stagger=not args.mixed and not args.no_stagger)
    pool, acts, seg, seg_bytes, run, lo, hi = ro.pool, ro.acts, ro.seg, ro.seg_bytes, ro.run, ro.lo, ro.hi
    do_gather = not args.no_gather

    run(args.warmup, do_gather, args.segment_launch)
    if not args.no_device_timing:
        env.set_timing(3)      # device-side (first wave start, last wave end) ticks of the next launches: no event, no sync in the loop
    barrier()
    t0 = time.perf_counter()
    run(args.steps, do_gather, args.segment_launch)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    barrier()
    kms_raw = env.step_times_ms(min(args.steps, 4096)) if not args.no_device_timing else []
    timed_launch_steps = list(ro.launch_steps)           # (--segment-launch: the env steps behind each of those durations)
    kms = sorted(kms_raw)
    env.set_timing(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same K steps without the hand-off, and the hand-off on its own (blocking, nothing to overlap with): N > 1 only
    gather_info = None
    if dist is not None and do_gather:
        barrier()
        t1 = time.perf_counter()
        run(args.steps, False)
        torch.cuda.synchronize(dev)
        e_plain = time.perf_counter() - t1
        barrier()
        reps = 10
        t2 = time.perf_counter()
        for _ in range(reps):
            gather_rollout(seg[0], slot=0)
        torch.cuda.synchronize(dev)
        g_ms = (time.perf_counter() - t2) / reps * 1e3
        barrier()
        tt = torch.tensor([e_plain, g_ms], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        e_plain, g_ms = float(tt[0].item()), float(tt[1].item())
        gather_info = {"every_steps": T, "segment_bytes_per_rank": seg_bytes, "segment_bytes_all_ranks": seg_bytes * world,
                       "bytes_per_env_step": seg_bytes / (T * n), "gather_ms_blocking": g_ms,
                       "value_without_gather": n * world * args.steps / e_plain, "ms_per_step_without_gather": e_plain / args.steps * 1e3,
                       "backend": args.backend, "overlap": "segment k is gathered on the collective stream while segment k+1 is stepped"}

    # north_star's own multi-GPU workload next to the weak-scaling line (default workload, N > 1 only)
    north_star = None
    if dist is not None and do_gather and args.config is None and args.envs_per_gpu is None and not args.mixed \
            and (args.task, args.signal, args.terrain, args.mark) == ("walk", "ik", "plane", "base"):
        north_star = north_star_workload(args, dev, local_rank, rank, world, dist, barrier, T)

    # The same K steps with the steps of a segment in ONE launch (rex_step_segment): the timed region above launches once per step, the
    # Gym surface a policy in the loop needs; a rollout whose actions are known a segment ahead -- this benchmark's random-action
    # rollouts draw them a segment ahead -- does not have to.  Same steps, same results bit for bit; what changes is that a wave goes on
    # to its envs' next step without waiting for the slowest wave of every step.
    segment_launch = None
    if not args.no_segment_launch and not args.segment_launch:
        # (its own step counts: at least 8 launches of each segment length, however short the timed region above was)
        steps_seg, steps_seg4 = max(args.steps, 8 * T), max(args.steps, 32 * T)
        run(2 * T - ro.clock % T, do_gather, True)          # (up to a segment boundary: the timed launches below are whole segments)
        barrier()
        t3 = time.perf_counter()
        run(steps_seg, do_gather, True)
        torch.cuda.synchronize(dev)
        e_seg = time.perf_counter() - t3
        barrier()
        if dist is not None:
            tt = torch.tensor([e_seg], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e_seg = float(tt.item())
        # ... and with segments four times as long (their own buffers; the same env goes on): the longer the segment, the closer a
        # wave's sum of step times is to the mean
        ro4 = Rollout(env, n, 4 * T, dev, gen)
        ro4.run(8 * T, do_gather, True)
        barrier()
        t4 = time.perf_counter()
        ro4.run(steps_seg4, do_gather, True)
        torch.cuda.synchronize(dev)
        e_seg4 = time.perf_counter() - t4
        barrier()
        if dist is not None:
            tt = torch.tensor([e_seg4], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e_seg4 = float(tt.item())
        del ro4
        ms_step = elapsed / args.steps * 1e3
        segment_launch = {"steps_per_launch": T, "steps": steps_seg, "value": n * world * steps_seg / e_seg, "unit": "env-steps/s", "ms_per_step": e_seg / steps_seg * 1e3,
                          "vs_one_launch_per_step": ms_step / (e_seg / steps_seg * 1e3),
                          "longer_segments": {"steps_per_launch": 4 * T, "steps": steps_seg4, "value": n * world * steps_seg4 / e_seg4,
                                              "ms_per_step": e_seg4 / steps_seg4 * 1e3, "vs_one_launch_per_step": ms_step / (e_seg4 / steps_seg4 * 1e3)},
                          "what": "the workload of the timed region again (its own step count: `steps`), the T steps of a rollout segment in one launch (RexBatchEnv.step_segment -> rex_step_segment; "
                                  "actions drawn a segment ahead, as in the timed region); results bit-identical to the per-step launches "
                                  "(tests/test_gpu_parity.py::test_segment_launch_is_bit_identical_to_single_steps)"}

    # does this sim regroup its envs every step?  (two sorting launches behind each step kernel, outside kernel_ms: include/rexsim.h)
    probe = torch.empty(n, dtype=torch.int32, device=dev)
    regroups = env._L.rex_get_sweeps(env._h, probe.data_ptr(), env._stream_ptr()) == 0
    del probe

    # Launch duration of the dominant kernel over the TIMED launches: device-side timestamps (rex_set_timing(3)): first-wave
    # start to last-wave end of each launch -- the kernel alone, comparable with rocprofv3's kernel trace (profiles/).
    # launch_ms: two HIP events around m back-to-back launches / m on the launch stream (RexBatchEnv launches on torch's
    # current stream, so torch.cuda.Event records on that very stream) -- the kernel plus the dispatch hand-over.
    m = min(256, max(20, args.steps // 8))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(torch.cuda.current_stream(dev))
    for k in range(m):
        env.step(acts[k % T])      # (launch-rate probe: one pre-drawn segment of actions, recycled)
    ev1.record(torch.cuda.current_stream(dev))
    ev1.synchronize()
    launch_ms = ev0.elapsed_time(ev1) / m
    if not kms:
        env.set_timing(3)
        for k in range(m):
            env.step(pool[k % T])
        kms = sorted(env.step_times_ms(m))
        env.set_timing(False)
    kernel_ms = sum(kms) / len(kms)
    if args.segment_launch and not args.no_device_timing:
        # the timed launches covered the K steps, up to T steps each: per STEP, as everywhere else in the line -- every duration divided by
        # the step count of ITS launch (the first and last launch of the timed call are shorter when K is not a multiple of T)
        ls = timed_launch_steps[-len(kms_raw):]
        kernel_ms = sum(kms_raw) / max(sum(ls), 1)
        kms = sorted(d / m for d, m in zip(kms_raw, ls))

    # HBM bytes per launch and the VALU issue fraction as measured with rocprofv3 PMC passes of this same command (cannot be
    # collected from inside the process); null when no measurement of this workload is committed
    traffic = issue_frac = traffic_commit = traffic_csrc = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f).get(f"{'mixed' if args.mixed else args.task}-{args.signal}/{args.terrain}/{args.mark}/{n}")
        traffic = t and t["bytes_per_launch"]
        issue_frac = t and t.get("issue_frac")
        traffic_commit, traffic_csrc = t and t.get("library_commit"), t and t.get("csrc_sha16")
    except (OSError, ValueError, KeyError):
        pass

    # sanity: the rollout must be alive (finite observations, some episodes running)
    finite = bool(torch.isfinite(seg[0]["obs"]).all().item() and torch.isfinite(seg[1]["obs"]).all().item())
    act_dim, obs_dim, state_words = env.action_dim, env.obs_dim, env.state_words
    body_contacts, repeat, sweeps = env.config.body_contacts, env.config.action_repeat, env.config.solver_iterations
    env.close()

    # the same workload with the gait phase clock of a host that runs 1.5 wall-seconds per simulated second (the regime in
    # which the reference's gait -- driven by time.time(), gait_planner.py:108-110 -- walks instead of falling after ~240
    # steps): reported next to the headline, never instead of it
    walking = None
    if world == 1 and not args.mixed and not args.no_walking_workload and args.signal == "ik" and args.task in ("walk", "turn", "gallop"):
        env2 = RexBatchEnv(n, device=local_rank, seed=0, env_index_base=rank * n, auto_reset=True, max_episode_steps=2000,
                           gait_clock_scale=1.5, check_actions=False, **env_kw)
        env2.reset()
        for k in range(PREROLL_STEPS + args.warmup):
            env2.step(pool[k % T])
            if k % 50 == 0 and k < PREROLL_STEPS - 200:
                env2.reset(torch.randperm(n, device=dev, generator=gen)[: max(1, n // 32)].to(torch.int32))
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for k in range(args.steps):
            env2.step(pool[k % T])
        torch.cuda.synchronize(dev)
        e2 = time.perf_counter() - t1
        walking = {"gait_clock_scale": 1.5, "value": n * args.steps / e2, "unit": "env-steps/s", "ms_per_step": e2 / args.steps * 1e3}
        env2.close()

    # the rollout a PPO learner runs: the policy in the loop (single-task workloads; the toes-only kernels carry the fused actor)
    closed_loop = None
    if not args.mixed and not args.no_closed_loop and not body_contacts:
        if dist is None:
            try:
                closed_loop = closed_loop_workload(n, dev, local_rank, rank, env_kw, max(args.steps, 200), T)
            except Exception as e:   # reporting only; never fail the headline for it
                closed_loop = {"failed": f"{type(e).__name__}: {e}"}
        else:                        # (multi-rank: the block's barriers and reductions are collective -- an exception on one rank must surface)
            closed_loop = closed_loop_workload(n, dev, local_rank, rank, env_kw, max(args.steps, 200), T, world, dist, args.backend)

    task_name = "mixed" if args.mixed else args.task
    ranks_seen = dist.get_world_size() if dist is not None else 1
    try:
        rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())       # torch's "nccl" on ROCm is RCCL
    except Exception:
        rccl_version = None
    if rank == 0:
        total_envs = n * world
        value = total_envs * args.steps / elapsed
        algo_bytes = algorithmic_bytes(task_name, args.signal, args.terrain, args.mark, act_dim, obs_dim)
        achieved_gbs = algo_bytes * n / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec (all envs) Rex " + ("mixed walk/gallop/turn-IK" if args.mixed else f"{args.task}-{args.signal.upper()}"),
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n} Rex envs per GPU, " + ("every env's task drawn from walk/gallop/turn-IK, mass and friction drawn per reset, one launch "
                                   "per step, " if args.mixed else f"{args.task}-{args.signal.upper()}, ") + f""
                                   f"{'flat plane' if args.terrain == 'plane' else 'random heightfield pool'}, {args.mark} mark, "
                                   f"{'link-box rows (ground + self collision) on, ' if body_contacts else ''}"
                                   f"dt 1 ms x {repeat} substeps, <= {sweeps} PGS sweeps "
                                   "(Bullet residual threshold 1e-7), auto-reset, episode cap 2000, uniform random actions "
                                   f"over the env's action Box drawn afresh for every step (one device-side draw per {T}-step segment on the "
                                   f"launch stream, inside the timed region); {PREROLL_STEPS} "
                                   "untimed pre-roll steps with staggered resets before --warmup (stationary episode-age mix); every step "
                                   f"writes obs / reward / done into a {T}-step rollout segment" +
                                   (f", all-gathered to all {world} ranks every {T} steps (RCCL, overlapped with the next segment)" if gather_info else ""),
                       "baseline_config": args.config if args.config is not None else (2 if (task_name, args.signal, args.terrain, args.mark, n) == ("walk", "ik", "plane", "base", ENVS_PER_GPU) else None),
                       "envs_total": total_envs, "parallelism": f"env-shards x{world} (no data-path collective; learner hand-off = one all-gather per rollout segment)"},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         # which library the counter passes behind `traffic` / `issue_frac` ran on, and which one is being timed now
                         "traffic_library_commit": traffic_commit, "traffic_csrc_sha16": traffic_csrc, "csrc_sha16": csrc_sha16(),
                         "traffic_is_of_this_library": (traffic_csrc == csrc_sha16()) if traffic_csrc else None,
                         "kernel": "rex_step_kernel", "kernel_ms": kernel_ms, "kernel_ms_min": kms[0],
                         "kernel_ms_over": f"{len(kms)} of the {args.steps} timed launches" if not args.no_device_timing else f"{len(kms)} launches after the timed region",
                         "launch_ms": launch_ms,
                         "regroup_launches_ms": (launch_ms - kernel_ms) if regroups else None,   # the per-step sort of a regrouped batch: in launch_ms and ms_per_step, not in kernel_ms
                         "algorithmic_bytes_per_env_step": algo_bytes,
                         "layout_bytes_per_env_step": layout_bytes(state_words, act_dim, obs_dim),
                         "note": "the fused step is bound by the VALU issue rate of one wave per SIMD through the sequential "
                                 "contact solver (about 550 flop/B, SURVEY.md 8d; DESIGN.md 5-6), not by HBM: see issue_frac / valu_frac",
                         "issue_frac": issue_frac,
                         "valu_tflops_est": ALGO_FLOP_PER_ENV_STEP * n / (kernel_ms * 1e-3) / 1e12,
                         "valu_frac": ALGO_FLOP_PER_ENV_STEP * n / (kernel_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS},
            "finite": finite,
            # what the process group itself reports: proof that the collective library saw all ranks of a multi-GPU run
            "n_ranks_seen": ranks_seen, "backend": args.backend if world > 1 else None, "rccl_version": rccl_version,
        }
        if gather_info:
            out["rollout_gather"] = gather_info
        if segment_launch:
            out["segment_launch"] = segment_launch
        out["config"]["launches"] = (f"one launch per {T}-step rollout segment (rex_step_segment)" if args.segment_launch else
                                     "one launch per env.step() (rex_step)")
        if north_star:
            out["north_star_workload"] = north_star
        if walking:
            out["walking_gait_workload"] = walking
        if closed_loop:
            out["closed_loop"] = closed_loop
        if not args.no_cpu_baseline and world == 1:
            if not args.mixed:
                try:   # SURVEY.md 8(d) baseline (ii): the CPU restatement on all granted threads and on one
                    out["cpu_baseline"] = cpu_baseline(args.task, args.signal, args.mark, n)
                    out["cpu_baseline_1thread"] = cpu_baseline(args.task, args.signal, args.mark, min(n, 256), cores=1, max_seconds=8.0)
                except Exception as e:  # the baseline is reporting only; never fail the GPU line for it
                    out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port",
                                           "sample": f"failed: {e}"}
            try:   # the other half of BASELINE.json's metric
                out["joint_rmse_vs_oracle"] = joint_rmse_vs_oracle(f"{task_name}-{args.signal}/{args.terrain}/{args.mark}", env_kw)
            except Exception as e:
                out["joint_rmse_vs_oracle"] = {"failed": str(e)}
            if (task_name, args.signal, args.terrain, args.mark) == ("walk", "ik", "plane", "base"):     # the default line only
                try:
                    out["vs_pybullet_record"] = vs_pybullet_record(local_rank)
                except Exception as e:
                    out["vs_pybullet_record"] = {"failed": f"{type(e).__name__}: {e}"}
        # compact copies of the secondary measurements inside `config` / `roofline` (flat scalars: the part of the line a record keeper that
        # reduces unknown top-level keys to their names still carries); the full blocks stay at the top level
        cfgd, rl = out["config"], out["roofline"]
        if segment_launch:
            lg = segment_launch["longer_segments"]
            cfgd[f"open_loop_segment_{segment_launch['steps_per_launch']}_env_steps_per_s"] = segment_launch["value"]
            cfgd[f"open_loop_segment_{lg['steps_per_launch']}_env_steps_per_s"] = lg["value"]
        if walking:
            cfgd["walking_gait_clock_1.5_env_steps_per_s"] = walking["value"]
        if closed_loop:
            for k, v in closed_loop.items():
                if isinstance(v, dict) and "value" in v:
                    cfgd[f"closed_loop_{k}_env_steps_per_s"] = v["value"]
                    cfgd[f"closed_loop_{k}_ms_per_step"] = v["ms_per_step"]
            if "failed" in closed_loop:
                cfgd["closed_loop_failed"] = closed_loop["failed"][:100]
        jr = out.get("joint_rmse_vs_oracle") or {}
        tol = jr.get("fp32_tolerance") or {}
        if tol:
            rl["joint_rmse_p99_rad"] = tol.get("p99_rad")
            rl["joint_rmse_float32_floor_p99_rad"] = tol.get("float32_floor_p99_rad")
            rl["joint_rmse_share_envs_with_the_oracles_events"] = tol.get("share_envs_with_the_oracles_event_sequence")
            rl["joint_rmse_p99_rad_on_them"] = tol.get("p99_rad_on_them")
            rl["joint_rmse_within_1e-3_rad"] = tol.get("within_1e-3_rad_absolute")
        pb = out.get("vs_pybullet_record") or {}
        if "turn_roll_pitch_rmse_rad" in pb:
            rl["pybullet_record_turn_roll_pitch_rmse_25_steps_rad"] = pb["turn_roll_pitch_rmse_rad"]["25"]
            rl["pybullet_record_turn_gait_events_on_the_records_step"] = pb["turn_gait_events_on_the_records_step"]
            rl["pybullet_record_within_1e-3_rad"] = pb["within_1e-3_rad"]
            rl["pybullet_record_standup_fell_default_friction"] = f"{pb['standup_default_friction']['fell']} of {pb['standup_episodes']}"
            rl["pybullet_record_standup_return_friction_0.25_vs_record"] = f"{pb['standup_friction_0.25']['return_mean']:.0f} vs {pb['standup_record_return_mean']:.0f}"
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched Rex walk-IK hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps 2000 --warmup 200
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE launch of rex_step_kernel = one env.step() of every env of the shard
(controller + 5 x [motor model + restated stepSimulation with 60 PGS iterations] + reward/done/obs,
with in-launch auto-reset).  Workload = BASELINE.json configs[1]: 4 096 Rex envs per GPU, walk-IK,
flat plane, random actions U(-0.4, 0.4) from a pre-generated pool that is resident in HBM before
the timed region (weak scaling: 4 096 envs on every GPU, independent shards, no data-path collective).
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
# algorithmic bytes per env-step, walk-IK base mark (state r+w, action, obs, reward, done)
ALGO_BYTES_PER_ENV_STEP = 541
# this implementation's own layout: 54 state words read + written, action 8 B, obs 16 B, reward 4 B, done 1 B
LAYOUT_BYTES_PER_ENV_STEP = 2 * 54 * 4 + 8 + 16 + 4 + 1
ALGO_FLOP_PER_ENV_STEP = 3.0e5   # SURVEY.md 8(d) estimate
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3


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


def cpu_baseline(max_seconds=20.0):
    """Time the CPU oracle (a restatement, NOT PyBullet) on the host cores: bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orclib
    cores = usable_cores()
    n = ENVS_PER_GPU            # the GPU line's own batch
    cfg = orclib.default_config("walk", "ik", n, seed=0, auto_reset=1, max_episode_steps=2000)
    env = orclib.OracleEnv(cfg, np.float32)
    threads = int(env.o.lib.orc_set_threads(cores))
    env.reset()
    rng = np.random.RandomState(0)
    acts = rng.uniform(-0.4, 0.4, (8, n, 2)).astype(np.float32)
    env.step(acts[0])
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
            "sample": f"{n} envs x {steps} steps walk-IK, oracle/rex_oracle.c fp32 build, OpenMP over envs on "
                      f"{threads} threads = the CPUs this container is granted "
                      f"({dt:.1f} s); CPU restatement, not PyBullet (pybullet is not installable here)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # the default line is BASELINE.json configs[1]; the other supported configs can be timed with these
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
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
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

    from rex_gym_amd import RexBatchEnv, RexMixedBatchEnv
    n = args.envs_per_gpu
    if args.mixed:
        env = RexMixedBatchEnv(n, device=local_rank, seed=0, env_index_base=rank * n, auto_reset=True, max_episode_steps=2000,
                               terrain_type=args.terrain, mark=args.mark, mass_scale_range=(0.8, 1.2), friction_range=(0.25, 0.625))
        env.action_space, env.config = env.envs[2].action_space, env.envs[0].config   # +-0.01: inside every task's Box
    else:
        env = RexBatchEnv(n, task=args.task, signal_type=args.signal, device=local_rank, seed=0, env_index_base=rank * n,
                          auto_reset=True, max_episode_steps=2000, terrain_type=args.terrain, mark=args.mark)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    lo = torch.as_tensor(env.action_space.low, device=dev).minimum(torch.as_tensor(env.action_space.high, device=dev))
    hi = torch.as_tensor(env.action_space.low, device=dev).maximum(torch.as_tensor(env.action_space.high, device=dev))
    pool = [(torch.rand((n, env.action_dim), device=dev, generator=gen) * (hi - lo) + lo).contiguous() for _ in range(16)]
    env.reset()

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        env.step(pool[k % 16])
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        env.step(pool[k % 16])
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-launch duration of the dominant kernel: HIP events recorded around each launch on the launch stream
    timed = env.envs[0] if args.mixed else env    # mixed: the kernel of the first sub-batch
    timed.set_timing(True)
    m = min(200, max(20, args.steps // 10))
    kms = []
    for k in range(m):
        env.step(pool[k % 16])
        kms.append(timed.last_step_ms())
    timed.set_timing(False)
    kms.sort()
    kernel_ms = sum(kms) / len(kms)

    # HBM bytes per launch as measured with rocprofv3 PMC passes of this same command (cannot be collected from inside
    # the process); null when no measurement of this workload is committed
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f).get(f"{args.task}-{args.signal}/{args.terrain}/{args.mark}/{n}")
        traffic = t and t["bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

    # sanity: the rollout must be alive (finite observations, some episodes running)
    obs = env._obs
    finite = bool(torch.isfinite(obs).all().item())
    env.close()

    if rank == 0:
        total_envs = n * world
        value = total_envs * args.steps / elapsed
        achieved_gbs = ALGO_BYTES_PER_ENV_STEP * n / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec (all envs) Rex " + ("mixed walk/gallop/turn-IK" if args.mixed else f"{args.task}-{args.signal.upper()}"),
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n} Rex envs per GPU, " + ("walk/gallop/turn-IK in equal parts with per-env mass and friction "
                                   "draws, " if args.mixed else f"{args.task}-{args.signal.upper()}, ") + f""
                                   f"{'flat plane' if args.terrain == 'plane' else 'random heightfield pool'}, {args.mark} mark, "
                                   f"dt 1 ms x {env.config.action_repeat} substeps, <= {env.config.solver_iterations} PGS sweeps "
                                   "(Bullet residual threshold 1e-7), auto-reset, episode cap 2000, uniform random actions "
                                   "over the env's action Box",
                       "envs_total": total_envs, "parallelism": f"env-shards x{world} (no data-path collective)"},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "rex_step_kernel", "kernel_ms": kernel_ms, "kernel_ms_min": kms[0],
                         "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP,
                         "layout_bytes_per_env_step": LAYOUT_BYTES_PER_ENV_STEP,
                         "note": "the fused step is bound by the VALU issue rate of one wave per SIMD through the sequential "
                                 "contact solver (about 550 flop/B, SURVEY.md 8d; DESIGN.md 5-6), not by HBM: see valu_frac",
                         "valu_tflops_est": ALGO_FLOP_PER_ENV_STEP * n / (kernel_ms * 1e-3) / 1e12,
                         "valu_frac": ALGO_FLOP_PER_ENV_STEP * n / (kernel_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS},
            "finite": finite,
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline is reporting only; never fail the GPU line for it
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

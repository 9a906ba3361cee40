"""The joint-angle half of BASELINE.json's metric ("joint RMSE vs PyBullet", bar 1e-3 rad), measured the only way it can be
here: the HIP path against the fp64 oracle (oracle/rex_oracle.c -- a restatement; PyBullet is not installable).

One *window* = `steps` control steps from reset of the same envs under the same random actions, on the GPU and in the
oracle; per env the RMSE over time and joints of (q_hip - q_oracle) while both episodes run, reported as median / p99 /
max over the envs.  The workloads are BASELINE.json's configs[1..4] at their per-GPU shard sizes plus the walking-gait
workload (gait clock 1.5: the robot does not fall, so the whole window counts).

TEST INFRASTRUCTURE: imported by tests/test_gpu_parity.py, by bench.py's reported `joint_rmse_vs_oracle` and by
tools/parity_report.py (profiles/r06_parity.json); never by the product.
"""
import numpy as np

import orclib

WORKLOADS = {
    # name: (envs, RexBatchEnv keywords)
    "walk_ik_4096": (4096, dict(task="walk", signal_type="ik")),                                   # configs[1]
    "walk_ik_8192": (8192, dict(task="walk", signal_type="ik")),                                   # north_star's shard: 65 536 / 8
    "gallop_ol_8192": (8192, dict(task="gallop", signal_type="ol")),                               # configs[2] shard
    "turn_ik_heightfield_4096": (4096, dict(task="turn", signal_type="ik", terrain_type="random")),  # configs[3] shard
    "mixed_arm_2048": (2048, dict(task="mixed", signal_type="ik", mark="arm", mass_scale_range=(0.8, 1.2),
                                  friction_range=(0.25, 0.625))),                                  # configs[4] shard
    "walk_ik_gait_clock_1.5_4096": (4096, dict(task="walk", signal_type="ik", gait_clock_scale=1.5)),
}


def make_env(name, n=None, seed=23, **extra):
    from rex_gym_amd import RexBatchEnv
    n0, kw = WORKLOADS[name]
    return RexBatchEnv(n or n0, seed=seed, **kw, **extra)


def oracle_for(env, dtype=np.float64):
    """An oracle batch with the env's own RexConfig (byte copy of the struct the product was created with)."""
    cfg = orclib.RexConfig.from_buffer_copy(bytes(env.config))
    orc = orclib.OracleEnv(cfg, dtype, env.mark)
    if env.terrain_type == "random":
        orc.set_terrain(env.terrain_heights.cpu().numpy(), env.terrain_mids.cpu().numpy())
    return orc


def action_box(env):
    return np.minimum(env.action_space.low, env.action_space.high), np.maximum(env.action_space.low, env.action_space.high)


_TRAJ = {}


def oracle_trajectory(name, env, steps, seed, threads=None, keep_states=(40, 120)):
    """fp64 oracle rollout of the workload from reset (cached: the kernel variants of a workload share it):
    actions [steps, n, A] float32, q [steps, nm, n] float64, done [steps, n], full states at `keep_states`, and the event
    trace after every step, uint32 [steps, 3, n] (chained hashes of the discrete events / of the sweep counts / of the events
    without the arm's bounds), with its value BEFORE each of the `keep_states` steps (a single step replayed from such a state
    starts its chain there)."""
    key = (name, env.num_envs, steps, seed)
    if key not in _TRAJ:
        orc = oracle_for(env)
        if threads:
            orc.o.lib.orc_set_threads(int(threads))
        orc.reset()
        otrace = orc.set_event_trace(True)
        lo, hi = action_box(env)
        rng = np.random.RandomState(seed)
        acts = rng.uniform(lo, hi, (steps, env.num_envs, env.action_dim)).astype(np.float32)
        nm = env.num_motors
        q = np.zeros((steps, nm, env.num_envs))
        pos = np.zeros((steps, 3, env.num_envs))
        done = np.zeros((steps, env.num_envs), bool)
        trace = np.zeros((steps, 3, env.num_envs), np.uint32)
        states, pre_trace = {}, {}
        for k in range(steps):
            if k in keep_states:
                states[k] = orc.get_state()
                pre_trace[k] = otrace.copy()
            _, _, d, _ = orc.step(acts[k])
            st = orc.get_state()
            q[k], pos[k], done[k] = st[orclib.S_Q:orclib.S_Q + nm], st[0:3], d
            trace[k] = otrace
        orc.close()
        _TRAJ[key] = (acts, q, pos, done, states, (trace, pre_trace))
    return _TRAJ[key]


_FLOOR = {}


def float32_floor(name, env, steps=200, seed=23, threads=None):
    """The FLOAT32 FLOOR of the window: the oracle's own fp32 build against its fp64 build, same envs, same actions, same
    statistics as `window` (cached per workload).  What float32 arithmetic costs on this workload whatever the implementation:
    the bound the whole-batch figures of the HIP path are held against where BASELINE.json's 1e-3 rad is below the floor itself."""
    key = (name, env.num_envs, steps, seed)
    if key not in _FLOOR:
        acts, oq, _opos, odone, _, (otrace, _pre) = oracle_trajectory(name, env, steps, seed, threads)
        orc = oracle_for(env, np.float32)
        if threads:
            orc.o.lib.orc_set_threads(int(threads))
        orc.reset()
        t32 = orc.set_event_trace(True)
        n, nm = env.num_envs, env.num_motors
        sq = np.zeros(n); cnt = np.zeros(n); alive = np.ones(n, bool); same = np.ones(n, bool)
        sq_same = np.zeros(n); cnt_same = np.zeros(n)
        for k in range(steps):
            _, _, d, _ = orc.step(acts[k])
            e = orc.get_state()[orclib.S_Q:orclib.S_Q + nm] - oq[k]
            same &= (t32[0] == otrace[k, 0]) | ~alive
            sq += np.where(alive, (e * e).mean(0), 0.0); cnt += alive
            sq_same += np.where(alive & same, (e * e).mean(0), 0.0); cnt_same += alive & same
            alive &= ~(d.astype(bool) | odone[k])
        orc.close()
        rmse = np.sqrt(sq / np.maximum(cnt, 1))
        until = np.sqrt(sq_same[cnt_same > 0] / cnt_same[cnt_same > 0])
        _FLOOR[key] = dict(what="oracle fp32 build vs oracle fp64 build, same window and actions (CPU)",
                           median_rad=float(np.median(rmse)), p99_rad=float(np.percentile(rmse, 99)), max_rad=float(rmse.max()),
                           share_same_event_sequence=float(same.mean()),
                           p99_rad_same_events=float(np.percentile(rmse[same], 99)) if same.any() else None,
                           p99_rad_until_first_divergence=float(np.percentile(until, 99)) if until.size else None)
    return _FLOOR[key]


def window(name, env, steps=200, seed=23, threads=None, trace=True):
    """Run the window on `env` (freshly created with `seed`; its episode counters must be 0) -> record dict.

    Two passes over the same reset state and actions:
      1. the PRODUCT kernels (no event trace set: the instantiations bench.py times and users run) -- every error figure of the
         record (median / p99 / max, the matched-subset figures, the divergence curve) is computed from THIS pass's joint angles;
      2. (trace=True) the `_trace` instantiations (rex_set_event_trace: separate code objects, -DREX_TU_TRACE=1) for the event
         split alone -- which envs took every discrete decision as the oracle did -- and, after every step, the whole state block
         is compared bit for bit with pass 1 (`trace_pass_bit_identical`): the debug kernels are tied to the product kernels
         over the very window they annotate."""
    import torch
    acts, oq, opos, odone, _, (otrace, _pre) = oracle_trajectory(name, env, steps, seed, threads)
    n, nm = env.num_envs, env.num_motors
    # ---- pass 1: product kernels
    env.set_event_trace(False)
    env.reset()
    dev_acts = torch.as_tensor(acts, device=env.device)
    states, dones = [], []
    for k in range(steps):
        _, _, d, _ = env.step(dev_acts[k])
        states.append(env.state.clone()); dones.append(d.clone())
    pq = np.stack([s[orclib.S_Q:orclib.S_Q + nm].cpu().numpy().astype(np.float64) for s in states])      # [steps, nm, n] (float words)
    ppos = np.stack([s[0:3].cpu().numpy().astype(np.float64) for s in states])
    pdone = np.stack([d.cpu().numpy().astype(bool) for d in dones])
    # ---- pass 2: the trace instantiations, from the same reset state (episode counters back to 0: they key the Philox draws)
    ktr = None
    identical_steps = None
    if trace:
        env.state.zero_()
        env.reset()
        ktrace = env.set_event_trace(True)
        ktr = np.zeros((steps, 3, n), np.uint32)
        identical_steps = 0
        for k in range(steps):
            _, _, d, _ = env.step(dev_acts[k])
            identical_steps += int(torch.equal(env.state, states[k]) and torch.equal(d, dones[k]))
            ktr[k] = ktrace.cpu().numpy().view(np.uint32)
        env.set_event_trace(False)
    del states, dones
    same_events = np.ones(n, bool)      # the env's event sequence has been the oracle's so far (chained hash equal)
    same_sweeps = np.ones(n, bool)      # ... and so have its solver sweep counts
    same_leg_events = np.ones(n, bool)  # the event sequence without the arm's bounds (mark arm)
    first_div = np.full(n, -1)          # control step at which the event sequences parted (while both episodes ran)
    sq = np.zeros(n); cnt = np.zeros(n); alive = np.ones(n, bool)
    sq_same = np.zeros(n); cnt_same = np.zeros(n)     # the same sums over the steps up to which the env's events were the oracle's
    sq_legs = np.zeros(n)            # the 12 leg joints alone (mark 'arm': its arm joints sit ON their bounds and jitter)
    pos_err = np.zeros(n)
    curve = {}
    for k in range(steps):
        if trace:
            kt = ktr[k]
            ev_eq, sw_eq = kt[0] == otrace[k, 0], kt[1] == otrace[k, 1]
            same_leg_events &= (kt[2] == otrace[k, 2]) | ~alive
            first_div = np.where(alive & same_events & ~ev_eq, k, first_div)
            same_events &= ev_eq | ~alive; same_sweeps &= sw_eq | ~alive
        e = pq[k] - oq[k]
        sq += np.where(alive, (e * e).mean(0), 0.0); cnt += alive
        sq_same += np.where(alive & same_events, (e * e).mean(0), 0.0); cnt_same += alive & same_events
        sq_legs += np.where(alive, (e[:12] * e[:12]).mean(0), 0.0)
        pos_err = np.where(alive, np.abs(ppos[k] - opos[k]).max(0), pos_err)
        if k + 1 in (1, 5, 10, 25, 50, 100, 200) and alive.any():
            a = np.abs(e).max(0)[alive]
            curve[k + 1] = dict(median=float(np.median(a)), p99=float(np.percentile(a, 99)), max=float(a.max()), envs=int(alive.sum()))
        alive &= ~(pdone[k] | odone[k])      # an episode that ended (fall, goal) leaves the comparison
    rmse = np.sqrt(sq / np.maximum(cnt, 1))
    legs = np.sqrt(sq_legs / np.maximum(cnt, 1))

    def stats(x):
        return dict(envs=int(x.size), median_rad=float(np.median(x)), p99_rad=float(np.percentile(x, 99)), max_rad=float(x.max())) if x.size else dict(envs=0)
    hist_edges = [0, 1, 2, 5, 10, 25, 50, 100, 200, 10 ** 9]
    div = first_div[first_div >= 0]
    events = None if not trace else dict(
        what="per env and substep: toe points within the breaking distance, the heightfield facet under each of them (end centre "
             "and contact point), joint and arm bounds reached -- chained hash, HIP path vs fp64 oracle, compared after every "
             "control step while both episodes run (rex_set_event_trace / orc_set_event_trace)",
        share_same_event_sequence=float(same_events.mean()),
        share_same_events_and_sweep_counts=float((same_events & same_sweeps).mean()),
        share_same_events_without_arm_bounds=float(same_leg_events.mean()),
        leg_joints_rmse_same_events_without_arm_bounds=stats(legs[same_leg_events]),
        joint_rmse_same_events=stats(rmse[same_events]), joint_rmse_other_events=stats(rmse[~same_events]),
        # every env, over the control steps up to which its event sequence was still the oracle's (mark arm: no env keeps it for
        # the whole window -- its arm joints sit ON their bounds --, all of them keep it for a while)
        joint_rmse_until_first_divergence=dict(stats(np.sqrt(sq_same[cnt_same > 0] / cnt_same[cnt_same > 0])), mean_steps=float(cnt_same.mean())),
        leg_joints_rmse_same_events=stats(legs[same_events]),
        first_divergence_step_histogram={f"[{a}, {b})" if b < 10 ** 9 else f">= {a}": int(((div >= a) & (div < b)).sum())
                                         for a, b in zip(hist_edges[:-1], hist_edges[1:])})
    return dict(events=events, workload=name, envs=n, window_steps=steps, seed=seed,
                kernels="product kernels (rex_step_kernel<..., TRACE = false>, no event trace set): every error figure of this record; "
                        "the event split comes from a second pass with the _trace instantiations" if trace else
                        "product kernels (rex_step_kernel<..., TRACE = false>, no event trace set)",
                trace_pass_bit_identical=None if not trace else bool(identical_steps == steps),
                trace_pass_identical_steps=identical_steps,
                median_rad=float(np.median(rmse)),
                p99_rad=float(np.percentile(rmse, 99)), max_rad=float(rmse.max()),
                leg_joints_median_rad=float(np.median(legs)), leg_joints_p99_rad=float(np.percentile(legs, 99)),
                base_pos_err_p99_m=float(np.percentile(pos_err, 99)), base_pos_err_max_m=float(pos_err.max()),
                envs_compared_to_the_end=int(alive.sum()), mean_steps_compared=float(cnt.mean()),
                abs_error_by_step=curve, against="oracle/rex_oracle.c fp64 (restatement; PyBullet is not installable)")

"""Replay of the recorded PyBullet rollouts (tests/golden/pybullet_turn_ol_rollouts.npz, see tests/golden/make_pybullet_golden.py) on a
simulator of this repo -- TEST INFRASTRUCTURE.  The record holds, per episode, the RangeNormalize'd observation before every step, the
policy's action and the reward of the reference's RexTurnEnv (signal 'ol') on real PyBullet; what it does not hold is recovered here:

* substeps per control step: 6 (`RexConfig.action_repeat`; the record's leg switches -- `_open_loop_signal`, turn_env.py:271-311, every
  0.1 s of sim time -- sit 16.7 steps apart, and the first toe touch-down after the reset's teleport comes at step 6 = 36 ms; with the
  5 substeps of today's constructor default both come late);
* the turning direction (`_solve_direction`, :313-322) -- both are played, the one whose roll / pitch fit is kept;
* the start yaw (`reset`, :139-143: U(0.2, 6)) -- the world-frame rates (`GetBaseRollPitchYawRate`, rex.py:530-537) and |x| + |y| of the
  reward turn with it: the rotation about z that maps this run's rates onto the recorded ones (closed-form least squares over the first
  steps).  The dynamics themselves depend on it only through the friction pyramid, whose two directions are world axes (btPlaneSpace1 of
  the ground normal) in Bullet and here: `summarize(at_fitted_yaw=True)` replays every episode a second time from its fitted yaw
  (roll / pitch error 10-17 % smaller); the default, and what the tests and the bench line use, is one yaw per turning direction;
* the target yaw -- not needed before the goal is reached; the replay's target is placed 3 rad away in the turning direction.

Observation 0 of the record is exactly zero: the training-time env read the pose back after `resetBasePositionAndOrientation`
(yaw-only orientation, base velocity zeroed).  Today's reset returns the cached pre-teleport observation (`turn_env.py:159`,
reproduced by the env here, golden `turn_*` rollouts); the physical start state is the same in both, so step 1 onwards compares.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "pybullet_turn_ol_rollouts.npz")
OBS_HI = np.array([2 * np.pi + 0.01] * 2 + [2 * np.pi / 0.001 + 0.01] * 2)        # turn_env.py:404-417 + OBSERVATION_EPS
ACTION_REPEAT = 6
CCW, CW = (0.5, 3.5), (3.5, 0.5)                 # (init_orient, target_orient): _solve_direction() False / True


def load():
    d = np.load(FIXTURE)
    return [dict(observ=d["observ"][k, :n].astype(np.float64) * OBS_HI, action=d["action"][k, :n].astype(np.float64),
                 reward=d["reward"][k, :n].astype(np.float64), length=int(n)) for k, n in enumerate(d["length"])]


def replay_oracle(action, direction, steps, dtype=np.float64, probes=None, **cfg_kw):
    """One episode's actions on the CPU oracle.  Returns (obs [steps + 1, 4] in rad, rad/s; xy [steps + 1, 2]; done step or None)."""
    import ctypes
    import orclib
    kw = dict(num_envs=1, range_normalize=1, init_orient=direction[0], target_orient=direction[1], orient_fixed=3, max_episode_steps=1000,
              action_repeat=ACTION_REPEAT, solver_iterations=300 // ACTION_REPEAT)
    kw.update(cfg_kw)
    cfg = orclib.default_config("turn", "ol", **kw)
    env = orclib.OracleEnv(cfg, dtype)
    if probes:
        env.o.lib.orc_set_probe.argtypes = [ctypes.c_char_p, ctypes.c_double]
        for k, v in probes.items():
            assert env.o.lib.orc_set_probe(k.encode(), float(v)) == 0, k
        env.close()
        env = orclib.OracleEnv(cfg, dtype)        # (settled again under the probes)
    try:
        obs, xy, ended = [env.reset()[0] * OBS_HI], [env.get_state()[:2, 0].copy()], None
        for t in range(steps):
            o, _, d, _ = env.step(action[t][None])
            obs.append(o[0] * OBS_HI)
            xy.append(env.get_state()[:2, 0].copy())
            if d[0]:
                ended = t + 1
                break
        return np.asarray(obs, np.float64), np.asarray(xy, np.float64), ended
    finally:
        env.close()


def fit_yaw(ref_rates, rates):
    """Rotation about z (rad) that takes this run's world-frame (w_x, w_y) onto the recorded ones, least squares."""
    c = float((ref_rates * rates).sum())
    s = float((ref_rates[:, 1] * rates[:, 0] - ref_rates[:, 0] * rates[:, 1]).sum())
    return float(np.arctan2(s, c))


def rotate(v, a):
    c, s = np.cos(a), np.sin(a)
    return np.stack([c * v[:, 0] - s * v[:, 1], s * v[:, 0] + c * v[:, 1]], 1)


def compare(ep, obs, xy, init_yaw, windows=(25, 50, 100, 200), fit_steps=40):
    """Errors of one replay against the record, per window [1, K] of control steps."""
    n = min(len(obs), ep["length"])
    dyaw = fit_yaw(ep["observ"][1:min(n, fit_steps), 2:], obs[1:min(n, fit_steps), 2:])
    rates = rotate(obs[:n, 2:], dyaw)
    pos = rotate(xy[:n] - xy[0], dyaw)           # (the reset leaves the base at x = y = 0: rex.py init_position)
    reward = 0.035 - np.abs(pos[1:, 0]) - np.abs(pos[1:, 1])
    out = dict(start_yaw=float((init_yaw + dyaw) % (2 * np.pi)), steps=n)
    for k in windows:
        k = min(k, n)
        e = obs[1:k, :2] - ep["observ"][1:k, :2]
        out[k] = dict(rp_rmse=float(np.sqrt((e ** 2).mean())), rp_ref_rms=float(np.sqrt((ep["observ"][1:k, :2] ** 2).mean())),
                      rate_rmse=float(np.sqrt(((rates[1:k] - ep["observ"][1:k, 2:]) ** 2).mean())),
                      rate_ref_rms=float(np.sqrt((ep["observ"][1:k, 2:] ** 2).mean())),
                      reward_rmse=float(np.sqrt(((reward[:k - 1] - ep["reward"][:k - 1]) ** 2).mean())))
    return out


def best_direction(ep, replay, steps=60, **kw):
    """(direction, replay output) of the better roll / pitch fit over the first `steps` steps."""
    runs = {}
    for name, d in (("ccw", CCW), ("cw", CW)):
        obs, xy, ended = replay(ep["action"], d, min(steps, ep["length"] - 1), **kw)
        k = min(len(obs), ep["length"])
        runs[name] = (float(np.sqrt(((obs[1:k, :2] - ep["observ"][1:k, :2]) ** 2).mean())), d)
    name = min(runs, key=lambda s: runs[s][0])
    return name, runs[name][1], {s: r[0] for s, r in runs.items()}


EVENT_WINDOWS = ((3, 12), (13, 27), (45, 60), (61, 77), (78, 95), (96, 112))   # touch-down after the teleport, then the leg switches of
#                                             _open_loop_signal (every 0.1 s = 16.7 control steps of 6 ms); the broad third one (28-44) is left out


def rate_profile(episodes, replay, steps=120, **kw):
    """Median over the episodes of |(w_x, w_y)| per control step: the record's and the replay's (yaw-free, so no fit is involved)."""
    ours, ref, names = [], [], []
    for ep in episodes:
        name, d, _ = best_direction(ep, replay, **kw)
        obs, _, _ = replay(ep["action"], d, min(steps, ep["length"] - 1), **kw)
        pad = lambda a: np.pad(a, (0, steps + 1 - len(a)), constant_values=np.nan)   # noqa: E731
        ours.append(pad(np.hypot(obs[:steps + 1, 2], obs[:steps + 1, 3])))
        ref.append(pad(np.hypot(ep["observ"][:steps + 1, 2], ep["observ"][:steps + 1, 3])))
        names.append(name)
    return np.nanmedian(ref, 0), np.nanmedian(ours, 0), names


def direction_at(yaw, ccw):
    """(init_orient, target_orient) that starts at `yaw` and turns the given way with the target 2.5 rad off (`_solve_direction`)."""
    yaw = float(min(max(yaw % (2 * np.pi), 0.05), 6.2))
    t = yaw + 2.5 if ccw else yaw - 2.5
    if ccw and t > 6.2:
        t -= 2 * np.pi            # init > target, diff > 3.14: clockwise False
    if not ccw and t < 0.1:
        t += 2 * np.pi            # init < target, diff > 3.14: clockwise True
    return yaw, float(t)


def summarize(episodes, replay, steps=200, windows=(25, 50, 100, 200), at_fitted_yaw=False, **kw):
    """The record of a replay of every episode: per-window errors averaged over the episodes + the event timing."""
    rows = []
    for ep in episodes:
        name, d, fits = best_direction(ep, replay, **kw)
        obs, xy, ended = replay(ep["action"], d, min(steps, ep["length"] - 1), **kw)
        c = compare(ep, obs, xy, d[0], windows)
        for _ in range(2 if at_fitted_yaw else 0):      # (the world-aligned friction pyramid: replay from the episode's own start yaw)
            d = direction_at(c["start_yaw"], name == "ccw")
            obs, xy, ended = replay(ep["action"], d, min(steps, ep["length"] - 1), **kw)
            c = compare(ep, obs, xy, d[0], windows)
        c.update(direction=name, direction_fit_rmse=fits, ended=ended)
        rows.append(c)
    ref, ours, _ = rate_profile(episodes, replay, **kw)
    keys = ("rp_rmse", "rp_ref_rms", "rate_rmse", "rate_ref_rms", "reward_rmse")
    return dict(
        episodes=len(rows), steps_compared=int(sum(r["steps"] for r in rows)),
        windows={int(k): {m: float(np.mean([r[k][m] for r in rows])) for m in keys} for k in windows},
        directions=[r["direction"] for r in rows],
        direction_fit_ratio_min=float(min(max(r["direction_fit_rmse"].values()) / min(r["direction_fit_rmse"].values()) for r in rows)),
        rate_profile_correlation=float(np.corrcoef(ref[1:], ours[1:])[0, 1]),
        event_peaks={f"{lo}-{hi}": dict(record=int(lo + np.argmax(ref[lo:hi])), replay=int(lo + np.argmax(ours[lo:hi])),
                                        record_rad_s=float(ref[lo:hi].max()), replay_rad_s=float(ours[lo:hi].max())) for lo, hi in EVENT_WINDOWS},
        per_episode=[{k: v for k, v in r.items() if not isinstance(k, int)} for r in rows])      # (without the per-window blocks)


class HipReplayer:
    """replay() on the HIP path: all episodes of the record as one RexBatchEnv per turning direction (two launches per control step for
    the whole record), looked up per episode -- the same call shape as replay_oracle."""

    def __init__(self, episodes, steps=200, device=0, **env_kw):
        import torch
        from rex_gym_amd import RexBatchEnv
        self._by_action = {id(ep["action"]): k for k, ep in enumerate(episodes)}
        n = len(episodes)
        steps = min(steps, max(ep["length"] for ep in episodes) - 1)
        act = np.zeros((steps, n, 2), np.float32)
        for k, ep in enumerate(episodes):
            m = min(steps, ep["length"])
            act[:m, k] = ep["action"][:m]
        self.runs = {}
        for d in (CCW, CW):
            kw = dict(task="turn", signal_type="ol", device=device, range_normalize=True, init_orient=d[0], target_orient=d[1],
                      max_episode_steps=1000, action_repeat=ACTION_REPEAT, check_actions=False)
            kw.update(env_kw)
            env = RexBatchEnv(n, **kw)
            obs = [env.reset().double().cpu().numpy() * OBS_HI]
            xy = [env.state[:2].double().cpu().numpy().T.copy()]
            done_at = np.full(n, -1)
            a = torch.as_tensor(act, device=env.device)
            for t in range(steps):
                o, _, d_, _ = env.step(a[t])
                obs.append(o.double().cpu().numpy() * OBS_HI)
                xy.append(env.state[:2].double().cpu().numpy().T.copy())
                dn = d_.cpu().numpy().astype(bool)
                done_at[(done_at < 0) & dn] = t + 1
            env.close()
            self.runs[d] = (np.stack(obs, 1), np.stack(xy, 1), done_at)       # [n, steps + 1, .]

    def __call__(self, action, direction, steps, **ignored):
        k = self._by_action[id(action)]
        obs, xy, done_at = self.runs[direction]
        end = steps + 1 if done_at[k] < 0 else min(steps, int(done_at[k])) + 1
        return obs[k, :end], xy[k, :end], (int(done_at[k]) if 0 <= done_at[k] <= steps else None)


# ---------------------------------------------------------------------------------------------------------------------------------
# The second record: RexStandupEnv (signal 'ol') -- tests/golden/pybullet_standup_ol_rollouts.npz, 25 episodes x 400 control steps.
# No hidden draws (standup_env.py:108-118: the crouch of INIT_POSES['rest_position'], then `stand * ((0.1 + a) / (1 + t) + 1.5)` for
# 0.1 s, then INIT_POSES['stand']); the reward is the base position folded as standup_env.py:141-155 folds it.
STANDUP_FIXTURE = os.path.join(HERE, "golden", "pybullet_standup_ol_rollouts.npz")


def load_standup():
    d = np.load(STANDUP_FIXTURE)
    return [dict(observ=d["observ"][k, :n].astype(np.float64) * OBS_HI, action=d["action"][k, :n].astype(np.float64),
                 reward=d["reward"][k, :n].astype(np.float64), length=int(n)) for k, n in enumerate(d["length"])]


def standup_position_error(reward):
    """(|x| + |y| + |0.21 - z|, base above 0.21 m) out of the standup reward: r = 1 - e (e < 0.1) or -e below the target height,
    -1 - that above it (standup_env.py:146-153).  (-1, 0] is read as 'below' -- above the target with e >= 0.1 does not occur in the record.)"""
    r = np.asarray(reward, np.float64)
    above = r <= -1.0
    e = np.where(above, r + 2.0, np.where(r > 0, 1.0 - r, -r))
    return e, above


def replay_standup_oracle(action, steps, dtype=np.float64, probes=None, **cfg_kw):
    """-> (obs [steps + 1, 4] rad, rad/s; base position [steps + 1, 3]; reward [steps]; the step at which is_fallen ended it, or None)"""
    import ctypes
    import orclib
    kw = dict(num_envs=1, range_normalize=1, max_episode_steps=1000)
    kw.update(cfg_kw)
    cfg = orclib.default_config("standup", "ol", **kw)
    env = orclib.OracleEnv(cfg, dtype)
    if probes:
        env.o.lib.orc_set_probe.argtypes = [ctypes.c_char_p, ctypes.c_double]
        for k, v in probes.items():
            assert env.o.lib.orc_set_probe(k.encode(), float(v)) == 0, k
        env.close()
        env = orclib.OracleEnv(cfg, dtype)
    try:
        obs, pos, rew, fell = [env.reset()[0] * OBS_HI], [env.get_state()[:3, 0].copy()], [], None
        for t in range(steps):
            o, r, d, _ = env.step(action[t][None])
            obs.append(o[0] * OBS_HI)
            pos.append(env.get_state()[:3, 0].copy())
            rew.append(float(r[0]))
            if d[0]:
                fell = t + 1
                break
        return np.asarray(obs, np.float64), np.asarray(pos, np.float64), np.asarray(rew, np.float64), fell
    finally:
        env.close()


def summarize_standup(episodes, replay, steps=400, **kw):
    rows = []
    for ep in episodes:
        obs, pos, rew, fell = replay(ep["action"], min(steps, ep["length"]), **kw)
        n = len(rew)
        e_ref, above_ref = standup_position_error(ep["reward"][:n])
        e, above = standup_position_error(rew)
        k = min(n, 30)
        rows.append(dict(
            fell_at=fell, steps=n, return_replay=float(rew.sum()), return_record=float(ep["reward"][:n].sum()),
            return_record_400=float(ep["reward"].sum()),
            crouch_error_replay=float(e[0]), crouch_error_record=float(e_ref[0]),
            rise_mm_per_step_replay=float((e[2] - e[12]) / 10 * 1e3), rise_mm_per_step_record=float((e_ref[2] - e_ref[12]) / 10 * 1e3),
            first_above_replay=int(np.argmax(above)) if above.any() else None, first_above_record=int(np.argmax(above_ref)) if above_ref.any() else None,
            pitch_peak_replay=float(obs[1:k + 1, 1].max()), pitch_peak_record=float(ep["observ"][:k, 1].max()),
            pitch_rmse_30=float(np.sqrt(((obs[:k, 1] - ep["observ"][:k, 1]) ** 2).mean())),
            pitch_rmse_all=float(np.sqrt(((obs[:n, 1] - ep["observ"][:n, 1]) ** 2).mean()))))
    agg = {k: float(np.mean([r[k] for r in rows])) for k in rows[0] if k not in ("fell_at", "first_above_replay", "first_above_record")}
    agg["episodes"] = len(rows)
    agg["fell"] = int(sum(r["fell_at"] is not None for r in rows))
    agg["fell_at_median"] = float(np.median([r["fell_at"] for r in rows if r["fell_at"] is not None])) if agg["fell"] else None
    agg["first_above_replay_median"] = float(np.median([r["first_above_replay"] for r in rows if r["first_above_replay"] is not None]))
    agg["first_above_record_median"] = float(np.median([r["first_above_record"] for r in rows if r["first_above_record"] is not None]))
    return dict(summary=agg, per_episode=rows)


def replay_standup_hip(episodes, steps=400, device=0, **env_kw):
    """All standup episodes of the record as one RexBatchEnv.  -> list of (obs, None, reward, fell) per episode, as replay_standup_oracle returns."""
    import torch
    from rex_gym_amd import RexBatchEnv
    n = len(episodes)
    kw = dict(task="standup", signal_type="ol", device=device, range_normalize=True, max_episode_steps=1000, check_actions=False)
    kw.update(env_kw)
    env = RexBatchEnv(n, **kw)
    act = torch.as_tensor(np.stack([ep["action"][:steps] for ep in episodes], 1).astype(np.float32), device=env.device)    # [steps, n, 1]
    obs = [env.reset().double().cpu().numpy() * OBS_HI]
    rew, fell = [], np.full(n, -1)
    for t in range(steps):
        o, r, d, _ = env.step(act[t])
        obs.append(o.double().cpu().numpy() * OBS_HI)
        rew.append(r.double().cpu().numpy())
        dn = d.cpu().numpy().astype(bool)
        fell[(fell < 0) & dn] = t + 1
    env.close()
    obs, rew = np.stack(obs, 1), np.stack(rew, 1)
    out = []
    for k in range(n):
        end = steps if fell[k] < 0 else int(fell[k])
        out.append((obs[k, :end + 1], None, rew[k, :end], int(fell[k]) if fell[k] >= 0 else None))
    return out

"""The oracle's env-level command logic (walk / gallop / turn / poses / standup `_transform_action_to_motor_command`:
ramp-in, goal detection, brake, hold pose, delayed termination) against vectors produced by the reference's OWN methods
(tests/golden/make_env_golden.py drives the real rex_gym.envs.gym.* code with stand-ins for pybullet / gym).  The
sequences script time, base position / yaw and actions; the oracle is put into the same situation step by step and asked
for the motor command alone (orc_env_command: no physics)."""
import json
import math
import os

import numpy as np
import pytest

import orclib
from orclib import OracleEnv, default_config

HERE = os.path.dirname(os.path.abspath(__file__))
F_GOAL, F_TERM, F_STAY, F_ENV_GOAL = 1, 2, 4, 32


@pytest.fixture(scope="module")
def sequences():
    with open(os.path.join(HERE, "golden", "env_command_golden.json")) as f:
        return json.load(f)


def _config(seq):
    cfg_kw = dict(seq["cfg"])
    if "init_orient" in cfg_kw:
        cfg_kw["orient_fixed"] = 3
    return default_config(seq["task"], seq["signal"], 1, **cfg_kw)


def test_env_command_logic_matches_the_reference_methods(sequences):
    assert len(sequences) == 13
    worst = 0.0
    for seq in sequences:
        env = OracleEnv(_config(seq), np.float64)
        env.reset()
        label = (seq["task"], seq["signal"], seq["cfg"])
        for row in seq["rows"]:
            st = env.get_state()
            st[orclib.S_STEPS, 0] = row["k"]                       # t = steps * action_repeat * dt
            st[0, 0] = row["x"]
            st[3:7, 0] = [0.0, 0.0, math.sin(row["yaw"] / 2), math.cos(row["yaw"] / 2)]
            env.set_state(st)
            cmd = env.command(0, np.asarray(row["action"], np.float64))
            flags = int(env.get_state()[orclib.S_FLAGS, 0])
            got = (bool(flags & F_GOAL), bool(flags & F_TERM), bool(flags & F_STAY), bool(flags & F_ENV_GOAL))
            assert got == (row["goal"], row["terminating"], row["stay"], row["env_goal"]), (label, row["k"], got)
            if row["terminating"]:
                # the state keeps end_time as the env step it was taken at (rexsim.h, "Clocks"); its clock is that of the script
                end_t = env.get_state()[orclib.S_ENDTIME, 0] * env.cfg.action_repeat * 0.001
                assert abs(end_t - row["end_time"]) < 1e-12, (label, row["k"])
            if "cmd" in row:
                err = float(np.abs(cmd[:12] - np.asarray(row["cmd"])).max())
                # RexConfig.sim_time_step is a float32 (0.001f = 0.001 (1 + 4.7e-8)): the oracle's clock runs that much
                # ahead of the script's k * 0.005, worth up to ~1e-7 rad on a fast-moving joint target
                assert err < 5e-7, (label, row["k"], err)
                worst = max(worst, err)
        env.close()
    print("largest |command - reference| over all sequences: %.2e rad" % worst)


def test_forward_reward_cap_is_applied_to_the_forward_term_only():
    """RexGymEnv(forward_reward_cap=c): `forward_reward = min(forward_reward, c)` (rex_gym_env.py:525) -- the reward of a
    capped env equals that of the uncapped one minus distance_weight x the clipped part of the forward term."""
    n = 6
    envs = [OracleEnv(default_config("walk", "ik", n, seed=2, backwards=0, target_position=1.0, gait_clock_scale=1.5, forward_reward_cap=c), np.float64)
            for c in (float("inf"), 0.06)]
    for e in envs:
        e.reset()
    rng = np.random.RandomState(0)
    clipped = 0
    for k in range(200):
        a = rng.uniform(-0.4, 0.4, (n, 2))
        (_, r0, d0, _), (_, r1, d1, _) = envs[0].step(a), envs[1].step(a)
        x = -envs[0].get_state()[0]                       # the capped env's physics is the same
        fwd = np.where(x > 1.15, 1.0 - x, np.where(x >= 1.0, 1.0, np.where(x <= 0.05, 0.0, x / 1.0)))
        np.testing.assert_allclose(r1, r0 - 1.0 * (fwd - np.minimum(fwd, 0.06)), atol=1e-12)
        assert (d0 == d1).all()
        clipped += int((fwd > 0.06).sum())
    assert clipped > 100                                   # the cap did act
    for e in envs:
        e.close()

#!/usr/bin/env python3
"""Golden vectors for the ENV-LEVEL command logic, produced by the reference's own methods.

rex_gym/envs/gym/{walk,gallop,turn,poses,standup}_env.py import pybullet, pybullet_data and gym at module level, none
of which exist here; their `_transform_action_to_motor_command` / `_signal` / `_check_target_position` methods, however,
are plain functions of (time since reset, action, base pose, a few flags).  This script puts inert stand-ins for the
three modules into sys.modules (nothing of them is called on this path except getEulerFromQuaternion, which is fed
scripted Euler angles), builds each env object WITHOUT running its constructor, gives it the attributes its own
reset() would set, and drives the reference's real methods over scripted sequences of (t, action, base x / yaw).
The oracle replays the sequences in tests/test_oracle_env_commands.py.

Run in the build container:  PYTHONPATH=/root/reference python tests/golden/make_env_golden.py
"""
import json
import math
import os
import sys
import types

import numpy as np

np.math = math   # gait_planner.py:24 uses the numpy.math alias removed in numpy 2 (SURVEY.md 8c shim 1)
sys.path.insert(0, os.environ.get("REX_REFERENCE", "/root/reference"))
for name in ["pybullet", "pybullet_data", "gym", "gym.spaces", "gym.utils", "gym.utils.seeding"]:
    sys.modules[name] = types.ModuleType(name)
sys.modules["gym"].Env = type("Env", (), {})
sys.modules["gym.spaces"].Box = lambda low, high, dtype=None: (np.asarray(low), np.asarray(high))
sys.modules["gym"].spaces = sys.modules["gym.spaces"]
sys.modules["gym"].utils = sys.modules["gym.utils"]
sys.modules["gym.utils"].seeding = sys.modules["gym.utils.seeding"]
sys.modules["pybullet_data"].getDataPath = lambda: "/nonexistent"

import rex_gym.model.gait_planner as gp                      # noqa: E402
from rex_gym.model import rex_constants                      # noqa: E402
from rex_gym.model.kinematics import Kinematics              # noqa: E402
from rex_gym.envs.gym import gallop_env, poses_env, standup_env, turn_env, walk_env   # noqa: E402


class Clock:
    t = 0.0
clock = Clock()
gp.time.time = lambda: clock.t                               # the phase clock is simulation time (SURVEY.md 8c shim 2)


class FakeRex:
    """The three getters the command path reads."""
    def __init__(self):
        self.t, self.x, self.yaw = 0.0, 0.0, 0.0
        self.initial_pose = rex_constants.INIT_POSES["stand"]
    def GetTimeSinceReset(self): return self.t
    def GetBasePosition(self): return (self.x, 0.0, 0.2)
    def GetBaseOrientation(self): return ("quat-of-yaw", self.yaw)


class FakeBullet:
    def getEulerFromQuaternion(self, q): return (0.0, 0.0, q[1])


def bare(cls, signal, planner, **attrs):
    """An env object with the attributes RexGymEnv.__init__ / <Env>.reset() leave behind, constructor not run."""
    e = object.__new__(cls)
    e.rex, e.mark, e._signal_type = FakeRex(), "base", signal
    e._is_render = e._is_debug = False
    e._base_x, e._base_y, e._base_z, e._base_roll, e._base_pitch, e._base_yaw = 0.01, 0.0, 0.0, 0.0, 0.0, 0.0
    e.step_length = e.step_rotation = e.step_angle = e.step_period = None
    e.goal_reached = e.is_terminating = e._stay_still = False
    e.env_goal_reached = False
    e.init_pose = rex_constants.INIT_POSES["stand_ol" if signal == "ol" else "stand"]
    e._gait_planner, e._kinematics = (gp.GaitPlanner(planner) if planner else None), Kinematics()
    e._pybullet_client = FakeBullet()      # the envs read it through the `pybullet_client` property
    for k, v in attrs.items():
        setattr(e, k, v)
    return e


def drive(env, dt, steps, action_lo, action_hi, dim, seed, x_of=None, yaw_of=None):
    rng = np.random.RandomState(seed)
    rows = []
    for k in range(steps):
        t = k * dt
        clock.t = env.rex.t = t
        env.rex.x = x_of(t) if x_of else 0.0
        env.rex.yaw = yaw_of(t) if yaw_of else 0.0
        a = rng.uniform(action_lo, action_hi, dim)
        cmd = np.asarray(env._transform_action_to_motor_command(a.copy()), dtype=np.float64)
        rows.append({"k": k, "action": a.tolist(), "x": env.rex.x, "yaw": env.rex.yaw, "cmd": cmd.tolist(),
                     "goal": bool(env.goal_reached), "terminating": bool(env.is_terminating),
                     "stay": bool(env._stay_still), "end_time": float(getattr(env, "end_time", 0.0)),
                     "env_goal": bool(env.env_goal_reached)})
    return rows


out = []
W, G, T, P, S = walk_env.RexWalkEnv, gallop_env.RexReactiveEnv, turn_env.RexTurnEnv, poses_env.RexPosesEnv, standup_env.RexStandupEnv

# (targets and headings are float32-representable because RexConfig carries them as floats; slopes are chosen so that
#  no scripted position lands exactly on a threshold)
# ---- walk (dt 5 ms): ramp-in, goal at |x| >= |target| - 0.15, brake ramp, stay still (walk_env.py:207-324)
out.append({"task": "walk", "signal": "ik", "cfg": {"backwards": 0, "target_position": 0.5},
            "rows": drive(bare(W, "ik", "walk", backwards=False, _target_position=0.5), 0.005, 520, -0.4, 0.4, 2, 1,
                          x_of=lambda t: -0.2817 * t)})
out.append({"task": "walk", "signal": "ik", "cfg": {"backwards": 1, "target_position": -0.4375},
            "rows": drive(bare(W, "ik", "walk", backwards=True, _target_position=-0.4375), 0.005, 520, -0.4, 0.4, 2, 2,
                          x_of=lambda t: 0.1931 * t)})
out.append({"task": "walk", "signal": "ol", "cfg": {"backwards": 0, "target_position": 0.375},
            "rows": drive(bare(W, "ol", "walk", backwards=False, _target_position=0.375), 0.005, 500, -0.01, 0.01, 8, 3,
                          x_of=lambda t: -0.1877 * t)})
# ---- gallop (dt 6 ms): goal at |x| >= |target| (gallop_env.py:212-313); the OL branch scales the action in place
out.append({"task": "gallop", "signal": "ik", "cfg": {"target_position": 0.625},
            "rows": drive(bare(G, "ik", "gallop", _target_position=0.625), 0.006, 420, -0.4, 0.4, 2, 4, x_of=lambda t: -0.5113 * t)})
out.append({"task": "gallop", "signal": "ol", "cfg": {"target_position": 0.5},
            "rows": drive(bare(G, "ol", "gallop", _target_position=0.5), 0.006, 420, -0.3, 0.3, 4, 5, x_of=lambda t: -0.4139 * t)})
# ---- turn (dt 5 ms): goal when the yaw comes within 0.01 of the target, hold, env goal 1 s later (turn_env.py:239-347)
for signal, init, target, seed in (("ik", 1.0, 2.25, 6), ("ik", 5.5, 0.625, 7), ("ol", 0.375, 1.25, 8), ("ol", 2.0, 0.875, 9)):
    e = bare(T, signal, "walk", _init_orient=init, _target_orient=target)
    e.clockwise = e._solve_direction()
    # a yaw ramp from the start heading through the target (wrapping through 0 / 6.28 where the short way does)
    def yaw_of(t, init=init, target=target):
        diff = target - init
        if abs(diff) > 3.14:
            diff -= math.copysign(6.28, diff)
        y = init + diff * min(t / 1.2, 1.0)
        y = y % 6.28
        return y if y <= 3.14 else y - 6.28        # getEulerFromQuaternion range (-pi, pi]
    out.append({"task": "turn", "signal": signal, "cfg": {"init_orient": init, "target_orient": target},
                "rows": drive(e, 0.005, 520, -0.01, 0.01, 2, seed, yaw_of=yaw_of)})
# ---- poses (dt 6 ms): one body-pose component ramps to its value (poses_env.py:172-225)
for kw, idx, val, seed in (({"_base_y": 0.006}, 0, 0.006, 10), ({"_base_roll": 0.35}, 2, 0.35, 11), ({"_base_yaw": -0.3}, 4, -0.3, 12)):
    attrs = {"_base_y": 0.0, "_base_z": 0.0, "_base_roll": 0.0, "_base_pitch": 0.0, "_base_yaw": 0.0}
    attrs.update(kw)
    e = bare(P, "ik", None, manual_control=False, **attrs)
    e._ranges = {"base_x": (-0.02, 0.02, 0.01), "base_y": (-0.007, 0.007, 0), "base_z": (-0.048, 0.021, 0),
                 "roll": (-np.pi / 4, np.pi / 4, 0), "pitch": (-np.pi / 4, np.pi / 4, 0), "yaw": (-np.pi / 4, np.pi / 4, 0)}
    e.fill_next_pose_and_target()
    e.values = e._ranges.copy()
    out.append({"task": "poses", "signal": "ik", "cfg": {"pose_index": idx, "pose_value": val},
                "rows": drive(e, 0.006, 260, -0.1, 0.1, 1, seed)})
# ---- standup (dt 5 ms): the 'brake' overshoot of the stand pose (standup_env.py:113-134)
out.append({"task": "standup", "signal": "ol", "cfg": {}, "rows": drive(bare(S, "ol", None), 0.005, 60, -0.1, 0.1, 1, 13)})

# keep the fixture small: every step carries its inputs and flags (the replay needs them all), the 12-vector command is
# kept for every 4th step and for the steps around every flag change
for s in out:
    r = s["rows"]
    flags = [(x["goal"], x["terminating"], x["stay"], x["env_goal"]) for x in r]
    keep = {k for k in range(len(r)) if k % 4 == 0}
    for k in range(1, len(r)):
        if flags[k] != flags[k - 1]:
            keep.update(range(max(0, k - 3), min(len(r), k + 4)))
    for k, x in enumerate(r):
        if k not in keep:
            del x["cmd"]
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "env_command_golden.json")
with open(dst, "w") as f:
    json.dump(out, f, separators=(",", ":"))
print("wrote", dst, os.path.getsize(dst), "bytes;", sum(len(s["rows"]) for s in out), "command vectors in", len(out), "sequences")
for s in out:
    r = s["rows"]
    print(s["task"], s["signal"], s["cfg"], "goal at step", next((x["k"] for x in r if x["goal"]), None),
          "stay at", next((x["k"] for x in r if x["stay"]), None), "env_goal at", next((x["k"] for x in r if x["env_goal"]), None))

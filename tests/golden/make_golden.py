#!/usr/bin/env python3
"""Generate golden controller vectors by RUNNING the reference's own numpy code.

Runs only in the build container (needs /root/reference); the JSON it writes is committed so the
CPU and GPU test suites never touch /root/reference.  No reference source is copied: the modules
are imported and called.

Harness shims (no edits to the reference):
  * numpy.math = math            (gait_planner.py:24 uses np.math.factorial, gone in numpy 2)
  * gait_planner.time.time -> simulated clock (gait_planner.py:108-110 reads wall time)
"""
import json
import math
import os
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
sys.path.insert(0, "/root/reference")
np.math = math
import rex_gym.model.gait_planner as gp  # noqa: E402
from rex_gym.model.kinematics import Kinematics  # noqa: E402
from rex_gym.model.motor import MotorModel  # noqa: E402


class Clock:
    def __init__(self):
        self.t = 0.0

    def time(self):
        return self.t


clock = Clock()
gp.time = clock  # module attribute `time` -> object with .time()

out = {}
rng = np.random.RandomState(1234)

# ---------------- IK ----------------
ik_cases = []
k = Kinematics()
fixed = [([0, 0, 0], [0.01, 0, 0], None), ([0.1, -0.05, 0.2], [0.01, 0.005, -0.02], None)]
default_frames = np.array(k._frames)
for orn, pos, fr in fixed:
    ik_cases.append((np.array(orn, float), np.array(pos, float), default_frames.copy()))
for i in range(200):
    orn = rng.uniform(-0.4, 0.4, 3) if i % 4 else np.zeros(3)
    pos = rng.uniform(-0.03, 0.03, 3)
    fr = default_frames + rng.uniform(-0.06, 0.06, (4, 3))
    if i % 17 == 0:  # unreachable targets exercise check_domain / sqrt clamps
        fr = default_frames * rng.uniform(1.2, 2.0)
    if i % 19 == 0:
        fr = default_frames * 0.1
    ik_cases.append((orn, pos, fr))
ik = {"orn": [], "pos": [], "frames": [], "angles": [], "tframes": []}
for orn, pos, fr in ik_cases:
    kk = Kinematics()
    a = kk.solve(orn, pos, np.asmatrix(fr))
    ik["orn"].append(orn.tolist())
    ik["pos"].append(pos.tolist())
    ik["frames"].append(np.asarray(fr).tolist())
    ik["angles"].append(np.concatenate([np.asarray(x, float).ravel() for x in a[:4]]).tolist())
    ik["tframes"].append(np.asarray(a[4]).ravel().tolist())
out["ik"] = ik

# ---------------- motor ----------------
m = MotorModel(12, kp=1.0, kd=0.02)
mc = {"cmd": [], "q": [], "qd": [], "qd_true": [], "kp": 1.0, "kd": 0.02, "actual": [], "observed": []}
cases = [(np.full(12, 0.3), np.zeros(12), np.zeros(12), np.zeros(12)),
         (np.linspace(-.5, .6, 12), np.linspace(.2, -.2, 12), np.linspace(-3, 3, 12), np.linspace(-3, 3, 12))]
for i in range(60):
    cases.append((rng.uniform(-2, 2, 12), rng.uniform(-2, 2, 12), rng.uniform(-30, 30, 12), rng.uniform(-400, 400, 12)))
for i in range(20):
    cases.append((rng.uniform(-0.2, 0.2, 12), rng.uniform(-0.2, 0.2, 12), rng.uniform(-3, 3, 12), rng.uniform(-30, 30, 12)))
for cmd, q, qd, qdt in cases:
    act, obs = m.convert_to_torque(cmd, q, qd, qdt)
    mc["cmd"].append(cmd.tolist()); mc["q"].append(q.tolist()); mc["qd"].append(qd.tolist())
    mc["qd_true"].append(qdt.tolist()); mc["actual"].append(np.asarray(act).tolist()); mc["observed"].append(np.asarray(obs).tolist())
out["motor"] = mc

# ---------------- gait planner: stateful sequences on a simulated clock ----------------
def run_gait(mode, calls):
    """calls: list of (now, v, angle, w_rot, T, direction). Planner starts fresh with clock at 0."""
    g = gp.GaitPlanner(mode)
    res = []
    for now, v, angle, w_rot, T, direction in calls:
        clock.t = now
        fr = np.array(g.loop(v, angle, w_rot, T, direction)).copy()
        res.append({"now": now, "v": v, "angle": angle, "w_rot": w_rot, "T": T, "direction": direction,
                    "frames": fr.ravel().tolist(), "phi": float(g._phi), "last_time": float(g._last_time),
                    "alpha": float(g._alpha)})
    return res

seqs = []
# walk, forward, control steps of 5 ms for 2 s (phase wraps 3 times at T = 0.65)
seqs.append({"mode": "walk", "calls": run_gait("walk", [(i * 0.005, 0.6 * min(1.0, i * 0.005 if i * 0.005 <= 0.8 else 1.0), 0.0, 0.0, 0.65, 1.0) for i in range(400)])})
# walk, backwards
seqs.append({"mode": "walk", "calls": run_gait("walk", [(i * 0.005, -0.3, 0.0, 0.0, 0.5, -1.0) for i in range(300)])})
# walk planner with rotation + step angle (turn-style parameters; exercises alpha carry)
seqs.append({"mode": "walk", "calls": run_gait("walk", [(i * 0.005, 0.02, 0.0, -0.5 * min(1.0, i * 0.005) + 0.004, 0.75 - 0.006, 1.0) for i in range(400)])})
seqs.append({"mode": "walk", "calls": run_gait("walk", [(i * 0.005, 0.6, 10.0, 0.5, 0.65, 1.0) for i in range(300)])})
seqs.append({"mode": "walk", "calls": run_gait("walk", [(i * 0.005, 0.4, -35.0, -0.7, 0.4, 1.0) for i in range(300)])})
# gallop
seqs.append({"mode": "gallop", "calls": run_gait("gallop", [(i * 0.006, 1.3 * (i * 0.006 if i * 0.006 <= 1.0 else 1.0), 0.0, 0.0, 0.3, 1.0) for i in range(400)])})
# tiny period clamps to 0.01
seqs.append({"mode": "walk", "calls": run_gait("walk", [(i * 0.005, 0.3, 0.0, 0.0, 0.001, 1.0) for i in range(50)])})
out["gait"] = seqs

# ---------------- the SURVEY.md 8(c) known-answer vectors, re-derived here ----------------
ka = {}
g = gp.GaitPlanner("walk")
clock.t = 10.0; g.loop(0.6, 0, 0, 0.65, 1)
clock.t = 20.0; ka["GAIT-1"] = np.array(g.loop(0.6, 0, 0, 0.65, 1)).ravel().tolist()
clock.t = 20.0 + 0.65 * 0.3; ka["GAIT-2"] = np.array(g.loop(0.6, 0, 0, 0.65, 1)).ravel().tolist()
clock.t = 20.0 + 0.65 * 0.8; ka["GAIT-3"] = np.array(g.loop(0.6, 10, 0.5, 0.65, 1)).ravel().tolist()
ka["GAIT-3-alpha"] = float(g._alpha)
out["known_answers"] = ka

# ---------------- composed walk-IK signal: gait -> IK -> motor ordering (walk_env.py:252-290) ----------------
def walk_signal_sequence(backwards, a0, nsteps):
    g = gp.GaitPlanner("walk")
    kk = Kinematics()
    rows = []
    for i in range(nsteps):
        t = i * 0.005
        clock.t = t
        step, period, base_x = (-0.3, 0.5, 0.0) if backwards else (0.6, 0.65, 0.01)
        coeff = t if 0.0 <= t <= 0.8 + a0 else 1.0
        sl = step * coeff
        direction = -1.0 if sl < 0 else 1.0
        frames = g.loop(sl, 0.0, 0.0, period, direction)
        fr, fl, rr, rl, _ = kk.solve(np.array([0.0, 0.0, 0.0]), np.array([base_x, 0.0, 0.0]), frames)
        rows.append(np.concatenate([fl, fr, rl, rr]).astype(float).tolist())
    return rows

out["walk_signal"] = [
    {"backwards": False, "a0": 0.13, "cmd": walk_signal_sequence(False, 0.13, 400)},
    {"backwards": True, "a0": -0.3, "cmd": walk_signal_sequence(True, -0.3, 300)},
]

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "controller_golden.json")
with open(dst, "w") as f:
    json.dump(out, f)
print("wrote", dst, os.path.getsize(dst), "bytes")

#!/usr/bin/env python3
"""SURVEY.md 8(d) baseline (iii): the reference's OWN pure-Python controller path -- GaitPlanner.loop + Kinematics.solve + action_repeat x
MotorModel.convert_to_torque, what one walk-IK control step computes outside PyBullet -- timed on one host core.  Imports the reference
(/root/reference: the build container only; the GPU boxes do not have it, which is why bench.py cannot time it), with the two harness
shims of SURVEY 8(c) (numpy.math, a simulated clock).  Writes one JSON record:
    python tools/time_reference_controller.py [--reference /root/reference] [--seconds 10] > profiles/r06_reference_python_controller.json"""
import argparse
import json
import math
import os
import sys
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference"); ap.add_argument("--seconds", type=float, default=10.0)
    a = ap.parse_args()
    sys.path.insert(0, a.reference)
    import warnings
    warnings.filterwarnings("ignore")
    import numpy
    numpy.math = math                                           # gait_planner.py:24 uses np.math.factorial
    import rex_gym.model.gait_planner as gp
    from rex_gym.model.kinematics import Kinematics
    from rex_gym.model.motor import MotorModel
    clock = [100.0]
    gp.time.time = lambda: clock[0]                             # gait_planner.py:108-110 reads the wall clock
    planner, kin, motor = gp.GaitPlanner("walk"), Kinematics(), MotorModel(12, kp=1.0, kd=0.02)
    planner.loop(0.6, 0, 0, 0.65, 1); clock[0] += 10.0; planner.loop(0.6, 0, 0, 0.65, 1)      # latch _last_time
    q = numpy.zeros(12); qd = numpy.zeros(12)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < a.seconds:
        clock[0] += 0.005                                       # one control step of simulated time
        frames = planner.loop(0.6, 0, 0, 0.65, 1)               # walk_env.py:246-290 forward gait
        fr, fl, rr, rl, _ = kin.solve(numpy.zeros(3), numpy.array([0.01, 0.0, 0.0]), frames)
        cmd = numpy.concatenate([fl, fr, rl, rr])
        for _ in range(5):                                      # Rex.Step: action_repeat x ApplyAction (rex.py:158-163,568-641)
            motor.convert_to_torque(cmd, q, qd, qd)
        steps += 1
    dt = time.perf_counter() - t0
    print(json.dumps({"what": "SURVEY 8(d) baseline (iii): the reference's pure-Python walk-IK controller path per control step -- GaitPlanner.loop + "
                              "Kinematics.solve + 5 x MotorModel.convert_to_torque -- on ONE core of the build container (no physics: PyBullet is not "
                              "installable)", "control_steps_per_s": steps / dt, "steps": steps, "seconds": dt, "cores": 1,
                      "host": os.uname().machine, "python": sys.version.split()[0], "numpy": numpy.__version__,
                      "reference": "nicrusso7/rex-gym @ /root/reference (rex_gym/model/gait_planner.py, kinematics.py, motor.py)"}))


if __name__ == "__main__":
    main()

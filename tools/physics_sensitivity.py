#!/usr/bin/env python3
"""Sensitivity of the restated stepSimulation to every Bullet-behaviour assumption of SURVEY.md 9.2.

TEST INFRASTRUCTURE (drives oracle/rex_oracle.c through its probe setters; nothing here is product code).

For each probe setting it plays the forward walk-IK env (RexWalkEnv, signal 'ik', the BASELINE workload) on the fp64
oracle with uniform-random actions, 16 envs x 2 500 control steps, and reports
  * the fraction of envs that never trip `is_fallen` (walk_env.py:326-338), the mean distance walked toward -x,
  * the same at 1.5 x gait clock (GaitPlanner.loop runs on wall-clock time, gait_planner.py:108-110: `gait_clock` is
    the wall-clock seconds that pass per simulated second on the host that runs the reference),
  * the smallest gait clock on a 0.05 grid at which >= 90 % of the envs survive.
Writes a markdown table (profiles/r02_sensitivity.md by default).
"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from orclib import OracleEnv, default_config  # noqa: E402

N_ENVS = 16
STEPS = 2500

DEFAULTS = dict(erp=0.2, slop=0.0, inertia_scale=1.0, leg_inertia_add=0.0, toe_mode=2, limit_exact=1, breaking=0.02,
                margin=0.001, friction_dirs=2, cone=0, gait_clock=1.0, mu=0.5, lin_damping=0.04, ang_damping=0.04,
                joint_friction=0.0, body_contacts=0)


def play(probes, clock, task="walk", kp=1.0, kd=0.02, iters=60, thr=1e-7, steps=STEPS, n=N_ENVS):
    cfg = default_config(task, "ik", num_envs=n, backwards=0, target_position=3.0 if task == "walk" else 0.0,
                         motor_kp=kp, motor_kd=kd, solver_iterations=iters, solver_residual_threshold=thr)
    env = OracleEnv(cfg)
    lib = env.o.lib
    lib.orc_set_probe.argtypes = [ctypes.c_char_p, ctypes.c_double]
    allp = dict(DEFAULTS)
    allp.update(probes)
    allp["gait_clock"] = clock
    for k, v in allp.items():
        assert lib.orc_set_probe(k.encode(), float(v)) == 0, k
    env.close()
    env = OracleEnv(cfg)   # settle again under the probes
    env.reset()
    rng = np.random.default_rng(0)
    alive = np.ones(n, bool)
    dist = np.zeros(n)
    b = 0.4 if task == "walk" else 0.01
    for _ in range(steps):
        obs, rew, done, cmd = env.step(rng.uniform(-b, b, (n, 2)))
        st = env.get_state()
        fell = done & alive & (np.abs(st[0]) < 2.8)        # done before the goal zone = is_fallen
        dist[alive] = -st[0][alive]
        alive &= ~done
        if not alive.any():
            break
    for k, v in DEFAULTS.items():
        lib.orc_set_probe(k.encode(), float(v))
    env.close()
    reached = dist >= 2.8
    return float(np.mean(alive | reached)), float(dist.mean())


def threshold_clock(probes, **kw):
    for c in np.arange(1.0, 2.01, 0.05):
        ok, _ = play(probes, float(c), **kw)
        if ok >= 0.9:
            return float(c)
    return float("nan")


ROWS = [
    ("restatement as shipped", {}, {}),
    ("contact ERP 0.08 (PyBullet's createEmptyDynamicsWorld value as recalled) instead of 0.2", dict(erp=0.08), {}),
    ("linear slop 1e-5", dict(slop=1e-5), {}),
    ("friction mu 0.25", dict(mu=0.25), {}),
    ("friction mu 1.0", dict(mu=1.0), {}),
    ("friction mu 100 (if <contact_coefficients mu> were honoured)", dict(mu=100.0), {}),
    ("friction cone instead of pyramid", dict(cone=1), {}),
    ("one friction direction (no SOLVER_USE_2_FRICTION_DIRECTIONS)", dict(friction_dirs=1), {}),
    ("multibody damping 0 / 0", dict(lin_damping=0.0, ang_damping=0.0), {}),
    ("multibody damping 0.4 / 0.4", dict(lin_damping=0.4, ang_damping=0.4), {}),
    ("link inertias x 0.5", dict(inertia_scale=0.5), {}),
    ("link inertias x 2", dict(inertia_scale=2.0), {}),
    ("link inertias x 10", dict(inertia_scale=10.0), {}),
    ("leg links + 1e-3 kg m^2 (rotor-like armature)", dict(leg_inertia_add=1e-3), {}),
    ("toe manifold: 1 point (lower end)", dict(toe_mode=1), {}),
    ("toe manifold: 6 points (ends + arc points +-0.3 rad)", dict(toe_mode=4), {}),
    ("toe collision margin 0", dict(margin=0.0), {}),
    ("toe collision margin 4 mm", dict(margin=0.004), {}),
    ("contact breaking threshold 0 (penetrating points only)", dict(breaking=0.0), {}),
    ("joint-limit rows from 0.15 rad before the bound (predictive) instead of once reached (Bullet's literal rule)", dict(limit_exact=0), {}),
    ("URDF joint friction honoured: 0.5 N m Coulomb on shoulder / foot joints", dict(joint_friction=0.5), {}),
    ("body-vs-ground box contacts on", dict(body_contacts=1), {}),
    ("solver: 10 sweeps", {}, dict(iters=10)),
    ("solver: 200 sweeps, no residual exit", {}, dict(iters=200, thr=0.0)),
    ("motor gains kp 1.5, kd 0.03", {}, dict(kp=1.5, kd=0.03)),
    ("motor gains kp 3, kd 0.1", {}, dict(kp=3.0, kd=0.1)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r02_sensitivity.md"))
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    lines = ["# Sensitivity of the restated `stepSimulation` (fp64 oracle), forward walk-IK, 16 envs x 2 500 steps, random actions",
             "",
             "`ok` = fraction of envs that never trip `is_fallen`; `dist` = mean distance walked toward -x [m] (target 3 m);",
             "`clock` = wall-clock seconds per simulated second seen by `GaitPlanner.loop` (`gait_planner.py:108-110`);",
             "`min clock` = smallest clock on a 0.05 grid (1.0 .. 2.0) with ok >= 0.9.  Generated by `tools/physics_sensitivity.py`.",
             "",
             "| assumption varied | ok @ clock 1.0 | dist @ 1.0 | ok @ clock 1.5 | dist @ 1.5 | min clock |",
             "|---|---|---|---|---|---|"]
    for name, probes, kw in ROWS:
        ok1, d1 = play(probes, 1.0, **kw)
        ok2, d2 = play(probes, 1.5, **kw)
        mc = float("nan") if args.quick else threshold_clock(probes, **kw)
        line = f"| {name} | {ok1:.2f} | {d1:.2f} | {ok2:.2f} | {d2:.2f} | {mc:.2f} |"
        print(line, flush=True)
        lines.append(line)
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The restated `stepSimulation` against the RECORDED PYBULLET ROLLOUTS (tests/golden/pybullet_turn_ol_rollouts.npz), assumption by assumption.

TEST INFRASTRUCTURE (drives oracle/rex_oracle.c through tests/pybullet_replay.py and the oracle's probe setters; nothing here is product code).
tools/physics_sensitivity.py asked how much each Bullet-behaviour assumption of SURVEY.md 9.2 moves a walking robot; this asks which value
the one piece of real PyBullet data in the reference supports: every row replays the record's 20 episodes on the fp64 oracle with one
assumption changed and reports how far roll / pitch, the world-frame rates and the |x| + |y| drift end up from the record.

    python tools/pybullet_record_report.py            -> profiles/r06_pybullet_record.json, profiles/r06_pybullet_record.md
"""
import json
import multiprocessing as mp
import os
import sys

os.environ.setdefault("OMP_NUM_THREADS", "1")          # one env per replay: the oracle's OpenMP team would only spin
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))

ROWS = [
    ("restatement as shipped (6 substeps per control step, recovered from the record)", {}, {}),
    ("as shipped, every episode replayed from its fitted start yaw (the friction pyramid's directions are world axes)", {}, dict(at_fitted_yaw=True)),
    ("5 substeps per control step (today's constructor default)", {}, dict(action_repeat=5, solver_iterations=60)),
    ("friction mu 0.25", dict(mu=0.25), {}),
    ("friction mu 1.0", dict(mu=1.0), {}),
    ("friction mu 100 (if <contact_coefficients mu> were honoured)", dict(mu=100.0), {}),
    ("friction cone instead of pyramid", dict(cone=1), {}),
    ("friction cone, mu 0.35", dict(cone=1, mu=0.35), {}),
    ("friction cone, mu 0.3", dict(cone=1, mu=0.3), {}),
    ("friction mu 0.35 (pyramid)", dict(mu=0.35), {}),
    ("one friction direction", dict(friction_dirs=1), {}),
    ("contact ERP 0.08", dict(erp=0.08), {}),
    ("multibody damping 0 / 0", dict(lin_damping=0.0, ang_damping=0.0), {}),
    ("multibody damping 0.4 / 0.4", dict(lin_damping=0.4, ang_damping=0.4), {}),
    ("link inertias x 0.5", dict(inertia_scale=0.5), {}),
    ("link inertias x 2", dict(inertia_scale=2.0), {}),
    ("link inertias x 10", dict(inertia_scale=10.0), {}),
    ("leg links + 1e-3 kg m^2", dict(leg_inertia_add=1e-3), {}),
    ("toe manifold: 1 point", dict(toe_mode=1), {}),
    ("toe collision margin 0", dict(margin=0.0), {}),
    ("toe collision margin 4 mm", dict(margin=0.004), {}),
    ("contact breaking threshold 0", dict(breaking=0.0), {}),
    ("URDF joint friction honoured (0.5 N m on shoulder / foot joints)", dict(joint_friction=0.5), {}),
    ("body-vs-ground box contacts on", dict(body_contacts=1), {}),
    ("self-collision rows on", dict(self_collision=1), {}),
    ("solver: 10 sweeps", {}, dict(solver_iterations=10)),
    ("solver: 200 sweeps, no residual exit", {}, dict(solver_iterations=200, solver_residual_threshold=0.0)),
    ("motor gains kp 1.5, kd 0.03", {}, dict(motor_kp=1.5, motor_kd=0.03)),
    ("motor gains kp 0.7", {}, dict(motor_kp=0.7)),
    ("motor kd 0.01", {}, dict(motor_kd=0.01)),
    ("motor kd 0.04", {}, dict(motor_kd=0.04)),
]


STANDUP_ROWS = [
    ("restatement as shipped (toe friction 0.5)", {}, {}),
    ("friction mu 0.15", dict(mu=0.15), {}), ("friction mu 0.2", dict(mu=0.2), {}), ("friction mu 0.25", dict(mu=0.25), {}),
    ("friction mu 0.3", dict(mu=0.3), {}), ("friction mu 0.35", dict(mu=0.35), {}), ("friction mu 0.4", dict(mu=0.4), {}),
    ("friction mu 1.0", dict(mu=1.0), {}),
    ("friction cone instead of pyramid", dict(cone=1), {}),
    ("friction cone, mu 0.4", dict(cone=1, mu=0.4), {}),
    ("friction cone, mu 0.35", dict(cone=1, mu=0.35), {}),
    ("friction cone, mu 0.3", dict(cone=1, mu=0.3), {}),
    ("6 substeps per control step", {}, dict(action_repeat=6, solver_iterations=50)),
    ("multibody damping 0.4 / 0.4", dict(lin_damping=0.4, ang_damping=0.4), {}),
    ("link inertias x 2", dict(inertia_scale=2.0), {}),
    ("leg links + 1e-3 kg m^2", dict(leg_inertia_add=1e-3), {}),
    ("URDF joint friction honoured (0.5 N m on shoulder / foot joints)", dict(joint_friction=0.5), {}),
    ("body-vs-ground box contacts on", {}, dict(body_contacts=1)),
    ("toe collision margin 4 mm", dict(margin=0.004), {}),
    ("motor gains kp 1.5, kd 0.03", {}, dict(motor_kp=1.5, motor_kd=0.03)),
    ("solver: 200 sweeps, no residual exit", {}, dict(solver_iterations=200, solver_residual_threshold=0.0)),
]


def work_standup(row):
    import pybullet_replay as pr
    name, probes, kw = row
    return name, pr.summarize_standup(pr.load_standup()[:8], pr.replay_standup_oracle, steps=400, probes=probes or None, **kw)["summary"]


def work(row):
    import pybullet_replay as pr
    name, probes, kw = row
    kw = dict(kw)
    fitted = kw.pop("at_fitted_yaw", False)
    s = pr.summarize(pr.load(), pr.replay_oracle, steps=120, windows=(25, 50, 100), at_fitted_yaw=fitted, probes=probes or None, **kw)
    return name, s


def main():
    out = os.path.join(ROOT, "profiles", "r06_pybullet_record")
    with mp.get_context("spawn").Pool(min(8, os.cpu_count() or 1), maxtasksperchild=1) as pool:      # (the probes are process-wide statics of the oracle)
        results = list(pool.imap(work, ROWS, chunksize=1))
        standup = list(pool.imap(work_standup, STANDUP_ROWS, chunksize=1))
    lines = ["# The restated `stepSimulation` against the recorded PyBullet rollouts (fp64 oracle; `tools/pybullet_record_report.py`)", "",
             "20 episodes of the reference's RexTurnEnv (signal 'ol') on real PyBullet, out of the episode memory of its shipped checkpoint",
             "(`tests/golden/make_pybullet_golden.py`); the recorded actions replayed, first 25 / 50 / 100 control steps of every episode.",
             "`rp` = RMS error of roll and pitch [rad] (the record's own RMS over the 25-step window: 6.8e-3), `rate` = RMS error of the world-frame",
             "(w_x, w_y) [rad/s] after the start-yaw fit (record RMS 0.27), `drift` = RMS error of |x| + |y| [m], `corr` = correlation of the median",
             "|w| profile over 120 steps with the record's, `events` = gait events (touch-down, leg switches) whose |w| peak falls on the record's control step.", "",
             "| assumption varied | rp 25 | rp 50 | rp 100 | rate 25 | drift 25 | corr | events |", "|---|---|---|---|---|---|---|---|"]
    for name, s in results:
        w = s["windows"]
        ev = s["event_peaks"]
        hit = sum(1 for e in ev.values() if e["record"] == e["replay"])
        line = (f"| {name} | {w[25]['rp_rmse']:.2e} | {w[50]['rp_rmse']:.2e} | {w[100]['rp_rmse']:.2e} | {w[25]['rate_rmse']:.3f} | "
                f"{w[25]['reward_rmse']:.1e} | {s['rate_profile_correlation']:.2f} | {hit} / {len(ev)} |")
        print(line, flush=True)
        lines.append(line)
    lines += ["", "## The standup record (25 episodes x 400 control steps of RexStandupEnv, no hidden draws; 8 episodes replayed per row)", "",
              "On PyBullet every recorded episode stands up and stays up (return +304 ... +345, mean +333).  `fell` = replays that trip `is_fallen`",
              "(and the median control step), `return` = mean episode return of the replay over the steps it lasted, `crouch` = |x| + |y| + |0.21 - z| after the",
              "first step (record: 0.184), `rise` = mm per control step over steps 2-12 (record: 6.9), `pitch 30` / `pitch all` = RMS error of the pitch [rad].", "",
              "| assumption varied | fell | return | crouch | rise | pitch 30 | pitch all |", "|---|---|---|---|---|---|---|"]
    for name, a in standup:
        fell = f"{a['fell']} / {a['episodes']}" + (f" (step {a['fell_at_median']:.0f})" if a["fell"] else "")
        line = (f"| {name} | {fell} | {a['return_replay']:+.0f} | {a['crouch_error_replay']:.3f} | {a['rise_mm_per_step_replay']:.1f} | "
                f"{a['pitch_rmse_30']:.3f} | {a['pitch_rmse_all']:.3f} |")
        print(line, flush=True)
        lines.append(line)
    with open(out + ".md", "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(out + ".json", "w") as f:
        json.dump({"what": __doc__.split("\n")[0], "standup": dict(standup), "shipped": results[0][1], "variants": {n: {"windows": s["windows"], "rate_profile_correlation": s["rate_profile_correlation"], "event_peaks": s["event_peaks"]} for n, s in results[1:]}}, f, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""profiles/r06_parity.json: joint-angle RMSE of the HIP path against the fp64 oracle over the 200-step window of every
BASELINE config (records written by tests/test_gpu_parity.py::test_every_kernel_variant_... into gpurun_out/parity_windows.jsonl
on the GPU box), next to the FLOAT32 FLOOR of each workload: the same window, the fp32 build of the oracle against its own
fp64 build (CPU, --floor) -- what float32 arithmetic costs on that workload whatever the implementation.  Round 4: both
comparisons are split by the EVENT TRACE (rex_set_event_trace / orc_set_event_trace): the envs whose discrete decisions --
toe points in reach, heightfield facets, joint / arm bounds reached, controller flags, substep by substep -- were those of the
fp64 oracle over the whole window, and the others.

    python tools/parity_report.py --floor            # CPU: compute the floors (minutes), merge, write profiles/r06_parity.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib                                                    # noqa: E402

WORKLOADS = {   # name: (envs, oracle config keywords, mark, action bound, heightfield?) -- mirrors tests/parity_window.py
    "walk_ik_4096": (4096, dict(task="walk", signal="ik"), "base", 0.4, False),
    "walk_ik_8192": (8192, dict(task="walk", signal="ik"), "base", 0.4, False),
    "gallop_ol_8192": (8192, dict(task="gallop", signal="ol"), "base", 0.3, False),
    "turn_ik_heightfield_4096": (4096, dict(task="turn", signal="ik"), "base", 0.01, True),
    "mixed_arm_2048": (2048, dict(task="mixed", signal="ik", mark=1, task_mix=0b111, action_repeat=6, solver_iterations=60,
                                  mass_scale_lo=0.8, mass_scale_hi=1.2, friction_lo=0.25, friction_hi=0.625), "arm", 0.01, False),
    "walk_ik_gait_clock_1.5_4096": (4096, dict(task="walk", signal="ik", gait_clock_scale=1.5), "base", 0.4, False),
}


def floor(name, steps=200, seed=23, threads=8):
    n, kw, mark, bound, hf = WORKLOADS[name]
    kw = dict(kw); task, signal = kw.pop("task"), kw.pop("signal")
    envs = []
    for dt in (np.float32, np.float64):
        e = orclib.OracleEnv(orclib.default_config(task, signal, n, seed=seed, **kw), dt, mark)
        e.o.lib.orc_set_threads(threads)
        if hf:
            from rex_gym_amd.terrain import random_terrain_pool
            e.set_terrain(*random_terrain_pool(64, 10))
        e.reset()
        envs.append(e)
    a32, a64 = envs
    nm = a32.o.num_motors
    rng = np.random.RandomState(seed)
    acts = rng.uniform(-bound, bound, (steps, n, a32.action_dim)).astype(np.float32)
    sq = np.zeros(n); cnt = np.zeros(n); alive = np.ones(n, bool); sql = np.zeros(n)
    t32, t64 = a32.set_event_trace(True), a64.set_event_trace(True)
    same = np.ones(n, bool); sq_same = np.zeros(n); cnt_same = np.zeros(n)
    for k in range(steps):
        _, _, d, _ = a32.step(acts[k]); _, _, od, _ = a64.step(acts[k])
        e = a32.get_state()[13:13 + nm] - a64.get_state()[13:13 + nm]
        same &= (t32[0] == t64[0]) | ~alive
        sq += np.where(alive, (e * e).mean(0), 0.0); cnt += alive
        sq_same += np.where(alive & same, (e * e).mean(0), 0.0); cnt_same += alive & same
        sql += np.where(alive, (e[:12] * e[:12]).mean(0), 0.0)
        alive &= ~(d | od)
    rmse = np.sqrt(sq / np.maximum(cnt, 1))
    for e in envs:
        e.close()
    legs = np.sqrt(sql / np.maximum(cnt, 1))

    def stats(x):
        return dict(envs=int(x.size), median_rad=float(np.median(x)), p99_rad=float(np.percentile(x, 99)), max_rad=float(x.max())) if x.size else dict(envs=0)
    return dict(median_rad=float(np.median(rmse)), p99_rad=float(np.percentile(rmse, 99)), max_rad=float(rmse.max()),
                leg_joints_median_rad=float(np.median(legs)), leg_joints_p99_rad=float(np.percentile(legs, 99)),
                envs_compared_to_the_end=int(alive.sum()),
                events=dict(share_same_event_sequence=float(same.mean()), joint_rmse_same_events=stats(rmse[same]),
                            joint_rmse_other_events=stats(rmse[~same]),
                            joint_rmse_until_first_divergence=dict(stats(np.sqrt(sq_same[cnt_same > 0] / cnt_same[cnt_same > 0])),
                                                                   mean_steps=float(cnt_same.mean()))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--floor", action="store_true")
    ap.add_argument("--records", default=os.path.join(ROOT, "gpurun_out", "parity_windows.jsonl"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_parity.json"))
    a = ap.parse_args()
    out = {"_comment": "per-env joint-angle RMSE (rad) over the first 200 control steps from reset, HIP path -- the PRODUCT kernels, no event trace set; the event split is a second "
                       "pass with the _trace instantiations, bit-identical to the first after every step (trace_pass_bit_identical) -- vs the fp64 oracle "
                       "(oracle/rex_oracle.c: a restatement -- PyBullet is not installable), every kernel variant, MI355X; "
                       "float32_floor = the oracle's own fp32 build vs its fp64 build over the same window (CPU). BASELINE.json's bar: 1e-3 rad. "
                       "events = the same numbers split by the event trace (include/rexsim.h rex_set_event_trace): envs whose toe points in "
                       "reach, heightfield facets, joint / arm bounds reached and controller flags were the fp64 oracle's in every substep of "
                       "the window, the others, and every env up to the step at which its event sequence parted.",
           "workloads": {}}
    if os.path.exists(a.out):
        try:
            out["workloads"] = json.load(open(a.out)).get("workloads", {})
        except ValueError:
            pass
    if os.path.exists(a.records):
        for line in open(a.records):
            r = json.loads(line)
            w = out["workloads"].setdefault(r["workload"], {"envs": r["envs"], "window_steps": r["window_steps"], "hip_vs_fp64_oracle": {}})
            w["hip_vs_fp64_oracle"][f"{r['envs_per_wave']}_envs_per_wave"] = {k: r.get(k) for k in (
                "median_rad", "p99_rad", "max_rad", "leg_joints_median_rad", "leg_joints_p99_rad", "base_pos_err_p99_m", "base_pos_err_max_m", "envs_compared_to_the_end", "mean_steps_compared",
                "abs_error_by_step", "events", "single_steps_from_common_states", "kernels", "trace_pass_bit_identical", "trace_pass_identical_steps", "float32_floor",
                "meets_1e-3_rad_absolute")}
            if r.get("float32_floor") and "float32_floor" not in w:
                w["float32_floor"] = r["float32_floor"]        # (the floor of the window, measured in the same test run)
    if a.floor:
        for name in WORKLOADS:
            w = out["workloads"].setdefault(name, {"envs": WORKLOADS[name][0], "window_steps": 200, "hip_vs_fp64_oracle": {}})
            w["float32_floor"] = floor(name)
            print(name, w["float32_floor"], flush=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    for name, w in out["workloads"].items():
        fl = w.get("float32_floor")
        print(name, "| floor", fl and "%.1e / %.1e / %.1e" % (fl["median_rad"], fl["p99_rad"], fl["max_rad"]), "| hip",
              {k: "%.1e / %.1e / %.1e" % (v["median_rad"], v["p99_rad"], v["max_rad"]) for k, v in w["hip_vs_fp64_oracle"].items()})
        ev = lambda e: e and "share %.3f, on it %s, until divergence p99 %.1e" % (
            e["share_same_event_sequence"], e["joint_rmse_same_events"].get("p99_rad") and "p99 %.1e max %.1e" % (e["joint_rmse_same_events"]["p99_rad"], e["joint_rmse_same_events"]["max_rad"]),
            e["joint_rmse_until_first_divergence"].get("p99_rad", float("nan")))
        print("   same events | floor:", ev(fl and fl.get("events")), "| hip:", {k: ev(v.get("events")) for k, v in w["hip_vs_fp64_oracle"].items()})


if __name__ == "__main__":
    main()

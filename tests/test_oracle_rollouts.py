"""The oracle's orchestration against the reference's own env / robot code (tests/golden/make_rollout_golden.py).

The fixture holds rollouts of the reference's unmodified RexGymEnv / Rex / env-subclass code over a physics stand-in
whose `stepSimulation` is the oracle's rigid-body step; here the oracle's own reset / step (`orc_reset`, `orc_step`)
replays the same actions.  Agreement pins reset (drop, settle counts, teleports), action repeat, PD observation and
latency blending, motor model composition, overheat bookkeeping, observation assembly, reward and termination of all
five envs to the reference -- everything on the hot path except what happens inside `stepSimulation`.
"""
import json
import os

import numpy as np
import pytest

import orclib

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "rollout_golden.json")) as f:
    GOLDEN = json.load(f)

# Both sides run the same fp64 rigid-body code; they differ in who computes the torques (numpy vs C, last-bit
# differences) and contacts amplify that along a rollout, so the tolerance is far above 1 ulp and far below anything
# a logic difference (an off-by-one substep, a wrong observation slot) would produce (>= 1e-4).
TOL = 2e-7


@pytest.mark.parametrize("sc", GOLDEN["scenarios"], ids=[s["name"] for s in GOLDEN["scenarios"]])
def test_oracle_reproduces_reference_rollout(sc):
    kw = dict(sc["oracle_config"])
    mark = "arm" if kw.get("mark", 0) == 1 else "base"
    n = sc["env_kwargs"].get("batch", 1)        # batch scenarios: N envs behind the reference's BatchEnv, resets by index
    cfg = orclib.default_config(kw.pop("task"), kw.pop("signal"), num_envs=n, auto_reset=0, **kw)
    env = orclib.OracleEnv(cfg, np.float64, mark)
    nm = env.o.num_motors
    worst = dict(obs=0.0, reward=0.0, cmd=0.0, body=0.0)
    batched = "batch" in sc["env_kwargs"]
    shape = (lambda x: np.asarray(x)) if batched else (lambda x: np.asarray(x)[None])
    obs = np.zeros((n, env.obs_dim))
    for k, ev in enumerate(sc["events"]):
        if ev["kind"] == "reset":
            idx = ev.get("indices")
            if idx is None:
                obs = env.reset()
                ref_obs = shape(ev["obs"])
            else:                                     # BatchEnv.reset(indices) returns the rows of those envs only
                obs = env.reset(idx)
                ref_obs = np.asarray(ev["obs"])
        else:
            obs, r, d, c = env.step(shape(ev["action"]))
            ref_obs = shape(ev["obs"])
            assert d.tolist() == np.atleast_1d(ev["done"]).tolist(), f"event {k}: done"
            worst["reward"] = max(worst["reward"], float(np.max(np.abs(r - np.atleast_1d(ev["reward"])))))
            worst["cmd"] = max(worst["cmd"], float(np.max(np.abs(c - shape(ev["cmd"])))))
        body = env.get_state()[:13 + 2 * nm].T
        worst["obs"] = max(worst["obs"], float(np.max(np.abs(obs - ref_obs))))
        ref = shape(ev["body"])
        scale = np.maximum(1.0, np.abs(ref))            # velocities are O(1..10): compare relative to their scale
        worst["body"] = max(worst["body"], float(np.max(np.abs(body - ref) / scale)))
        assert max(worst.values()) < 1e-3, f"event {k} ({ev['kind']}): {worst}"
    print(sc["name"], worst)
    assert max(worst.values()) < TOL, worst


PYBULLET_GOLDEN = os.path.join(HERE, "golden", "pybullet_rollout_golden.json")


@pytest.mark.skipif(not os.path.exists(PYBULLET_GOLDEN),
                    reason="no pybullet rollouts committed (generate with make_rollout_golden.py --real-pybullet where pybullet "
                           "is installed): the physics half stays 'parity unpinned'")
def test_oracle_against_real_pybullet_rollouts():
    """The fixture that would pin the physics: the same scenarios on the real engine.  Reports the joint-angle RMSE and
    base-position error of the oracle over the first 200 control steps of every scenario (BASELINE.json's metric)."""
    with open(PYBULLET_GOLDEN) as f:
        scenarios = json.load(f)["scenarios"]
    report = {}
    for sc in scenarios:
        kw = dict(sc["oracle_config"])
        mark = "arm" if kw.get("mark", 0) == 1 else "base"
        env = orclib.OracleEnv(orclib.default_config(kw.pop("task"), kw.pop("signal"), num_envs=1, auto_reset=0, **kw), np.float64, mark)
        nm = env.o.num_motors
        sq, n, pos = 0.0, 0, 0.0
        for ev in sc["events"][:201]:
            if ev["kind"] == "reset":
                env.reset()
            else:
                env.step(np.asarray(ev["action"])[None, :])
            st, ref = env.get_state()[:, 0], np.asarray(ev["body"])
            sq += float(np.sum((st[13:13 + nm] - ref[13:13 + nm]) ** 2)); n += nm
            pos = max(pos, float(np.max(np.abs(st[0:3] - ref[0:3]))))
        report[sc["name"]] = dict(joint_rmse=(sq / n) ** 0.5, base_pos_err=pos)
    print(json.dumps(report, indent=1))
    assert max(r["joint_rmse"] for r in report.values()) < 1e-3, report     # BASELINE.json: joint-angle RMSE < 1e-3 rad

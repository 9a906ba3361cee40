"""The HIP path against the recorded PyBullet rollouts (tests/test_oracle_pybullet_record.py has the story): the same replay, the same
bars -- through the C ABI, all 20 episodes as one batch per turning direction -- and next to it the HIP path against the fp64 oracle
on these very action sequences, which is the usual float32-floor agreement."""
import json
import os

import numpy as np
import pytest

import pybullet_replay as pr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def test_hip_path_follows_the_pybullet_record(torch):
    episodes = pr.load()
    hip = pr.HipReplayer(episodes, steps=200)
    s = pr.summarize(episodes, hip, steps=200, windows=(25, 50, 100, 200))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "pybullet_record_hip.json"), "w") as f:
        json.dump(s, f, indent=1)
    assert s["rate_profile_correlation"] > 0.85, s["rate_profile_correlation"]
    for name, ev in s["event_peaks"].items():
        assert ev["record"] == ev["replay"], (name, ev)
        if int(name.split("-")[0]) < 70:
            assert abs(ev["replay_rad_s"] / ev["record_rad_s"] - 1) < 0.15, (name, ev)
    w = s["windows"]
    assert w[25]["rp_rmse"] < 3.2e-3 and w[25]["rp_rmse"] < 0.5 * w[25]["rp_ref_rms"], w[25]
    assert w[50]["rp_rmse"] < 6.0e-3 and w[50]["rp_rmse"] < 0.7 * w[50]["rp_ref_rms"], w[50]
    assert w[25]["rate_rmse"] < 0.6 * w[25]["rate_ref_rms"] and w[25]["reward_rmse"] < 2.0e-3, w[25]


def test_hip_path_and_oracle_agree_on_the_recorded_actions(torch):
    episodes = pr.load()[:6]
    hip = pr.HipReplayer(episodes, steps=100)
    for ep in episodes:
        for d in (pr.CCW, pr.CW):
            a, _, _ = hip(ep["action"], d, 100)
            b, _, _ = pr.replay_oracle(ep["action"], d, 100)
            n = min(len(a), len(b), 51)
            assert np.abs(a[:n, :2] - b[:n, :2]).max() < 2e-4, (d, np.abs(a[:n, :2] - b[:n, :2]).max())     # roll / pitch over 50 steps: float32 floor


def test_hip_path_on_the_standup_record(torch):
    """The second record (25 standup episodes x 400 steps).  With the product's default toe friction the replay tips over after the
    stand-up (the known discrepancy, tests/test_oracle_pybullet_record.py); with `friction_range=(0.25, 0.25)` -- a constructor keyword of
    the product, the per-reset draw of the env_randomizer hook pinned to one value -- nobody falls and the episode returns are PyBullet's."""
    episodes = pr.load_standup()
    shipped = pr.replay_standup_hip(episodes)
    low = pr.replay_standup_hip(episodes, friction_range=(0.25, 0.25))
    rec = np.array([ep["reward"].sum() for ep in episodes])
    ret_low = np.array([r[2].sum() for r in low])
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "pybullet_standup_record_hip.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(dict(record_return_mean=float(rec.mean()), default_friction=dict(fell=int(sum(r[3] is not None for r in shipped)),
                                                                                     fell_at_median=float(np.median([r[3] for r in shipped if r[3] is not None] or [0])),
                                                                                     return_mean=float(np.mean([r[2].sum() for r in shipped]))),
                       friction_0_25=dict(fell=int(sum(r[3] is not None for r in low)), return_mean=float(ret_low.mean()),
                                          pitch_rmse_all=float(np.mean([np.sqrt(((r[0][:400, 1] - ep["observ"][:, 1]) ** 2).mean()) for r, ep in zip(low, episodes)])))), f, indent=1)
    assert all(r[3] is None for r in low) and ret_low.mean() > 0.9 * rec.mean(), (ret_low.mean(), rec.mean())
    assert all(r[3] is not None for r in shipped)          # the discrepancy as it stands: if this starts failing, the xfail on the CPU side has moved too
    k = 30                                                   # out of the crouch: the default friction follows the record's pitch to 3.5e-2 rad
    assert np.mean([np.sqrt(((r[0][:k, 1] - ep["observ"][:k, 1]) ** 2).mean()) for r, ep in zip(shipped, episodes)]) < 0.035

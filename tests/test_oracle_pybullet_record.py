"""The oracle's restated `stepSimulation` against RECORDED PYBULLET ROLLOUTS.

tests/golden/pybullet_turn_ol_rollouts.npz is the episode memory of the reference's shipped turn / open-loop checkpoint: 20 episodes,
8 091 control steps of RexTurnEnv on real PyBullet -- observation, action, reward per step (tests/golden/make_pybullet_golden.py).  It
is the only PyBullet-produced trajectory data the reference holds, and the only pin on the inside of `stepSimulation` this repo has:
everything else about a16 is the oracle against its own kernels.  tests/pybullet_replay.py plays the recorded actions and recovers
what the record does not state (6 substeps per control step, turning direction, start yaw).

What is asserted is what the physics as shipped achieves, with margins -- NOT the 1e-3 rad of north_star: roll / pitch follow the record
to 2.4e-3 rad RMS over the first 25 control steps (35 % of the signal's RMS), 4.6e-3 over 50; every gait event -- the touch-down after
the reset's teleport and the leg switches -- lands on the record's control step with the record's angular-velocity peak to 5 % (first 70 steps)."""
import numpy as np
import pytest

import pybullet_replay as pr


@pytest.fixture(scope="module")
def episodes():
    return pr.load()


def test_record_shape(episodes):
    assert len(episodes) == 20 and sum(e["length"] for e in episodes) == 8091
    for e in episodes:
        assert np.all(e["observ"][0] == 0)                               # the training-time reset read the pose back after the teleport
        assert np.all(np.abs(e["observ"][:, :2]) < 0.2)                   # upright throughout: none of the 20 episodes ends in a fall
        assert 0.02 < e["reward"].max() <= 0.035                          # 0.035 - |x| - |y|, turn_env.py:362-367


@pytest.fixture(scope="module")
def summary(episodes):
    return pr.summarize(episodes, pr.replay_oracle, steps=120, windows=(25, 50, 100))


def test_gait_events_fall_on_the_records_control_steps(summary):
    assert summary["rate_profile_correlation"] > 0.85, summary["rate_profile_correlation"]          # measured 0.92
    for name, ev in summary["event_peaks"].items():
        assert ev["record"] == ev["replay"], (name, ev)                                              # all six, up to control step 105
        if int(name.split("-")[0]) < 70:     # the peak heights while the replay still tracks the record: within 5 % (16 % / 30 % at steps 86 / 105)
            assert abs(ev["replay_rad_s"] / ev["record_rad_s"] - 1) < 0.15, (name, ev)


def test_roll_pitch_follow_the_record(summary):
    w = summary["windows"]
    assert w[25]["rp_rmse"] < 3.2e-3 and w[25]["rp_rmse"] < 0.5 * w[25]["rp_ref_rms"], w[25]      # measured 2.4e-3 of 6.8e-3
    assert w[50]["rp_rmse"] < 6.0e-3 and w[50]["rp_rmse"] < 0.7 * w[50]["rp_ref_rms"], w[50]      # measured 4.6e-3 of 8.4e-3
    assert w[25]["rate_rmse"] < 0.6 * w[25]["rate_ref_rms"], w[25]                                 # world-frame rates after the yaw fit: 0.14 of 0.27
    assert w[25]["reward_rmse"] < 2.0e-3, w[25]                                                    # |x| + |y| drift: 1.2e-3 m
    assert summary["direction_fit_ratio_min"] > 1.05                                               # every episode tells its turning direction


def test_five_substeps_do_not_reproduce_the_record(episodes):
    """The recovered time base: with today's constructor default (action_repeat 5) the events drift off the record."""
    ref, ours, _ = pr.rate_profile(episodes[:8], pr.replay_oracle, action_repeat=5, solver_iterations=60)
    assert np.corrcoef(ref[1:], ours[1:])[0, 1] < 0.5


# ---------------------------------------------------------------- the second record: RexStandupEnv, 25 episodes x 400 steps, no hidden draws
@pytest.fixture(scope="module")
def standup():
    return pr.load_standup()


def test_standup_record_shape(standup):
    assert len(standup) == 25 and all(e["length"] == 400 for e in standup)
    ret = np.array([e["reward"].sum() for e in standup])
    assert ret.min() > 300 and ret.max() < 350                 # on PyBullet every episode stands up and stays up: +304 ... +345
    e, above = pr.standup_position_error(standup[0]["reward"])
    assert 0.17 < e[0] < 0.20 and above[30:60].any() and not above[100:].any() and e[200:].max() < 0.03


def test_standup_rise_follows_the_record(standup):
    """The first 30 control steps -- out of the crouch, through the 0.1 s of the action's influence, up to the overshoot: the rise rate
    and the pitch swing are the record's to 10-30 %; the crouch the reset leaves is not (|x| + |y| + |0.21 - z| = 0.152 against 0.184)."""
    s = pr.summarize_standup(standup[:8], pr.replay_standup_oracle, steps=60)["summary"]
    assert abs(s["rise_mm_per_step_replay"] / s["rise_mm_per_step_record"] - 1) < 0.15, s          # 6.4 against 6.9 mm per control step
    assert s["pitch_rmse_30"] < 0.035 and 0.8 < s["pitch_peak_replay"] / s["pitch_peak_record"] < 1.5, s   # 0.023 rad; peaks 0.154 / 0.119
    assert abs(s["first_above_replay_median"] - s["first_above_record_median"]) <= 6, s            # base passes 0.21 m at step 34 / 38
    assert 0.025 < s["crouch_error_record"] - s["crouch_error_replay"] < 0.04, s                   # the known 3 cm of the crouch


@pytest.mark.xfail(strict=True, reason="KNOWN DISCREPANCY (DESIGN.md section 2): on PyBullet the robot stands after the overshoot in all 25 "
                   "recorded episodes (+333 per episode); on the restated physics it vaults over its front feet and trips is_fallen at "
                   "step ~100.  With the toe friction at 0.2-0.25 instead of 0.5 the record is reproduced (+317); the turn record's drift "
                   "asks for 0.35-0.5.  strict: if a change of the physics makes this pass, the bars of this file move with it.")
def test_standup_outcome_of_the_record(standup):
    s = pr.summarize_standup(standup[:4], pr.replay_standup_oracle, steps=400)["summary"]
    assert s["fell"] == 0 and s["return_replay"] > 300


def test_standup_outcome_with_low_toe_friction(standup):
    """The same replay with the toe friction at 0.25 (a probe of the oracle, not a product setting): nobody falls, the episode return and
    the pitch trace over all 400 steps are the record's -- what the xfail above would have to look like."""
    import ctypes
    import orclib
    try:
        s = pr.summarize_standup(standup[:4], pr.replay_standup_oracle, steps=400, probes=dict(mu=0.25))["summary"]
    finally:                                                         # (the probe is a process-wide static of the oracle: back to the shipped value)
        lib = orclib.Oracle().lib
        lib.orc_set_probe.argtypes = [ctypes.c_char_p, ctypes.c_double]
        lib.orc_set_probe(b"mu", 0.5)
    assert s["fell"] == 0 and s["return_replay"] > 0.9 * s["return_record_400"] and s["pitch_rmse_all"] < 0.04, s

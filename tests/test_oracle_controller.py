"""Pins the oracle's controller half against vectors produced by the reference's own numpy code
(tests/golden/make_golden.py -> controller_golden.json) and the SURVEY.md 8(c) known answers."""
import numpy as np
import pytest

from orclib import Oracle


@pytest.fixture(scope="module")
def o64():
    return Oracle(np.float64)


@pytest.fixture(scope="module")
def o32():
    return Oracle(np.float32)


def test_ik_matches_reference(golden, o64):
    g = golden["ik"]
    ang, tfr = o64.ik_solve(g["orn"], g["pos"], g["frames"])
    np.testing.assert_allclose(ang, np.array(g["angles"]), rtol=0, atol=1e-12)
    np.testing.assert_allclose(tfr, np.array(g["tframes"]), rtol=0, atol=1e-12)


def test_ik_known_answers(o64):
    # SURVEY.md 8(c) IK-1 / IK-2
    d = np.array([[0.115, -0.0925, -0.2], [0.115, 0.0925, -0.2], [-0.115, -0.0925, -0.2], [-0.115, 0.0925, -0.2]])
    ang, tfr = o64.ik_solve([0, 0, 0], [0.01, 0, 0], d)
    np.testing.assert_allclose(ang.reshape(4, 3), np.tile([0, -0.87033234, 1.30787867], (4, 1)), atol=1e-8)
    assert set(np.round(tfr.reshape(4, 3)[:, 0], 9)) == {0.105, -0.125}
    ang, _ = o64.ik_solve([0.1, -0.05, 0.2], [0.01, 0.005, -0.02], d)
    np.testing.assert_allclose(ang.reshape(4, 3)[0], [0.276940086, -1.193986515, 1.533224120], atol=1e-8)
    np.testing.assert_allclose(ang.reshape(4, 3)[3], [0.031124584, -1.138158109, 1.782962307], atol=1e-8)


def test_motor_matches_reference(golden, o64):
    g = golden["motor"]
    act, obs = o64.motor_torque(g["cmd"], g["q"], g["qd"], g["qd_true"], g["kp"], g["kd"])
    np.testing.assert_allclose(act, np.array(g["actual"]).ravel(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(obs, np.array(g["observed"]).ravel(), rtol=0, atol=1e-12)


def test_motor_known_answers(o64):
    act, obs = o64.motor_torque([0.3], [0.0], [0.0], [0.0])
    np.testing.assert_allclose(act, [3.29032258], atol=1e-8)
    np.testing.assert_allclose(obs, [4.92387097], atol=1e-8)


def _run_gait_sequence(o, seq):
    mode = 0 if seq["mode"] == "walk" else 1
    planner = np.zeros((1, 3))
    frames, states = [], []
    for c in seq["calls"]:
        params = [[c["v"], c["angle"], c["w_rot"], c["T"], c["direction"], c["now"]]]
        planner, fr = o.gait_loop(mode, planner, params)
        frames.append(fr[0].copy())
        states.append(planner[0].copy())
    return np.array(frames), np.array(states)


def test_gait_sequences_match_reference(golden, o64):
    for seq in golden["gait"]:
        fr, st = _run_gait_sequence(o64, seq)
        ref_fr = np.array([c["frames"] for c in seq["calls"]])
        ref_st = np.array([[c["phi"], c["last_time"], c["alpha"]] for c in seq["calls"]])
        np.testing.assert_allclose(fr, ref_fr, rtol=0, atol=1e-12)
        np.testing.assert_allclose(st, ref_st, rtol=0, atol=1e-12)


def test_gait_known_answers(golden, o64):
    # SURVEY.md 8(c) GAIT-1..3: planner primed at t=10, then evaluated at phase 0, 0.3, 0.8
    ka = golden["known_answers"]
    planner = np.zeros((1, 3))
    planner, _ = o64.gait_loop(0, planner, [[0.6, 0, 0, 0.65, 1, 10.0]])
    planner, f1 = o64.gait_loop(0, planner, [[0.6, 0, 0, 0.65, 1, 20.0]])
    planner, f2 = o64.gait_loop(0, planner, [[0.6, 0, 0, 0.65, 1, 20.0 + 0.65 * 0.3]])
    planner, f3 = o64.gait_loop(0, planner, [[0.6, 10, 0.5, 0.65, 1, 20.0 + 0.65 * 0.8]])
    np.testing.assert_allclose(f1[0], ka["GAIT-1"], atol=1e-12)
    np.testing.assert_allclose(f2[0], ka["GAIT-2"], atol=1e-12)
    np.testing.assert_allclose(f3[0], ka["GAIT-3"], atol=1e-12)
    np.testing.assert_allclose(planner[0, 2], ka["GAIT-3-alpha"], atol=1e-12)
    np.testing.assert_allclose(f1[0][:3], [0.145, -0.0925, -0.2], atol=1e-9)
    np.testing.assert_allclose(f3[0][:3], [0.117776586, -0.091894355, -0.151854427], atol=1e-8)


def test_f32_build_tracks_f64(golden, o32, o64):
    g = golden["ik"]
    a32, _ = o32.ik_solve(g["orn"], g["pos"], g["frames"])
    a64, _ = o64.ik_solve(g["orn"], g["pos"], g["frames"])
    assert np.abs(a32 - a64).max() < 2e-4  # check_domain/sqrt edges amplify fp32 rounding
    seq = golden["gait"][0]
    f32, _ = _run_gait_sequence(o32, seq)
    f64, _ = _run_gait_sequence(o64, seq)
    # fp32 rounding may pick the other side of a phase threshold (phi <= 0.5, phi >= 0.99) on the
    # few steps that land exactly on it; everywhere else the two precisions agree tightly
    err = np.abs(f32 - f64).max(axis=1)
    assert np.mean(err < 2e-5) > 0.98 and np.median(err) < 1e-6

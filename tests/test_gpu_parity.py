"""GPU parity tests proper: every call goes through the C ABI of librexsim_hip.so (via rex_gym_amd),
and is checked against the CPU oracle / the golden vectors.  Tolerances are written next to each check.
Run on the MI355X box: python -m pytest tests -m gpu"""
import os

import numpy as np
import pytest

import orclib
from helpers import joint_rmse, make_pair, numeric_to_product_state, product_state_to_numeric

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "gpu tests need an MI355X"
    return t


@pytest.fixture(scope="module")
def L():
    from rex_gym_amd import _lib
    return _lib


def _dev(torch, a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")


# ------------------------------------------------------------------ controller half (pinned to the reference)
def test_ik_kernel_vs_reference_golden(torch, L, golden):
    g = golden["ik"]
    n = len(g["orn"])
    orn, pos, fr = _dev(torch, g["orn"]), _dev(torch, g["pos"]), _dev(torch, np.array(g["frames"]).reshape(n, 12))
    out = torch.zeros((n, 12), device="cuda")
    L.check(L.lib().rex_ik_solve(n, orn.data_ptr(), pos.data_ptr(), fr.data_ptr(), out.data_ptr(), None), "rex_ik_solve")
    torch.cuda.synchronize()
    ref = np.array(g["angles"])
    err = np.abs(out.cpu().numpy() - ref)
    # fp32 kernel vs fp64 reference: 1e-5 rad typical; the clamped-domain / sqrt(0) edge cases are
    # ill-conditioned (d acos / dx -> inf), allow 2e-3 there
    assert np.median(err) < 2e-6
    assert np.mean(err < 2e-5) > 0.97
    assert err.max() < 2e-3


def test_motor_kernel_vs_reference_golden(torch, L, golden):
    g = golden["motor"]
    cmd, q, qd, qdt = (_dev(torch, np.array(g[k]).ravel()) for k in ("cmd", "q", "qd", "qd_true"))
    n = cmd.numel()
    act, obs = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    L.check(L.lib().rex_motor_torque(n, cmd.data_ptr(), q.data_ptr(), qd.data_ptr(), qdt.data_ptr(), g["kp"], g["kd"],
                                     act.data_ptr(), obs.data_ptr(), None), "rex_motor_torque")
    torch.cuda.synchronize()
    # tolerance 1e-5 N m, relative to the 3.5 / 5.7 N m full scale
    np.testing.assert_allclose(act.cpu().numpy(), np.array(g["actual"]).ravel(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(obs.cpu().numpy(), np.array(g["observed"]).ravel(), atol=2e-5, rtol=1e-5)


def test_gait_kernel_vs_reference_golden(torch, L, golden):
    """Planner state and clock parameters go in as float64 (rexsim.h, "Clocks"): the 0.99 latch, the wrap at 1 and the
    stance / swing split are then the reference's own decisions on EVERY call -- including the sequence whose period
    clamps to 0.01 s, where every 5 ms call lands exactly on a threshold -- and only float32 trajectory arithmetic is
    left between the kernel and the reference."""
    worst = 0.0
    for seq in golden["gait"]:
        mode = 0 if seq["mode"] == "walk" else 1
        planner = torch.zeros((1, 3), device="cuda", dtype=torch.float64)
        frames = torch.zeros((1, 12), device="cuda")
        for k, c in enumerate(seq["calls"]):
            params = torch.tensor([[c["v"], c["angle"], c["w_rot"], c["T"], c["direction"], c["now"]]], device="cuda", dtype=torch.float64)
            L.check(L.lib().rex_gait_loop(1, mode, planner.data_ptr(), params.data_ptr(), frames.data_ptr(), None), "rex_gait_loop")
            err = np.abs(frames.cpu().numpy()[0] - np.array(c["frames"])).max()
            assert err < 2e-6, (seq["mode"], k, err)
            st = planner.cpu().numpy()[0]
            assert abs(st[0] - c["phi"]) < 1e-12 and abs(st[1] - c["last_time"]) < 1e-12, (seq["mode"], k)   # the double state, exactly
            assert abs(st[2] - c["alpha"]) < 2e-6
            worst = max(worst, err)
    print("largest |frame - reference| over all planner calls: %.2e m" % worst)


# ------------------------------------------------------------------ full step vs oracle
def test_settled_snapshot_matches_oracle(torch):
    env, orc = make_pair("walk", "ik", 4, np.float64, seed=3)
    obs = env.reset().cpu().numpy()
    oobs = orc.reset()
    ps, os_ = product_state_to_numeric(env.state), orc.get_state()
    # 600 substeps of contact dynamics, fp32 kernel vs fp64 oracle: 2e-4 on positions/angles,
    # 2e-3 on the (near zero) velocities
    np.testing.assert_allclose(ps[:7], os_[:7], atol=2e-4)
    np.testing.assert_allclose(ps[orclib.S_Q:orclib.S_Q + 12], os_[orclib.S_Q:orclib.S_Q + 12], atol=2e-4)
    np.testing.assert_allclose(ps[7:13], os_[7:13], atol=2e-3)
    np.testing.assert_allclose(ps[orclib.S_QD:orclib.S_QD + 12], os_[orclib.S_QD:orclib.S_QD + 12], atol=2e-3)
    np.testing.assert_allclose(obs, oobs, atol=2e-3)
    # episode draws are integer/fp32-exact: identical
    np.testing.assert_array_equal(ps[orclib.S_TARGET], os_[orclib.S_TARGET].astype(np.float32))
    np.testing.assert_array_equal(ps[orclib.S_FLAGS], os_[orclib.S_FLAGS])
    np.testing.assert_array_equal(ps[orclib.S_EPISODE], os_[orclib.S_EPISODE])
    env.close()


@pytest.mark.parametrize("task,signal", [("walk", "ik"), ("walk", "ol"), ("gallop", "ol"), ("gallop", "ik"),
                                         ("turn", "ik"), ("turn", "ol"), ("poses", "ik"), ("standup", "ol")])
def test_single_step_parity_from_common_states(torch, task, signal):
    """One env.step() from identical states against the fp64 oracle: isolates per-step error from chaotic divergence.
    Tolerance: 1e-4 rad / 1e-4 m / 2e-2 rad/s over 5-6 substeps x 50-60 PGS iterations.  The controller's discrete
    decisions (phi <= 0.5, phi >= 0.99, ramp and brake windows -- exact ties on a 5 ms control step) are taken in double
    by the kernels as by the reference (rexsim.h, "Clocks"), so the flags words must agree exactly and no env may sit
    on the other side of a threshold."""
    n = 64
    env, orc = make_pair(task, signal, n, np.float64, seed=11)
    env.reset(); orc.reset()
    rng = np.random.RandomState(5)
    lo, hi = np.minimum(env.action_space.low, env.action_space.high), np.maximum(env.action_space.low, env.action_space.high)
    # walk a few steps on the oracle to get diverse, physically consistent states
    for k in range(30):
        a = rng.uniform(lo, hi, (n, env.action_dim)).astype(np.float32)
        orc.step(a)
    for k in range(10):
        st = orc.get_state()
        env.state.copy_(numeric_to_product_state(st, torch, env.state.device))
        a = rng.uniform(lo, hi, (n, env.action_dim)).astype(np.float32)
        obs, rew, done, info = env.step(torch.as_tensor(a, device="cuda"))
        oobs, orew, odone, ocmd = orc.step(a)
        ps, os_ = product_state_to_numeric(env.state), orc.get_state()
        np.testing.assert_allclose(info["action"].cpu().numpy(), ocmd, atol=3e-5)
        np.testing.assert_allclose(ps[orclib.S_Q:orclib.S_Q + 12], os_[orclib.S_Q:orclib.S_Q + 12], atol=1e-4)
        np.testing.assert_allclose(ps[:7], os_[:7], atol=1e-4)
        np.testing.assert_allclose(ps[7:13], os_[7:13], atol=5e-3)
        np.testing.assert_allclose(ps[orclib.S_QD:orclib.S_QD + 12], os_[orclib.S_QD:orclib.S_QD + 12], atol=2e-2)
        np.testing.assert_allclose(rew.cpu().numpy(), orew, atol=1e-4)
        np.testing.assert_array_equal(done.cpu().numpy(), odone)
        np.testing.assert_allclose(obs.cpu().numpy(), oobs, atol=5e-3)
        for w in (orclib.S_FLAGS, orclib.S_LASTT, orclib.S_ENDTIME, orclib.S_STEPS):    # every discrete decision of the step
            np.testing.assert_array_equal(ps[w], os_[w])
    env.close()


def test_walk_ik_trajectory_rmse(torch):
    """BASELINE.json: "joint trajectories within 1e-3 rad RMSE" -- first 200 control steps (1 s) from the settled reset
    state, same actions, HIP path against the fp64 oracle; per env the RMSE over the window and the joints."""
    import parity_window as pw
    env = pw.make_env("walk_ik_4096", n=256, seed=1)
    rec = pw.window("walk_ik_4096", env, steps=200, seed=1)
    print("walk-ik 200-step joint RMSE vs the fp64 oracle: median %.2e p99 %.2e max %.2e rad" % (rec["median_rad"], rec["p99_rad"], rec["max_rad"]))
    assert rec["p99_rad"] <= 1e-3                                        # BASELINE.json's bar, on the 99th percentile env
    assert rec["median_rad"] <= 1e-5 and rec["max_rad"] <= 4e-3, rec     # measured 2.5e-6 / 1.8e-3: a toe touching down a substep apart
    env.close()


# ------------------------------------------------------------------ mark='arm' (18 motors, rex_arm.urdf)
def _arm_words(nm=18):
    q, qd = 13, 13 + nm
    return q, qd, 13 + 2 * nm + 6    # Q, QD, FLAGS


def test_arm_mark_settled_snapshot_and_shapes(torch):
    """mark='arm': 69 state words, 18-wide motor command, gallop observation 4 + 18; the settled reset state
    (arm held at ARM_POSES['rest'] against its +-1.5 rad bounds) matches the fp32 arm oracle."""
    n = 8
    env, orc = make_pair("gallop", "ol", n, np.float32, seed=2, mark="arm")
    assert env.num_motors == 18 and env.state_words == 69 and env.obs_dim == 22
    assert env.observation_space.shape == (22,)
    obs, oobs = env.reset().cpu().numpy(), orc.reset()
    ps, os_ = product_state_to_numeric(env.state), orc.get_state()
    Q, QD, FL = _arm_words()
    np.testing.assert_allclose(ps[Q:Q + 18], os_[Q:Q + 18], atol=2e-4)
    np.testing.assert_allclose(ps[:7], os_[:7], atol=1e-4)
    np.testing.assert_allclose(ps[7:13], os_[7:13], atol=2e-3)
    np.testing.assert_allclose(ps[QD:QD + 18], os_[QD:QD + 18], atol=2e-3)
    np.testing.assert_array_equal(ps[FL:], os_[FL:])
    np.testing.assert_allclose(obs, oobs, atol=2e-3)
    # the three out-of-range rest targets sit on their limits
    assert np.all(np.abs(ps[Q + 12:Q + 14]) < 1.52) and np.all(np.abs(ps[Q + 16]) < 1.52)
    env.close()


@pytest.mark.parametrize("task,signal", [("walk", "ik"), ("gallop", "ol"), ("turn", "ik")])
def test_arm_mark_single_step_parity(torch, task, signal):
    """One env.step() from identical states, mark='arm', against the fp32 arm oracle (same tolerances as the base
    mark: 1e-4 rad / 1e-4 m / 2e-2 rad/s)."""
    n = 32
    env, orc = make_pair(task, signal, n, np.float32, seed=13, mark="arm")
    env.reset(); orc.reset()
    rng = np.random.RandomState(8)
    lo, hi = np.minimum(env.action_space.low, env.action_space.high), np.maximum(env.action_space.low, env.action_space.high)
    Q, QD, FL = _arm_words()
    for k in range(20):
        orc.step(rng.uniform(lo, hi, (n, env.action_dim)))
    for k in range(8):
        st = orc.get_state()
        env.state.copy_(numeric_to_product_state(st, torch, env.state.device))
        a = rng.uniform(lo, hi, (n, env.action_dim)).astype(np.float32)
        obs, rew, done, info = env.step(torch.as_tensor(a, device="cuda"))
        oobs, orew, odone, ocmd = orc.step(a.astype(np.float64))
        ps, os_ = product_state_to_numeric(env.state), orc.get_state()
        np.testing.assert_allclose(info["action"].cpu().numpy(), ocmd, atol=3e-5)
        np.testing.assert_allclose(ps[Q:Q + 18], os_[Q:Q + 18], atol=1e-4)
        np.testing.assert_allclose(ps[:7], os_[:7], atol=1e-4)
        np.testing.assert_allclose(ps[7:13], os_[7:13], atol=5e-3)
        np.testing.assert_allclose(ps[QD:QD + 18], os_[QD:QD + 18], atol=2e-2)
        np.testing.assert_allclose(rew.cpu().numpy(), orew, atol=1e-4)
        np.testing.assert_array_equal(done.cpu().numpy(), odone)
        np.testing.assert_allclose(obs.cpu().numpy(), oobs, atol=5e-3)
    env.close()


def test_arm_mark_trajectory_and_auto_reset(torch):
    """100 lock-step control steps from reset (median joint RMSE < 1e-3 rad over all 18 joints), then a long
    auto-reset rollout at a ragged batch size stays finite and keeps the arm inside its limits."""
    n = 64
    env, orc = make_pair("walk", "ik", n, np.float32, seed=3, mark="arm")
    env.reset(); orc.reset()
    rng = np.random.RandomState(1)
    worst = np.zeros(n)
    for k in range(100):
        a = rng.uniform(-0.4, 0.4, (n, 2)).astype(np.float32)
        env.step(torch.as_tensor(a, device="cuda"))
        orc.step(a)
        worst = np.maximum(worst, joint_rmse(product_state_to_numeric(env.state), orc.get_state()))
    print("arm walk-ik 100-step joint RMSE vs f32 oracle: median %.3e max %.3e" % (np.median(worst), worst.max()))
    assert np.median(worst) < 1e-3
    env.close()
    from rex_gym_amd import RexBatchEnv
    env = RexBatchEnv(4099, task="walk", signal_type="ik", mark="arm", seed=5, auto_reset=True, max_episode_steps=150)
    env.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    ndone = 0
    for k in range(400):
        a = torch.rand((4099, 2), device="cuda", generator=g) * 0.8 - 0.4
        obs, rew, done, _ = env.step(a)
        ndone += int(done.sum())
    assert bool(torch.isfinite(env.state[:67]).all()) and bool(torch.isfinite(obs).all()) and ndone >= 4099
    Q = 13
    assert float(env.state[Q + 12:Q + 18].abs().max()) < 3.2
    env.close()


def test_arm_mark_latency_model_parity(torch):
    """The latency model for mark 'arm' (61-word history records: 18 q, 18 qd, 18 observed torques, quaternion, angular
    velocity; rex.py:717-763): PD latency 1 ms + control latency 5.5 ms, lock step with the fp32 arm oracle through a
    reset by index.  (With 2 ms or more of PD latency the arm links -- 0.2 kg, 5e-5 kg m^2 -- ring against their joint
    limits during the reset motion and the fp32 and fp64 oracles part ways: there is no trajectory to compare.)"""
    n = 16
    env, orc = make_pair("walk", "ik", n, np.float32, seed=3, mark="arm", pd_latency=0.001, control_latency=0.0055)
    assert env.history.shape[0] == 100 * 61
    np.testing.assert_allclose(env.reset().cpu().numpy(), orc.reset(), atol=2e-3)
    rng = np.random.RandomState(1)
    for k in range(40):
        if k == 20:
            idx = np.array([1, 5, 9], np.int32)
            np.testing.assert_allclose(env.reset(idx).cpu().numpy(), orc.reset(idx), atol=2e-3)
        a = rng.uniform(-0.4, 0.4, (n, 2)).astype(np.float32)
        o, r, d, info = env.step(torch.as_tensor(a, device="cuda"))
        oo, orr, od, ocmd = orc.step(a)
        np.testing.assert_allclose(info["action"].cpu().numpy(), ocmd, atol=5e-5)
        np.testing.assert_allclose(o.cpu().numpy()[:, :2], oo[:, :2], atol=5e-3)
        np.testing.assert_allclose(o.cpu().numpy()[:, 2:], oo[:, 2:], atol=0.3)        # angular rates: 1 / dt times the angle error
        np.testing.assert_allclose(r.cpu().numpy(), orr, atol=5e-3)
    ps, os_ = product_state_to_numeric(env.state), orc.get_state()
    err = np.abs(ps[13:31] - os_[13:31])                     # 40 steps of a lightly damped (delayed PD) 19-body chain in fp32
    assert np.median(err) < 2e-3 and err.max() < 5e-2, (np.median(err), err.max())
    env.close()


def test_turn_env_reset_and_goal_logic(torch):
    """RexTurnEnv: yaw draws, teleport to the start heading, goal detection -> hold pose -> done 1 s later
    (turn_env.py:129-160,324-347). Discrete outcomes must match the fp32 oracle exactly."""
    n = 64
    env, orc = make_pair("turn", "ol", n, np.float32, seed=21)
    obs, oobs = env.reset().cpu().numpy(), orc.reset()
    ps, os_ = product_state_to_numeric(env.state), orc.get_state()
    np.testing.assert_array_equal(ps[orclib.S_TARGET], os_[orclib.S_TARGET])
    np.testing.assert_array_equal(ps[orclib.S_AUX], os_[orclib.S_AUX])
    np.testing.assert_allclose(ps[3:7], os_[3:7], atol=1e-6)
    assert np.all(ps[2] == np.float32(0.21))
    np.testing.assert_allclose(obs, oobs, atol=2e-3)
    rng = np.random.RandomState(4)
    ndone = 0
    for k in range(500):
        a = rng.uniform(-0.01, 0.01, (n, 2)).astype(np.float32)
        o, r, d, _ = env.step(torch.as_tensor(a, device="cuda"))
        oo, orr, od, _ = orc.step(a)
        agree = np.mean(d.cpu().numpy() == od)
        assert agree > 0.95, (k, agree)       # chaotic contact dynamics: a few envs may cross a threshold a step apart
        ndone += int(od.sum())
    flags = product_state_to_numeric(env.state)[orclib.S_FLAGS].astype(int)
    assert (flags & 1).sum() > 0              # some envs reached their target heading
    env.close()


def test_random_heightfield_terrain_parity(torch):
    """terrain_type='random' (model/terrain.py:32-54), BASELINE config 4 shape (turn-IK on a heightfield):
    per-terrain settled snapshots and lock-step single steps from common states vs the fp32 oracle."""
    n = 64
    env, orc = make_pair("turn", "ik", n, np.float32, seed=8, terrain_type="random", terrain_pool=8)
    env.reset(); orc.reset()
    ps, os_ = product_state_to_numeric(env.state), orc.get_state()
    np.testing.assert_allclose(ps[:7], os_[:7], atol=2e-4)
    np.testing.assert_allclose(ps[orclib.S_Q:orclib.S_Q + 12], os_[orclib.S_Q:orclib.S_Q + 12], atol=3e-4)
    # different pool entries settle to different joint poses: the terrain is really in the loop
    assert np.ptp(ps[orclib.S_Q + 1]) > 1e-4
    rng = np.random.RandomState(1)
    for k in range(20):
        orc.step(rng.uniform(-0.01, 0.01, (n, 2)))
    worst = 0.0
    for k in range(10):
        st = orc.get_state()
        env.state.copy_(numeric_to_product_state(st, torch, env.state.device))
        a = rng.uniform(-0.01, 0.01, (n, 2)).astype(np.float32)
        env.step(torch.as_tensor(a, device="cuda")); orc.step(a)
        ps, os_ = product_state_to_numeric(env.state), orc.get_state()
        err = np.abs(ps[orclib.S_Q:orclib.S_Q + 12] - os_[orclib.S_Q:orclib.S_Q + 12]).max(axis=0)
        worst = max(worst, np.median(err))
        # a toe end sitting exactly on a cell diagonal / plane-field seam may pick the other facet in fp32:
        # require the bulk of the envs to agree tightly
        assert np.mean(err < 2e-4) > 0.9, np.sort(err)[-8:]
    print("heightfield single-step median joint error %.2e rad" % worst)
    env.close()


def test_latency_model_parity(torch):
    """pd_latency / control_latency (Rex._GetDelayedObservation, rex.py:735-763): history ring + interpolation.
    Lock-step rollout from reset vs the fp32 oracle; also checks that the delayed observation really lags."""
    n = 64
    env, orc = make_pair("walk", "ik", n, np.float32, seed=6, pd_latency=0.0025, control_latency=0.0125)
    obs, oobs = env.reset().cpu().numpy(), orc.reset()
    np.testing.assert_allclose(obs, oobs, atol=2e-3)
    rng = np.random.RandomState(8)
    worst = np.zeros(n)
    # a PD loop fed with 2.5 ms old velocities is lightly damped: rounding differences grow ~10x per 5 control
    # steps once the gait starts, so the lock-step window is the first 25 steps (5e-7 rad agreement there)
    for k in range(25):
        a = rng.uniform(-0.4, 0.4, (n, 2)).astype(np.float32)
        o, r, d, _ = env.step(torch.as_tensor(a, device="cuda"))
        oo, orr, od, _ = orc.step(a)
        ps, os_ = product_state_to_numeric(env.state), orc.get_state()
        worst = np.maximum(worst, joint_rmse(ps, os_))
        np.testing.assert_array_equal(ps[orclib.S_HIST], os_[orclib.S_HIST])
        np.testing.assert_allclose(o.cpu().numpy(), oo, atol=5e-3)
        np.testing.assert_allclose(r.cpu().numpy(), orr, atol=1e-4)
    assert np.median(worst) < 1e-4
    # the returned angular rates are the 12.5 ms old ones, not the current ones
    cur = env.state[orclib.S_ANGVEL:orclib.S_ANGVEL + 2].cpu().numpy().T
    assert np.abs(o.cpu().numpy()[:, 2:4] - cur).max() > 1e-3
    env.close()


def test_joint_limit_rows_parity(torch):
    """URDF joint limits as unilateral solver rows (btMultiBodyJointLimitConstraint): states at / beyond the bounds,
    moving into them, must be stopped identically by kernel and oracle."""
    n = 64
    env, orc = make_pair("walk", "ol", n, np.float32, seed=13)
    env.reset(); orc.reset()
    rng = np.random.RandomState(3)
    st = orc.get_state()
    st[orclib.S_POS + 2] += 0.5                                   # lift the robots: limits act without ground contact too
    st[orclib.S_Q + 0] = rng.uniform(0.9, 1.03, n)                # FL shoulder at its upper bound (1.0)
    st[orclib.S_QD + 0] = rng.uniform(0.0, 5.0, n)
    st[orclib.S_Q + 5] = rng.uniform(2.5, 2.62, n)                # FR foot at its upper bound (2.59)
    st[orclib.S_QD + 5] = rng.uniform(0.0, 8.0, n)
    st[orclib.S_Q + 7] = rng.uniform(-2.2, -2.1, n)               # RL leg at its lower bound (-2.17)
    st[orclib.S_QD + 7] = rng.uniform(-6.0, 0.0, n)
    orc.set_state(st)
    env.state.copy_(numeric_to_product_state(st, torch, env.state.device))
    for k in range(4):
        a = np.zeros((n, 8), np.float32)
        env.step(torch.as_tensor(a, device="cuda")); orc.step(a)
        ps, os_ = product_state_to_numeric(env.state), orc.get_state()
        np.testing.assert_allclose(ps[orclib.S_Q:orclib.S_Q + 12], os_[orclib.S_Q:orclib.S_Q + 12], atol=2e-4)
        np.testing.assert_allclose(ps[orclib.S_QD:orclib.S_QD + 12], os_[orclib.S_QD:orclib.S_QD + 12], atol=5e-2)
    assert ps[orclib.S_Q + 0].max() < 1.0 + 0.03 and ps[orclib.S_Q + 5].max() < 2.59 + 0.03 and ps[orclib.S_Q + 7].min() > -2.17 - 0.03
    env.close()


def test_domain_randomisation_params(torch):
    """Per-env mass scales and foot friction (rex_set_body_params): parity with the oracle, and the knobs act."""
    n = 64
    env, orc = make_pair("walk", "ik", n, np.float32, seed=2)
    rng = np.random.RandomState(9)
    params = np.stack([rng.uniform(0.8, 1.2, n), rng.uniform(0.8, 1.2, n), rng.uniform(0.25, 0.625, n)]).astype(np.float32)
    env.set_body_params(*[torch.as_tensor(p) for p in params])
    orc.set_body_params(params)
    env.reset(); orc.reset()
    for k in range(25):
        orc.step(rng.uniform(-0.4, 0.4, (n, 2)))
    for k in range(5):
        st = orc.get_state()
        env.state.copy_(numeric_to_product_state(st, torch, env.state.device))
        a = rng.uniform(-0.4, 0.4, (n, 2)).astype(np.float32)
        env.step(torch.as_tensor(a, device="cuda")); orc.step(a)
        ps, os_ = product_state_to_numeric(env.state), orc.get_state()
        np.testing.assert_allclose(ps[orclib.S_Q:orclib.S_Q + 12], os_[orclib.S_Q:orclib.S_Q + 12], atol=1e-4)
        np.testing.assert_allclose(ps[:7], os_[:7], atol=1e-4)
    # heavier base -> the stand sags more: compare two envs stepped from the same snapshot
    e2, _ = make_pair("walk", "ik", 2, np.float32, seed=2)
    e2.set_body_params(torch.tensor([0.8, 1.2]), None, None)
    e2.reset()
    for k in range(40):
        e2.step(torch.zeros((2, 2), device="cuda"))
    z = e2.state[2].cpu().numpy()
    assert z[1] < z[0] - 1e-4
    env.close(); e2.close()


def test_folded_wrappers_match_explicit_wrappers(torch):
    """ClipAction + RangeNormalize folded into the launch == the same wrappers applied around the raw env
    (agents/tools/wrappers.py:183-265), incl. gallop's inverted Box."""
    from rex_gym_amd import RexBatchEnv
    for task, sig in (("walk", "ik"), ("gallop", "ol")):
        raw = RexBatchEnv(32, task=task, signal_type=sig, seed=4)
        fold = RexBatchEnv(32, task=task, signal_type=sig, seed=4, range_normalize=True)
        o_raw, o_fold = raw.reset(), fold.reset()
        lo = torch.as_tensor(raw.action_space.low, device="cuda"); hi = torch.as_tensor(raw.action_space.high, device="cuda")
        olo = torch.as_tensor(raw.observation_space.low, device="cuda"); ohi = torch.as_tensor(raw.observation_space.high, device="cuda")
        torch.testing.assert_close(o_fold, 2 * (o_raw - olo) / (ohi - olo) - 1, atol=1e-6, rtol=1e-5)
        assert fold.action_space.contains(np.zeros(raw.action_dim, np.float32)) and np.all(fold.action_space.high == 1)
        g = torch.Generator(device="cuda"); g.manual_seed(0)
        for k in range(8):
            a = torch.rand((32, raw.action_dim), device="cuda", generator=g) * 3 - 1.5      # also outside [-1, 1]
            den = (a.clamp(-1, 1) + 1) / 2 * (hi - lo) + lo
            o1, r1, d1, i1 = raw.step(den)
            o2, r2, d2, i2 = fold.step(a)
            torch.testing.assert_close(i2["action"], i1["action"], atol=2e-6, rtol=0)
            torch.testing.assert_close(r2, r1, atol=1e-5, rtol=0)
            torch.testing.assert_close(o2, 2 * (o1 - olo) / (ohi - olo) - 1, atol=1e-5, rtol=1e-4)
        raw.close(); fold.close()


def test_auto_reset_and_episode_limit(torch):
    n = 128
    env, orc = make_pair("walk", "ik", n, np.float32, seed=9, auto_reset=1, max_episode_steps=20)
    env.reset(); orc.reset()
    rng = np.random.RandomState(2)
    for k in range(45):
        a = rng.uniform(-0.4, 0.4, (n, 2)).astype(np.float32)
        obs, rew, done, _ = env.step(torch.as_tensor(a, device="cuda"))
        oobs, orew, odone, _ = orc.step(a)
        np.testing.assert_array_equal(done.cpu().numpy(), odone)
        if (k + 1) % 20 == 0:
            assert odone.all()
    ps, os_ = product_state_to_numeric(env.state), orc.get_state()
    np.testing.assert_array_equal(ps[orclib.S_EPISODE], os_[orclib.S_EPISODE])
    np.testing.assert_array_equal(ps[orclib.S_STEPS], os_[orclib.S_STEPS])
    np.testing.assert_array_equal(ps[orclib.S_TARGET], os_[orclib.S_TARGET])
    env.close()


def test_reset_indices_and_sharding_invariance(torch):
    """Env g of a 2-shard run equals env g of a single run (RNG keyed by global index)."""
    from rex_gym_amd import RexBatchEnv
    whole = RexBatchEnv(128, seed=5)
    a, b = RexBatchEnv(64, seed=5, env_index_base=0), RexBatchEnv(64, seed=5, env_index_base=64)
    whole.reset(); a.reset(); b.reset()
    act = torch.rand((128, 2), device="cuda") * 0.8 - 0.4
    for k in range(5):
        ow = whole.step(act)[0].clone()
        oa, ob = a.step(act[:64])[0], b.step(act[64:])[0]
        assert torch.equal(ow[:64], oa) and torch.equal(ow[64:], ob)
    assert torch.equal(whole.state[:, :64], a.state) and torch.equal(whole.state[:, 64:], b.state)
    idx = torch.tensor([3, 17, 60], dtype=torch.int32)
    o = a.reset(idx)
    assert o.shape == (3, 4)
    st = product_state_to_numeric(a.state)
    assert (st[orclib.S_EPISODE][[3, 17, 60]] == 2).all() and st[orclib.S_EPISODE][0] == 1
    for e in (whole, a, b):
        e.close()


@pytest.mark.parametrize("n", [1, 3, 63, 65, 1000, 4097, 8195])   # 4 / 8 / 16 envs per wave, each with a ragged last wave
def test_ragged_batch_sizes(torch, n, monkeypatch):
    """Batch sizes that are not multiples of the wave / envs-per-wave: tail lanes must not corrupt neighbours.  The
    reference batch is larger and runs the SAME kernel variant (the envs-per-wave of the small batch is forced onto it),
    so every env must come out bit-identical."""
    from rex_gym_amd import RexBatchEnv
    nref = n + 133
    env = RexBatchEnv(n, seed=2, auto_reset=True, max_episode_steps=7)
    epw = env._L.rex_envs_per_wave(env._h)
    monkeypatch.setenv("REX_ENVS_PER_WAVE", str(epw))
    ref = RexBatchEnv(nref, seed=2, auto_reset=True, max_episode_steps=7)
    assert ref._L.rex_envs_per_wave(ref._h) == epw
    o, o_ref = env.reset(), ref.reset()
    assert o.shape == (n, 4) and torch.equal(o, o_ref[:n])
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    for k in range(9):
        a = torch.rand((nref, 2), device="cuda", generator=g) * 0.8 - 0.4
        on, rn, dn, _ = env.step(a[:n].contiguous())
        orf, rr, dr, _ = ref.step(a)
        assert torch.equal(on, orf[:n]) and torch.equal(rn, rr[:n]) and torch.equal(dn, dr[:n])
    assert torch.equal(env.state, ref.state[:, :n])
    assert torch.isfinite(env.state[:37]).all()
    env.close(); ref.close()


def test_reset_with_empty_and_duplicate_indices(torch):
    from rex_gym_amd import RexBatchEnv
    env = RexBatchEnv(16, seed=1)
    env.reset()
    assert env.reset(torch.zeros(0, dtype=torch.int32)).shape == (0, 4)
    before = product_state_to_numeric(env.state)[orclib.S_EPISODE].copy()
    o = env.reset([5, 5, 9])
    after = product_state_to_numeric(env.state)[orclib.S_EPISODE]
    assert o.shape == (3, 4) and after[9] == before[9] + 1 and after[5] >= before[5] + 1 and after[0] == before[0]
    env.close()


def test_gym_surface_single_env(torch):
    from rex_gym_amd.envs.gym import RexWalkEnv
    env = RexWalkEnv(render=False, signal_type="ik", terrain_type="plane", mark="base")
    obs = env.reset()
    assert obs.shape == (4,) and env.observation_space.contains(obs.astype(np.float32))
    o, r, d, info = env.step(env.action_space.sample())
    assert o.shape == (4,) and isinstance(r, float) and isinstance(d, bool) and info["action"].shape == (12,)
    env.close()


def test_every_reference_env_class_has_its_single_env_counterpart(torch):
    """The five task envs of the reference (envs/gym/*.py) with their default constructors: spaces, one reset, a few
    steps, the reference's return types; mark 'arm' widens info['action'] to 18."""
    from rex_gym_amd.envs import gym as g
    cases = [(g.RexWalkEnv, {}, 2, 4), (g.RexWalkEnv, {"signal_type": "ol"}, 8, 4), (g.RexReactiveEnv, {}, 2, 16),
             (g.RexReactiveEnv, {"signal_type": "ol"}, 4, 16), (g.RexTurnEnv, {}, 2, 4), (g.RexPosesEnv, {"base_y": 0.05}, 1, 4),
             (g.RexStandupEnv, {}, 1, 4), (g.RexWalkEnv, {"mark": "arm"}, 2, 4), (g.RexTurnEnv, {"terrain_type": "random"}, 2, 4),
             (g.RexReactiveEnv, {"signal_type": "ol", "use_angle_in_observation": False}, 4, 4)]    # gallop_env.py:56,344-356
    for cls, kw, adim, odim in cases:
        env = cls(render=False, **kw)
        assert env.action_space.shape == (adim,) and env.observation_space.shape == (odim,), (cls.__name__, kw)
        obs = env.reset()
        assert obs.shape == (odim,) and np.isfinite(obs).all()
        for _ in range(3):
            o, r, d, info = env.step(np.zeros(adim, np.float32))
        assert o.shape == (odim,) and isinstance(r, float) and isinstance(d, bool)
        assert info["action"].shape == ((18,) if kw.get("mark") == "arm" else (12,))
        env.close()


def test_gallop_observation_without_motor_angles(torch):
    """RexReactiveEnv(use_angle_in_observation=False) (envs/gym/gallop_env.py:56,93,344-356,374-377): the observation is roll, pitch
    and their rates alone.  Same seed and actions as the default env: the four words are the default observation's first four bit
    for bit, the physics is untouched, the Box has four bounds; the oracle built the same way agrees; with the wrappers folded the
    fused actor takes a 4-wide input."""
    from rex_gym_amd import RexBatchEnv
    n = 300
    full = RexBatchEnv(n, task="gallop", signal_type="ol", seed=6, auto_reset=True, max_episode_steps=30, check_actions=False)
    bare = RexBatchEnv(n, task="gallop", signal_type="ol", seed=6, auto_reset=True, max_episode_steps=30, check_actions=False, use_angle_in_observation=False)
    assert full.obs_dim == 16 and bare.obs_dim == 4 and bare.observation_space.shape == (4,) and bare.action_space.shape == (4,)
    cfg = orclib.RexConfig.from_buffer_copy(bytes(bare.config))
    orc = orclib.OracleEnv(cfg, np.float32)
    assert orc.obs_dim == 4
    o_full, o_bare = full.reset(), bare.reset()
    assert torch.equal(o_full[:, :4], o_bare)
    np.testing.assert_allclose(o_bare.cpu().numpy(), orc.reset(), atol=2e-3)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    for k in range(45):
        a = torch.rand((n, 4), device="cuda", generator=g) * 0.6 - 0.3
        of, rf, df, _ = full.step(a)
        ob, rb, db, _ = bare.step(a)
        assert ob.shape == (n, 4) and torch.equal(of[:, :4], ob) and torch.equal(rf, rb) and torch.equal(df, db), k
        if k < 5:
            oo, orr, od, _ = orc.step(a.cpu().numpy())
            np.testing.assert_allclose(ob.cpu().numpy(), oo, atol=5e-3); np.testing.assert_array_equal(db.cpu().numpy(), od)
    assert torch.equal(full.state, bare.state)
    full.close(); bare.close()
    with pytest.raises(ValueError):
        RexBatchEnv(4, task="walk", use_angle_in_observation=False)


def test_step_before_reset_raises(torch):
    from rex_gym_amd import RexBatchEnv
    env = RexBatchEnv(8)
    with pytest.raises(RuntimeError):
        env.step(torch.zeros((8, 2), device="cuda"))
    with pytest.raises(ValueError):
        env.reset(); env.step(torch.zeros((8, 3), device="cuda"))
    env.close()


@pytest.mark.parametrize("n,epw", [(96, 4), (5000, 8), (9000, 16)])
def test_mixed_task_batch_matches_single_task_batches_and_the_oracle(torch, n, epw):
    """BASELINE configs[4] shape in ONE launch: arm mark, every env's task drawn per env from {walk, gallop, turn}-IK
    (REX_TASK_MIXED), mass and friction drawn per reset inside the launch.  (a) Env g of the mixed batch equals, to fp32
    round-off, env g of a single-task batch of its task with the same seed and randomisation ranges -- observations, rewards,
    done flags, motor commands, through falls, in-launch resets and the episode cap; (b) the mixed batch follows the
    fp32 oracle's mixed batch in lock step (same task draw, same per-reset draws)."""
    from rex_gym_amd import RexBatchEnv, RexMixedBatchEnv, _lib
    kw = dict(mark="arm", seed=4, auto_reset=True, max_episode_steps=60, mass_scale_range=(0.8, 1.2), friction_range=(0.25, 0.625))
    mix = RexMixedBatchEnv(n, **kw)
    assert mix._L.rex_envs_per_wave(mix._h) == epw
    assert (mix.action_dim, mix.obs_dim, mix.num_motors) == (2, 22, 18)
    ids = mix.task_ids().cpu().numpy()
    assert set(ids) == {0, 1, 2} and min(np.bincount(ids)) > n // 6            # drawn per env, about a third each
    obs0 = mix.reset().cpu().numpy()
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    steps = 80 if n < 1000 else 14
    acts = [torch.rand((n, 2), device="cuda", generator=g) * 0.02 - 0.01 for _ in range(steps)]
    outs = []
    for a in acts:
        o, r, d, info = mix.step(a)
        outs.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), d.cpu().numpy().copy(), info["action"].cpu().numpy().copy()))
    assert np.isfinite(outs[-1][0]).all()
    if n < 1000:
        assert sum(int(x[2].sum()) for x in outs) >= n          # every env ended an episode (cap 60)
    # (a) single-task batches of the same size: env g has the same global index, hence the same draws.  The mixed and the
    #     single-task kernels are different template instantiations (the compiler contracts their FMAs differently), so
    #     "identical" is to fp32 round-off: 1e-6 on the first steps, and -- contact dynamics amplify the last bit -- an env
    #     whose fall lands one step apart leaves the comparison
    for tid, name in ((0, "walk"), (1, "gallop"), (2, "turn")):
        sel = ids == tid
        ref = RexBatchEnv(n, task=name, signal_type="ik", **kw)
        ro = ref.reset().cpu().numpy()
        np.testing.assert_array_equal(ro[sel], obs0[sel][:, :ref.obs_dim])
        assert not obs0[sel][:, ref.obs_dim:].any()
        agree = sel.copy()
        for k, a in enumerate(acts):
            o, r, d, info = ref.step(a)
            agree &= d.cpu().numpy() == outs[k][2]
            oa, ob = o.cpu().numpy()[agree], outs[k][0][agree][:, :ref.obs_dim]
            tol = 5e-4 if k < 3 else 5e-3     # angles; rates (columns 2, 3) are 1 / dt looser.  The arm joints sit ON their
            # limits: a limit row that switches on one substep apart in the two kernels is a jump for that env, so the
            # bound is on 99.5 % of the envs, not on every one
            def mostly(x, y, atol):
                return x.size == 0 or (np.abs(x - y) <= atol).mean() >= 0.995
            assert mostly(oa[:, :2], ob[:, :2], tol), f"{name} step {k}"
            assert mostly(oa[:, 2:4], ob[:, 2:4], 100 * tol), f"{name} step {k}"
            assert mostly(oa[:, 4:], ob[:, 4:], tol), f"{name} step {k}"
            assert mostly(r.cpu().numpy()[agree], outs[k][1][agree], max(tol, 1e-4)), f"{name} step {k}"
            bad = np.abs(info["action"].cpu().numpy() - outs[k][3]).max(1) > 1e-5     # a goal / brake threshold crossed one step apart
            agree &= ~bad
            assert not outs[k][0][sel][:, ref.obs_dim:].any()
        assert agree.sum() >= 0.9 * sel.sum()
        ref.close()
    mix.close()
    # (b) the oracle's mixed batch
    if n < 1000:
        mix = RexMixedBatchEnv(n, **kw)
        cfg = orclib.default_config("mixed", "ik", n, seed=4, auto_reset=1, max_episode_steps=60, mark=1, task_mix=0b111,
                                    action_repeat=6, solver_iterations=60, mass_scale_lo=0.8, mass_scale_hi=1.2,
                                    friction_lo=0.25, friction_hi=0.625)
        orc = orclib.OracleEnv(cfg, np.float32, "arm")
        np.testing.assert_allclose(mix.reset().cpu().numpy(), orc.reset(), atol=2e-3)
        agree = np.ones(n, bool)
        for k, a in enumerate(acts[:30]):
            o, r, d, info = mix.step(a)
            oo, orr, od, ocmd = orc.step(a.cpu().numpy())
            d = d.cpu().numpy().astype(bool)
            agree &= d == od                                     # an env whose fall lands one step apart leaves the comparison
            agree &= np.abs(info["action"].cpu().numpy() - ocmd).max(1) < 5e-5
            assert (np.abs(o.cpu().numpy()[agree] - oo[agree]) <= 2e-2).mean() >= 0.995
            assert (np.abs(r.cpu().numpy()[agree] - orr[agree]) <= 5e-3).mean() >= 0.995
        assert agree.mean() > 0.9
        mix.close()


@pytest.mark.parametrize("task,signal,n", [("walk", "ik", 4096), ("gallop", "ol", 8192), ("turn", "ik", 4096)])
def test_kernel_variants_agree_at_benchmark_sizes(torch, task, signal, n, monkeypatch):
    """BASELINE.json's per-GPU sizes (configs[1], [2], [3]).  The library has three implementations of a substep: one
    env per lane with velocity-form Gauss-Seidel (64 envs per wave), and lane groups of 4 / 8 lanes per env with the
    impulse-space solver (16 / 8 / 4 envs per wave).  From identical states and actions they must give the same step
    to fp32 round-off (1e-4 rad, 1e-4 m, 2e-2 rad/s; identical discrete outcomes), a rerun must be bit-identical, and
    unit quaternions / finite state are size-independent invariants of every env of the batch."""
    from rex_gym_amd import RexBatchEnv
    kw = dict(task=task, signal_type=signal, seed=17, auto_reset=True, max_episode_steps=0,
              terrain_type="random" if task == "turn" else "plane")
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    finals = {}
    for epw in (64, 16, 8, 4, 4):
        monkeypatch.setenv("REX_ENVS_PER_WAVE", str(epw))
        env = RexBatchEnv(n, **kw)
        lo = torch.as_tensor(np.minimum(env.action_space.low, env.action_space.high), device="cuda")
        hi = torch.as_tensor(np.maximum(env.action_space.low, env.action_space.high), device="cuda")
        if "acts" not in finals:
            finals["acts"] = [torch.rand((n, env.action_dim), device="cuda", generator=g) * (hi - lo) + lo for _ in range(12)]
        env.reset()
        # 10 common steps on this variant, then restart every variant from the 64-lane variant's state for 2 more
        for a in finals["acts"][:10]:
            env.step(a)
        if epw == 64:
            finals["state"] = env.state.clone()
        env.state.copy_(finals["state"])
        outs = [tuple(t.clone() for t in env.step(a)[:3]) for a in finals["acts"][10:]]
        st = product_state_to_numeric(env.state)
        assert np.isfinite(st[:37]).all()
        np.testing.assert_allclose(np.linalg.norm(st[3:7], axis=0), 1.0, atol=1e-5)
        key = (epw, "rerun") if (epw, "first") in finals else (epw, "first")
        finals[key] = (st, outs)
        env.close()
    ref, routs = finals[(64, "first")]
    for epw in (16, 8, 4):
        st, outs = finals[(epw, "first")]
        # contact activation is discontinuous (breaking distance, heightfield triangle edges): a handful of envs in
        # tens of thousands may switch a contact one substep apart; everything else agrees to round-off
        for rows, tol, cap in ((slice(13, 25), 1e-4, 5e-3), (slice(0, 7), 1e-4, 5e-3), (slice(25, 37), 2e-2, 5.0)):
            err = np.abs(st[rows] - ref[rows])
            assert (err <= tol).mean() > 0.999 and err.max() < cap, (epw, rows, err.max())
        np.testing.assert_array_equal(st[43:47], ref[43:47])                       # flags, steps, episode, motor enable
        for (o, r, d), (ro, rr, rd) in zip(outs, routs):
            assert ((o - ro).abs() <= 5e-3).float().mean() > 0.999
            assert ((r - rr).abs() <= 1e-4).float().mean() > 0.999
            assert (d == rd).float().mean() > 0.999
    np.testing.assert_array_equal(finals[(4, "first")][0], finals[(4, "rerun")][0])   # determinism


def test_standup_env_crouch_and_rise(torch):
    """RexStandupEnv (standup_env.py:108-166): reset ends crouched on the foot joint limits (2.59 rad), the first
    steps push the base up towards 0.21 m; state, reward and done follow the fp32 oracle in lock step."""
    n = 32
    env, orc = make_pair("standup", "ol", n, np.float32, seed=6)
    obs, oobs = env.reset().cpu().numpy(), orc.reset()
    ps, os_ = product_state_to_numeric(env.state), orc.get_state()
    np.testing.assert_allclose(ps[:7], os_[:7], atol=2e-4)
    np.testing.assert_allclose(ps[13:25], os_[13:25], atol=5e-4)
    assert np.all(np.abs(ps[2] - 0.066) < 0.01) and np.all(np.abs(ps[[15, 18, 21, 24]] - 2.59) < 0.01)
    np.testing.assert_allclose(obs, oobs, atol=2e-3)
    rng = np.random.RandomState(3)
    zmax = np.zeros(n)
    for k in range(40):
        a = rng.uniform(-0.1, 0.1, (n, 1)).astype(np.float32)
        o, r, d, _ = env.step(torch.as_tensor(a, device="cuda"))
        oo, orr, od, _ = orc.step(a)
        ps, os_ = product_state_to_numeric(env.state), orc.get_state()
        if k < 20:   # lock step while the trajectories are still within round-off of each other
            np.testing.assert_allclose(ps[:3], os_[:3], atol=1e-3)
            # the reward jumps by 1 where the distance to (0, 0, 0.21) crosses 0.1 and again at z = 0.21: an env that
            # sits on a threshold within round-off may land on the other side
            assert (np.abs(r.cpu().numpy() - orr) <= 5e-3).mean() > 0.9
        np.testing.assert_array_equal(d.cpu().numpy(), od)
        zmax = np.maximum(zmax, ps[2])
    assert np.all(zmax > 0.15)
    env.close()


def test_ppo_learner_runs_on_the_batch_env_tensors(torch):
    """The PPO learner takes RexBatchEnv's device tensors as they are: a few hundred control steps of
    walk-IK with short episodes fill its memory, it trains, the policy changes and everything stays finite."""
    from rex_gym_amd import RexBatchEnv
    from rex_gym_amd.agents import PPOAgent, PPOConfig, train
    n = 256
    env = RexBatchEnv(n, task="walk", signal_type="ik", seed=5, max_episode_steps=40, check_actions=False)   # a raw Gaussian policy on the raw env: no ClipAction in front
    cfg = PPOConfig(update_every=n, update_epochs_policy=5, update_epochs_value=5, max_length=40)
    agent = PPOAgent(n, env.obs_dim, env.action_dim, cfg, device="cuda", seed=1)
    before = [p.detach().clone() for p in agent.net.parameters()]
    score, length = train(env, agent, 130)
    assert agent.updates >= 2 and np.isfinite(score) and 1 <= length <= 40
    assert all(np.isfinite(v) for s in agent.log for v in s.values())
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, agent.net.parameters()))
    assert int(agent.observ_filter.count) == 130 * n
    env.close()


def test_bench_prints_one_json_line_with_the_contract_fields(torch):
    """bench.py --gpus 1 with a short run: ONE JSON line carrying the driver's contract fields, the roofline object
    and (bounded) the CPU baseline of the oracle."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "60", "--warmup", "20",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["warmup"] == 20 and d["unit"] == "env-steps/s" and d["finite"]
    assert d["scaling"] == "weak" and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(d["value"] - 4096 * 60 / (d["ms_per_step"] * 1e-3 * 60)) / d["value"] < 1e-9
    assert 0.05 < r["kernel_ms"] < 5.0 and r["kernel_ms_min"] <= r["kernel_ms"]
    assert r["kernel_ms_over"].startswith("60 of the 60 timed launches")         # the kernel time is that of the timed launches
    assert abs(r["kernel_ms"] - d["ms_per_step"]) / d["ms_per_step"] < 0.2
    # the secondary measurements: full blocks at the top level, compact flat copies inside `config` / `roofline` (what a record keeper
    # that reduces unknown top-level keys to their names still carries)
    cl, cfg = d["closed_loop"], d["config"]
    assert cl["finite"] and cl["torch_policy_per_step"]["value"] > 0 and cl["fused_segment_100"]["launches_per_step"] == 0.01
    assert cl["fused_segment_25"]["value"] > cl["torch_policy_per_step"]["value"]           # a segment per launch beats a launch + ~20 PyTorch ops per step
    for k in ("open_loop_segment_25_env_steps_per_s", "open_loop_segment_100_env_steps_per_s", "walking_gait_clock_1.5_env_steps_per_s",
              "closed_loop_torch_policy_per_step_env_steps_per_s", "closed_loop_fused_per_step_env_steps_per_s",
              "closed_loop_fused_segment_25_env_steps_per_s", "closed_loop_fused_segment_100_env_steps_per_s"):
        assert isinstance(cfg[k], float) and cfg[k] > 0, k
    # (--no-cpu-baseline also skips the oracle window behind roofline.joint_rmse_*: covered by the default run, profiles/r06z_bench.json)
    assert len(r["csrc_sha16"]) == 16 and "traffic_library_commit" in r and "traffic_is_of_this_library" in r


def test_bench_two_ranks_gather_rollout_segments(torch):
    """The multi-GPU launch path of bench.py on a one-GPU box: two ranks (gloo, sharing the GPU) run a strong-scaling shard of
    BASELINE configs[2] each and all-gather their rollout segments every 25 steps (the design's one collective);
    the line reports the hand-off's bytes and time next to the throughput with and without it."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the plain spelling the driver uses: bench.py starts its own two ranks (torch.distributed.run, rendezvous on 127.0.0.1)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "50", "--warmup", "25",
                          "--config", "3", "--envs-per-gpu", "1024", "--backend", "gloo", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["envs_total"] == 2048 and d["finite"]
    assert d["n_ranks_seen"] == 2 and d["backend"] == "gloo"                     # what the process group itself reports
    g = d["rollout_gather"]
    # gallop-OL: obs 16 x 4 + action 4 x 4 + reward 4 + done 1 = 85 B per env-step
    assert g["every_steps"] == 25 and g["bytes_per_env_step"] == 85 and g["segment_bytes_per_rank"] == 85 * 25 * 1024
    assert g["gather_ms_blocking"] > 0 and g["value_without_gather"] > 0


def test_bench_two_ranks_report_the_north_star_workload_and_segment_launches(torch):
    """`python bench.py --gpus 2` on the default workload (gloo on the one-GPU box): `value` stays the weak-scaling line of configs[1]
    (4 096 envs per rank), and the line carries north_star's own multi-GPU workload -- walk-IK at 65 536 / 8 envs per rank, with and
    without the learner hand-off -- and the same steps launched once per rollout segment (rex_step_segment) next to it."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "50", "--warmup", "10", "--backend", "gloo",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["config"]["envs_total"] == 2 * 4096 and d["scaling"] == "weak" and d["finite"]
    assert d["config"]["baseline_config"] == 2 and d["config"]["launches"].startswith("one launch per env.step()")
    ns = d["north_star_workload"]
    assert ns["envs_per_gpu"] == 65536 // 8 and ns["envs_total"] == 2 * 8192 and ns["finite"]
    assert ns["value"] > 0 and ns["value_without_gather"] >= 0.9 * ns["value"]
    sg = d["segment_launch"]
    assert sg["steps_per_launch"] == 25 and sg["value"] > 0 and sg["longer_segments"]["steps_per_launch"] == 100
    # ... and the closed-loop rollout (policy in the loop) over both ranks' shards, its compact copies in `config`
    cl = d["closed_loop"]
    assert cl["envs_total"] == 2 * 4096 and cl["finite"] and all(cl[k]["value"] > 0 for k in ("torch_policy_per_step", "fused_per_step", "fused_segment_25", "fused_segment_100"))
    assert d["config"]["closed_loop_fused_segment_100_env_steps_per_s"] == cl["fused_segment_100"]["value"]


def test_mark_arm_window_with_quiet_arm_rows(torch, tmp_path):
    """BASELINE configs[4] with a parity window that says something.  In the product the reference's ARM_POSES['rest'] commands three arm
    joints 0.1 rad beyond their bounds (model/rex_constants.py:3-8, rex_arm.urdf:610-791): their limit rows switch with the last bit of
    the joint angle, no env keeps the oracle's event sequence for more than ~8 control steps, and the 200-step window of `mixed_arm_2048`
    compares chaos.  Here the same workload -- 2 048 mark-arm envs, mixed tasks, per-reset mass / friction draws -- runs on DIAGNOSTIC
    builds of the HIP library and of the oracle (rex_gym_amd/librexsim_hip_diag.so, oracle/_build/librex_oracle_arm_diag_*.so: one
    compile-time define that puts the three targets 0.3 rad INSIDE the bounds; test artefacts, never loaded by the product), so
    the arm's rows stay quiet: >= 90 % of the envs must keep the fp64 oracle's event sequence over the whole window and the batch p99 of
    the joint RMSE must meet north_star's 1e-3 rad -- if it does not, that is a kernel finding, not chaos."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from rex_gym_amd import build as hb
    assert os.path.exists(hb.DIAG_LIB_PATH), "build the diagnostic twin first: python -m rex_gym_amd.build --diag (__graft_entry__.build() does)"
    out_path = str(tmp_path / "diag.json")
    env = dict(os.environ, REX_LIB_PATH=hb.DIAG_LIB_PATH, REX_ORACLE_DIAG="1")
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "diag_arm_window.py"), out_path], capture_output=True, text=True, timeout=900,
                         cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.load(open(out_path))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "parity_mixed_arm_quiet_rows.json"), "w") as f:
        json.dump(rec, f, indent=1)
    ev, fl = rec["events"], rec["float32_floor"]
    print("mixed_arm_2048, arm rows quiet: median %.2e p99 %.2e max %.2e rad; envs with the oracle's events %.3f (fp32 oracle %.3f); floor p99 %.2e"
          % (rec["median_rad"], rec["p99_rad"], rec["max_rad"], ev["share_same_event_sequence"], fl["share_same_event_sequence"], fl["p99_rad"]))
    assert rec["trace_pass_bit_identical"]
    assert ev["share_same_event_sequence"] >= 0.9, ev["share_same_event_sequence"]
    assert rec["p99_rad"] <= 1e-3 and rec["median_rad"] <= 1e-4, (rec["median_rad"], rec["p99_rad"])


@pytest.mark.parametrize("case", ["walk_ik", "mixed_arm", "walk_ik_policy"])
def test_hip_shards_on_two_ranks_reproduce_the_single_process_batch(torch, case, tmp_path):
    """"seeds = base_seed (+) global_env_index so results are invariant to G" -- on the HIP path, across REAL ranks: two
    processes (torch.distributed.run, gloo, sharing the one GPU) each step RexBatchEnv(n / 2, env_index_base = rank n / 2) through a
    50-step segment with in-launch resets (episode cap 17) and all-gather it (sharding.gather_rollout); the gathered observation /
    reward / done / action blocks, the reset observations and the final state blocks equal a single-process RexBatchEnv(n) run BIT
    FOR BIT.  walk-IK, the mixed-task mark-arm batch with per-reset mass / friction draws (BASELINE configs[4]), and a closed-loop
    segment (the fused actor: its Gaussian samples are keyed by the global env index too).  tests/hip_shard_worker.py."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_path = str(tmp_path / "res.json")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "tests", "hip_shard_worker.py"), case, out_path],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.load(open(out_path))
    assert res["ranks"] == 2 and res["resets_in_segment"] >= 2 * 2048
    assert all(res[k] for k in ("obs0", "obs", "reward", "done", "action", "state")), res


@pytest.mark.parametrize("task,signal,mark,terrain", [
    ("walk", "ik", "base", "random"), ("gallop", "ik", "base", "random"), ("poses", "ik", "base", "plane"),
    ("standup", "ol", "base", "plane"), ("turn", "ol", "arm", "random"), ("gallop", "ol", "arm", "plane"),
    ("standup", "ol", "arm", "plane"), ("walk", "ol", "base", "plane")])
def test_single_step_parity_along_long_oracle_rollouts(torch, task, signal, mark, terrain):
    """Combinations the other tests do not pair (terrain x mark x task), followed deep into their episodes -- falls,
    joint limits, robots lying on the ground, auto-resets: the oracle runs 120 control steps and every third step the
    HIP step is taken from the oracle's current state and compared (same tolerances as the single-step tests; the
    quantile form allows for contacts that switch one substep apart)."""
    n = 48
    env, orc = make_pair(task, signal, n, np.float32, seed=31, mark=mark, terrain_type=terrain, terrain_pool=8,
                         auto_reset=True, max_episode_steps=90, check_actions=False)
    env.reset(); orc.reset()
    rng = np.random.RandomState(12)
    lo, hi = np.minimum(env.action_space.low, env.action_space.high), np.maximum(env.action_space.low, env.action_space.high)
    nm = env.num_motors
    Q, QD, FL = 13, 13 + nm, 13 + 2 * nm + 6
    bad = 0
    for k in range(120):
        a = (rng.uniform(lo, hi, (n, env.action_dim)) * 3.0).clip(lo * 3, hi * 3).astype(np.float32)   # also outside the Box
        if k % 3 == 0:
            env.state.copy_(numeric_to_product_state(orc.get_state(), torch, env.state.device))
            obs, rew, done, info = env.step(torch.as_tensor(a, device="cuda"))
            oobs, orew, odone, ocmd = orc.step(a.astype(np.float64))
            ps, os_ = product_state_to_numeric(env.state), orc.get_state()
            same = done.cpu().numpy() == odone
            assert same.mean() >= 0.95, (k, same.mean())
            live = same & ~odone                       # compare envs that did not just reset on one side only
            for rows, tol in ((slice(Q, Q + nm), 2e-4), (slice(0, 7), 2e-4), (slice(QD, QD + nm), 5e-2)):
                err = np.abs(ps[rows][:, live] - os_[rows][:, live])
                bad += int((err > tol).any(axis=0).sum())
            np.testing.assert_array_equal(ps[FL + 1:FL + 3][:, same], os_[FL + 1:FL + 3][:, same])   # steps, episode
        else:
            orc.step(a.astype(np.float64))
    assert bad <= 0.02 * n * 40, bad                   # at most 2 % of the (env, step) samples off by more than round-off
    env.close()


@pytest.mark.parametrize("task,signal", [("walk", "ol"), ("gallop", "ol"), ("walk", "ik"), ("gallop", "ik")])
def test_goal_brake_and_hold_sequence_on_the_gpu(torch, task, signal):
    """Goal reached -> brake ramp -> hold pose (walk_env.py:207-324, gallop_env.py:212-313), the part of the command
    logic a short rollout never reaches: the base is put past the target, then every step is taken from the oracle's
    state on both sides and commands and flags are compared (the oracle itself is pinned to the reference's methods for
    exactly these sequences, tests/test_oracle_env_commands.py -- including the `coeff is 0.0` identity test of the
    open-loop branches)."""
    n = 16
    env, orc = make_pair(task, signal, n, np.float32, seed=3, target_position=0.375, backwards=False)
    env.reset(); orc.reset()
    rng = np.random.RandomState(2)
    lo, hi = np.minimum(env.action_space.low, env.action_space.high), np.maximum(env.action_space.low, env.action_space.high)
    seen = set()
    for k in range(330):
        st = orc.get_state()
        if k >= 4:
            st[0] = -0.45                                # |x| past the target (0.375, stop space 0.15 for walk)
            st[1] = 0.0; st[2] = 0.19; st[3:7] = np.array([0, 0, 0, 1.0])[:, None]; st[7:13] = 0.0   # and upright, so that
            st[orclib.S_FLAGS] = st[orclib.S_FLAGS].astype(np.int64) & ~16                          # nobody is done
            orc.set_state(st)
        env.state.copy_(numeric_to_product_state(st, torch, env.state.device))
        a = rng.uniform(lo, hi, (n, env.action_dim)).astype(np.float32)
        _, _, _, info = env.step(torch.as_tensor(a, device="cuda"))
        _, _, _, ocmd = orc.step(a.astype(np.float64))
        ps, os_ = product_state_to_numeric(env.state), orc.get_state()
        np.testing.assert_allclose(info["action"].cpu().numpy(), ocmd, atol=3e-5)
        np.testing.assert_array_equal(ps[orclib.S_FLAGS].astype(int) & 7, os_[orclib.S_FLAGS].astype(int) & 7)
        np.testing.assert_allclose(ps[orclib.S_ENDTIME], os_[orclib.S_ENDTIME], atol=1e-6)
        seen.update((os_[orclib.S_FLAGS].astype(int) & 7).tolist())
    assert 3 in seen and (7 in seen or (task, signal) == ("gallop", "ik"))   # goal+terminating, then hold (gallop-IK never holds)
    env.close()


REPLAY_WINDOW = 60


def _rollout_scenarios():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rollout_golden.json")) as f:
        return json.load(f)["scenarios"]


@pytest.mark.parametrize("sc", _rollout_scenarios(), ids=lambda s: s["name"])
def test_hip_path_reproduces_reference_rollouts(torch, sc):
    """The rollouts the reference's own env / robot code produced over the oracle's rigid-body step
    (tests/golden/make_rollout_golden.py), replayed on the HIP path through the reference's constructor arguments:
    reset observation, then per step observation, reward, done and motor command.  fp32 against fp64 along a contact-rich
    trajectory: the command (no physics in it until the goal logic reads the base) must agree to float rounding, the
    rest to the single-step tolerances of this file accumulated over the episode."""
    from rex_gym_amd import RexBatchEnv
    kw = dict(sc["env_kwargs"])
    if kw.pop("action_bias", None) is not None:
        kw["check_actions"] = False     # the overheat scenario commands a joint far outside the env's Box on purpose
    batch = kw.pop("batch", None)       # N different envs behind the reference's BatchEnv (per-env actions, resets by index)
    n = batch or 8                      # otherwise 8 copies of the one env
    task = sc["oracle_config"]["task"]
    signal = kw.pop("signal_type", sc["oracle_config"]["signal"])
    if "wrap" in kw:    # the reference's training stack around the env (LimitDuration, RangeNormalize, ClipAction): folded into the launch
        kw.update(range_normalize=True, max_episode_steps=kw.pop("wrap"))
    env = RexBatchEnv(n, task=task, signal_type=signal, **kw)
    worst = dict(obs=0.0, rate=0.0, reward=0.0, cmd=0.0)
    since_reset = np.zeros(n, int)
    rows = (lambda x: np.asarray(x)) if batch else (lambda x: np.tile(np.asarray(x), (n, 1)) if np.ndim(x) else np.full(n, x))
    for k, ev in enumerate(sc["events"]):
        if ev["kind"] == "reset":
            idx = ev.get("indices")
            if idx is None:
                obs, ref, who = env.reset().cpu().numpy(), rows(ev["obs"]), np.arange(n)
            else:
                obs, ref, who = env.reset(idx).cpu().numpy(), np.asarray(ev["obs"]), np.asarray(idx)
            since_reset[who] = 0
        else:
            o, r, d, info = env.step(torch.as_tensor(rows(ev["action"]).astype(np.float32), device="cuda"))
            obs, ref, who = o.cpu().numpy(), rows(ev["obs"]), np.arange(n)
            since_reset += 1
            # fp32 vs fp64 along contacts: the comparison stops before rounding differences dominate.  Measured divergence
            # (profiles/r02_parity.json, 4 096 / 8 192 envs against the fp32 oracle): max |joint error| over all envs 3e-6 rad
            # at step 25, 1e-5 at step 100 for walk-IK; 7e-3 at step 25 for gallop-OL (first landings) -- hence 60 steps.
            live = since_reset <= (25 if "latency" in sc["name"] else REPLAY_WINDOW)   # a PD loop on delayed velocities amplifies round-off 10x per 5 steps
            if not live.any():
                continue
            assert (d.cpu().numpy().astype(bool) == rows(ev["done"]).astype(bool))[live].all(), f"event {k}: done"
            worst["reward"] = max(worst["reward"], float(np.max(np.abs(r.cpu().numpy() - rows(ev["reward"]))[live])))
            worst["cmd"] = max(worst["cmd"], float(np.max(np.abs(info["action"].cpu().numpy() - rows(ev["cmd"]))[live])))
            obs, ref = obs[live], ref[live]
        if not batch:
            assert np.all(obs == obs[0]), "identical envs must stay identical"
        err = np.abs(obs - ref)
        rates = np.zeros(ref.shape[1], bool); rates[2:4] = True          # roll / pitch rate: O(1) rad/s signals
        worst["obs"] = max(worst["obs"], float(err[:, ~rates].max()))
        worst["rate"] = max(worst["rate"], float(err[:, rates].max()))
    print(sc["name"], worst)
    # mark 'arm': three arm joints are commanded beyond their limits (ARM_POSES['rest'] = -1.6 against -1.5 rad); Bullet's
    # limit rows exist only while a bound is violated, so they switch on and off around the bound and fp32 / fp64 runs
    # differ by one switch now and then: looser angular rates for those scenarios
    rate_tol = 0.25 if "arm" in sc["name"] else 5e-2
    assert worst["cmd"] < 2e-5 and worst["obs"] < 2e-3 and worst["rate"] < rate_tol and worst["reward"] < 2e-3, worst
    env.close()


def test_overheat_shutdown_on_the_gpu(torch):
    """Rex.ApplyAction's overheat protection (rex.py:601-608,617-623; every task env enables it): a motor whose torque
    stays above 2.45 N m for more than 1 000 substeps is switched off until the next Reset.  The fixture is what the
    reference's own Rex class did on the debug rack when the front-left foot joint was commanded 3 rad below its lower
    bound (rollout golden `walk_ol_on_rack_overheat`; the reference's _motor_enabled_list and _overheat_counter are
    recorded with every event).  The HIP path, through the same constructor arguments and actions: the enable mask agrees
    on EVERY event -- the shut-down lands on the same control step, the reset re-enables the motor -- the counter of the
    saturated motor agrees exactly (the others hover around the 2.45 N m threshold in both precisions: within a few
    counts), and after the shut-down the joint falls away from its bound under zero torque as the reference's does."""
    from rex_gym_amd import RexBatchEnv
    sc = next(s for s in _rollout_scenarios() if s["name"] == "walk_ol_on_rack_overheat")
    kw = dict(sc["env_kwargs"]); kw.pop("action_bias"); signal = kw.pop("signal_type")
    n = 8
    env = RexBatchEnv(n, task="walk", signal_type=signal, check_actions=False, **kw)
    en_w, oh_w = orclib.S_MOTOR_EN, orclib.S_OVERHEAT
    off_event, seen_off, seen_back_on = None, False, False
    for k, ev in enumerate(sc["events"]):
        if ev["kind"] == "reset":
            env.reset()
        else:
            env.step(torch.as_tensor(np.tile(np.asarray(ev["action"], np.float32), (n, 1)), device="cuda"))
        ps = product_state_to_numeric(env.state)
        assert np.all(ps == ps[:, :1]), "identical envs must stay identical"
        mask = int(ps[en_w, 0])
        assert [bool((mask >> j) & 1) for j in range(12)] == ev["motor_enabled"], (k, bin(mask), ev["motor_enabled"])
        packed = ps[oh_w:oh_w + 6, 0].astype(np.int64)
        counters = np.array([int(packed[j // 2] >> (16 * (j & 1))) & 0xFFFF for j in range(12)])
        assert counters[2] == ev["overheat"][2], (k, counters, ev["overheat"])
        assert np.abs(counters - np.asarray(ev["overheat"])).max() <= 6, (k, counters, ev["overheat"])
        if not ev["motor_enabled"][2]:
            if not seen_off:
                off_event, seen_off = k, True
                assert ev["overheat"][2] > 1000
            if k - off_event <= 10:     # zero torque on the joint: it leaves the bound exactly as the reference's does
                q, ref = ps[orclib.S_Q:orclib.S_Q + 12, 0], np.asarray(ev["body"])[13:25]
                others = np.arange(12) != 2
                np.testing.assert_allclose(q[others], ref[others], atol=2e-3)
                assert abs(q[2] - ref[2]) <= 3e-2        # the released joint swings away from its bound at ~5 rad/s: 2 ms apart is 1e-2 rad
        elif seen_off:
            seen_back_on = True
    assert seen_off and seen_back_on
    env.close()


def test_forward_reward_cap_and_action_check_of_the_batch_env(torch):
    """Two pieces of the reference's caller-facing semantics: RexGymEnv(forward_reward_cap=c) clips the forward term of
    the reward (rex_gym_env.py:525; lock step with the oracle), and the batch env rejects an action outside the env's Box
    like BatchEnv.step does (agents/tools/batch_env.py:76-79) -- by default, in float32, naming the env."""
    n = 64
    import parity_window as pw
    from rex_gym_amd import RexBatchEnv
    kw = dict(task="walk", signal_type="ik", seed=2, backwards=False, target_position=1.0, gait_clock_scale=1.5)   # the walking regime
    env = RexBatchEnv(n, forward_reward_cap=0.06, **kw)
    orc = pw.oracle_for(env)
    ref = RexBatchEnv(n, **kw)
    env.reset(); orc.reset(); ref.reset()
    rng = np.random.RandomState(0)
    gap = 0.0
    for k in range(200):
        a = rng.uniform(-0.4, 0.4, (n, 2)).astype(np.float32)
        _, r, _, _ = env.step(torch.as_tensor(a, device="cuda"))
        _, r0, _, _ = ref.step(torch.as_tensor(a, device="cuda"))
        _, orr, _, _ = orc.step(a)
        np.testing.assert_allclose(r.cpu().numpy(), orr, atol=1e-3)     # 200 steps of a walking robot, fp32 against fp64
        gap = max(gap, float((r0 - r).max()))
    assert gap > 0.03                                        # the uncapped reward did run ahead of the capped one
    with pytest.raises(ValueError, match="Invalid action at index 5"):
        bad = np.zeros((n, 2), np.float32); bad[5, 1] = 0.41
        env.step(bad)
    with pytest.raises(ValueError, match="Invalid action at index 0"):
        env.step(np.full((n, 2), np.nan, np.float32))
    env.step(np.full((n, 2), 0.4, np.float32))               # the bound itself (float32(0.4) > 0.4 in double) is inside
    env.close(); ref.close()


def test_policy_player_runs_a_checkpoint_on_the_batch_env(torch, tmp_path):
    """SimplePPOPolicy + play() on device tensors: a bundle written here stands in for a shipped checkpoint."""
    from test_agents_policy_player import policy_tensors, write_bundle
    from rex_gym_amd import RexBatchEnv
    from rex_gym_amd.agents.policy_player import SimplePPOPolicy, play
    prefix = str(tmp_path / "model.ckpt-9")
    write_bundle(prefix, policy_tensors(np.random.RandomState(5), obs_dim=4, layers=(200, 100), action_dim=2))
    env = RexBatchEnv(96, task="walk", signal_type="ik", seed=4)
    policy = SimplePPOPolicy(env, prefix)
    obs = env.reset()
    a = policy.get_action(obs)
    assert a.is_cuda and tuple(a.shape) == (96, 2) and bool((a.abs() <= 0.4 + 1e-6).all())
    ret, length, ended = play(env, policy, max_steps=60)
    assert tuple(ret.shape) == (96,) and bool(torch.isfinite(ret).all()) and int(length.max()) == 60 and not bool(ended.any())
    env.close()


@pytest.mark.parametrize("task,signal,obs_dim,action_dim", [("walk", "ik", 4, 2), ("gallop", "ol", 16, 4)])
def test_policy_player_with_the_actor_inside_the_launch(torch, tmp_path, task, signal, obs_dim, action_dim):
    """play_segments: the reference's policy player (playground/policy_player.py:22-56 -> simple_ppo_agent.py:53-88) with the checkpoint's
    network and observation filter evaluated inside the step launch, 20 closed-loop steps per launch, against play() -- the same
    checkpoint as PyTorch ops around one launch per step on the bare env.  The two evaluate the same mean action up to float32
    rounding of the network (different summation orders), so the episodes agree closely, not bit for bit: per-env return within 2e-2
    and length equal for >= 97 % of the envs over 60 steps; the first action, a function of the reset observation alone, within 1e-5.
    gallop: the inverted action Box of the reference (gallop_env.py:128-130) goes through both paths alike."""
    from test_agents_policy_player import policy_tensors, write_bundle
    from rex_gym_amd import RexBatchEnv
    from rex_gym_amd.agents.policy_player import SimplePPOPolicy, play, play_segments
    prefix = str(tmp_path / "model.ckpt-9")
    write_bundle(prefix, policy_tensors(np.random.RandomState(5), obs_dim=obs_dim, layers=(200, 100), action_dim=action_dim))
    n = 512
    bare = RexBatchEnv(n, task=task, signal_type=signal, seed=4, check_actions=False)
    fold = RexBatchEnv(n, task=task, signal_type=signal, seed=4, check_actions=False, range_normalize=True, auto_reset=True)
    p_bare, p_fold = SimplePPOPolicy(bare, prefix), SimplePPOPolicy(fold, prefix)
    # the first action: the policy on the reset observation, through both mappings
    a_bare = p_bare.get_action(bare.reset())
    from rex_gym_amd.agents.fused_actor import FusedActor
    FusedActor(fold, p_fold.network, p_fold._observ_filter, sample=False)
    _, _, _, info = fold.step_policy(fold.reset())
    lo = torch.as_tensor(bare.action_space.low, device="cuda", dtype=torch.float32); hi = torch.as_tensor(bare.action_space.high, device="cuda", dtype=torch.float32)
    a_fold = (info["policy_action"] + 1) / 2 * (hi - lo) + lo
    assert float((a_bare - a_fold).abs().max()) <= 1e-5 * max(1.0, float(hi.abs().max())) + 1e-6
    r0, l0, e0 = play(bare, p_bare, max_steps=60)
    r1, l1, e1 = play_segments(fold, p_fold, max_steps=60, segment=20)
    assert tuple(r1.shape) == (n,) and bool(torch.isfinite(r1).all())
    same_len = (l0 == l1)
    assert float(same_len.float().mean()) >= 0.97, float(same_len.float().mean())
    assert float(((r0 - r1).abs() <= 2e-2)[same_len].float().mean()) >= 0.97, float((r0 - r1).abs()[same_len].max())
    assert torch.equal(e0[same_len], e1[same_len])
    with pytest.raises(ValueError):
        play_segments(bare, p_bare, max_steps=10)
    bare.close(); fold.close()


def test_make_builds_the_registered_envs_with_their_step_limits(torch):
    import rex_gym_amd
    for env_id, (task, limit) in rex_gym_amd.ENV_IDS.items():
        env = rex_gym_amd.make(env_id, num_envs=8)
        assert env.task == task and env.num_envs == 8
        env.reset()
        _, _, done, _ = env.step(torch.zeros((8, env.action_dim), device="cuda"))
        assert done.shape == (8,)
        env.close()
    env = rex_gym_amd.make("RexPoses-v0", num_envs=4, max_episode_steps=3)       # the limit ends the episode (poses never falls)
    env.reset()
    dones = [bool(env.step(torch.zeros((4, 1), device="cuda"))[2].all()) for _ in range(3)]
    assert dones == [False, False, True]
    env.close()


@pytest.mark.parametrize("n", [64, 8195])
def test_latency_ring_survives_in_launch_auto_reset(torch, n):
    """With a latency configured the observation ring is part of what reset() restores (the reference's deque holds the
    last 100 observations of the reset motion, rex.py:309-323).  In-launch auto-reset must restore it exactly: with a
    fixed target and direction and the same actions, the second episode repeats the first bit for bit, and both follow
    the oracle."""
    from rex_gym_amd import RexBatchEnv
    kw = dict(task="walk", signal_type="ik", target_position=1.0, backwards=False, pd_latency=0.003, control_latency=0.0125,
              auto_reset=True, max_episode_steps=12, seed=2)
    env = RexBatchEnv(n, **kw)
    cfg = orclib.default_config("walk", "ik", 4, target_position=1.0, backwards=0, pd_latency=0.003, control_latency=0.0125,
                                auto_reset=1, max_episode_steps=12, seed=2)
    orc = orclib.OracleEnv(cfg, np.float32)
    first = env.reset().clone()
    orc.reset()
    rng = np.random.RandomState(11)
    acts = rng.uniform(-0.4, 0.4, (12, 2)).astype(np.float32)
    episodes = []
    for ep in range(3):
        seq = []
        for k in range(12):
            o, r, d, _ = env.step(torch.as_tensor(np.tile(acts[k], (n, 1)), device="cuda"))
            assert bool(d.all()) == (k == 11)
            seq.append(torch.cat([o, r[:, None]], dim=1).clone())
            oo, orr, od, _ = orc.step(np.tile(acts[k], (4, 1)))
            np.testing.assert_allclose(o[:4].cpu().numpy(), oo, atol=2e-3)
            np.testing.assert_allclose(r[:4].cpu().numpy(), orr, atol=1e-4)
        assert torch.equal(seq[-1][:, :first.shape[1]], first), "the observation after the auto-reset is the reset observation"
        episodes.append(torch.stack(seq))
    assert torch.equal(episodes[0], episodes[1]) and torch.equal(episodes[1], episodes[2])
    assert bool((episodes[0] == episodes[0][:, :1]).all()), "identical envs stay identical across the lane groups of a wave"
    env.close()


def test_rollout_gather_runs_on_rccl(torch):
    """The learner hand-off (rex_gym_amd/sharding.py) on the real backend: backend "nccl" is RCCL on ROCm.  One rank is
    all a 1-GPU box allows (RCCL refuses two ranks on one device); the collectives are still issued, for every dtype a
    rollout segment carries (float32 observations / actions / rewards, bool done flags).  The multi-rank layout is
    covered on CPU with gloo (tests/test_sharding_gloo.py)."""
    import torch.distributed as dist
    from rex_gym_amd import RexBatchEnv
    from rex_gym_amd.sharding import Shard, gather_rollout
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        shard = Shard(0, 1, 64)
        env = RexBatchEnv(task="walk", signal_type="ik", seed=5, **shard.env_kwargs())
        env.reset()
        seg = dict(obs=[], action=[], reward=[], done=[])
        for k in range(5):
            a = torch.zeros((64, 2), device="cuda")
            o, r, d, _ = env.step(a)
            seg["obs"].append(o.clone()); seg["action"].append(a); seg["reward"].append(r.clone()); seg["done"].append(d.clone())
        seg = {k: torch.stack(v) for k, v in seg.items()}
        out = gather_rollout(seg, always=True)
        for k in seg:
            assert out[k].dtype == seg[k].dtype and torch.equal(out[k], seg[k]), k
        env.close()
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------ every kernel variant against the oracle, at BASELINE sizes
# per-env joint RMSE over the 200-step window against the fp64 oracle: (median, p99, max) bounds = 2 x the values measured
# on MI355X (profiles/r03_parity.json, unchanged in profiles/r04_parity.json); BASELINE.json's bar is 1e-3 rad
_PARITY_BOUNDS = {
    # measured (4 / 8 / 16 / 64 envs per wave):      median            p99               max
    "walk_ik_4096": (6e-6, 5e-4, 4e-3),                # 1.5-2.9e-6      1.4-2.4e-4        1.2-1.8e-3
    "walk_ik_8192": (6e-6, 5e-4, 6e-3),                # north_star's shard (65 536 / 8): the walk-IK window at twice the envs
    "gallop_ol_8192": (3e-6, 6e-5, 4e-3),              # 1.3e-6          2.3-2.5e-5        3.9-5.7e-4 (round 5: 2.1e-3 at 4 / 8 / 16, ONE env of
    #                                                     8 192 after an event flip; 6.7e-4 at 64; the oracle's own fp32 build 6.1e-4)
    # the heightfield is the float32 floor, not the kernels: the fp32 build of the ORACLE against its own fp64 build reads
    # median 7e-6 / p99 3.0e-3 / max 1.9e-2 on this workload (tools/parity_report.py --floor) -- the turn env drops the robot
    # onto 45-degree facets with its toes up to 2.5 cm inside the terrain, and a toe on the other side of a facet edge is
    # another contact normal.  BASELINE.json's 1e-3 bar is met by the median only.
    "turn_ik_heightfield_4096": (2e-5, 7.5e-3, 4.5e-2),  # 8.0-8.6e-6    3.0-3.6e-3        2.2e-2
    # mark arm: three arm joints are commanded beyond their bounds and sit ON them; their limit rows switch with the last
    # bit of the joint angle, in float32 on other substeps than in float64: the floor of this workload (fp32 oracle
    # against fp64 oracle) is median 2.9e-4 / p99 3.7e-3 / max 3.9e-2 over the 18 joints
    "mixed_arm_2048": (6e-4, 8e-3, 8e-2),
    "walk_ik_gait_clock_1.5_4096": (8e-6, 1.6e-4, 3.2e-3),  # 2.5-3.7e-6  3.8-7.7e-5       1.2-1.6e-3
}
_MEETS_THE_BAR_AT_P99 = {"walk_ik_4096", "walk_ik_8192", "gallop_ol_8192", "walk_ik_gait_clock_1.5_4096"}
# Share of the envs whose event sequence over the 200-step window (toe points in reach, heightfield facets, joint / arm bounds
# reached, substep by substep: rex_set_event_trace) equals the fp64 oracle's -- lower bounds at ~0.8 x measured.  On that subset
# the joint RMSE meets BASELINE.json's 1e-3 rad at the 99th percentile in EVERY workload (heightfield: p99 7.8e-5, max 7.8e-4):
# what misses the bar in the table above is envs in which float32 took a discrete decision the other way, not drift.  Mark arm
# keeps no env: its three arm joints sit ON their bounds and the limit rows switch with the last bit in every env within ~25
# steps (its single steps from common states are split the same way above: same events -> 2e-4 rad).
_SINGLE_STEP_SAME_EVENTS_RAD = {"walk_ik_4096": 2e-4, "walk_ik_8192": 2e-4, "gallop_ol_8192": 2e-4, "turn_ik_heightfield_4096": 1, "mixed_arm_2048": 1,
                                "walk_ik_gait_clock_1.5_4096": 2e-4}
_SAME_EVENTS_SHARE = {"walk_ik_4096": 0.95, "walk_ik_8192": 0.95, "gallop_ol_8192": 0.97, "turn_ik_heightfield_4096": 0.33, "mixed_arm_2048": 0.0,
                      "walk_ik_gait_clock_1.5_4096": 0.9}


@pytest.mark.parametrize("epw", [4, 8, 16, 64])
@pytest.mark.parametrize("name", ["walk_ik_4096", "gallop_ol_8192", "turn_ik_heightfield_4096", "mixed_arm_2048", "walk_ik_gait_clock_1.5_4096", "walk_ik_8192"])
def test_every_kernel_variant_against_the_oracle_at_baseline_sizes(torch, name, epw, monkeypatch):
    """BASELINE.json configs[1..4] at their per-GPU shard sizes (4 096 walk-IK, 8 192 gallop-OL, 4 096 turn-IK on the
    heightfield pool, 2 048 mark-arm envs with per-env mixed tasks and per-reset mass / friction draws) and the walking
    workload (gait clock 1.5), each kernel variant (4 / 8 / 16 envs per wave = lane groups, 64 = one env per lane) DIRECTLY
    against the FP64 oracle:
      * single env.step() from common oracle states (taken 40 and 120 steps into the rollout): max over ALL envs of the
        joint-angle error <= 2e-4 rad, base position <= 2e-4 m, and the flags word -- every discrete decision -- equal;
      * the first 200 control steps (1 s of robot time) from reset under the same actions: per-env joint RMSE over the
        window -- 99th percentile <= 1e-3 rad (BASELINE.json's bar) and median / p99 / max within twice the measured
        values.  Contact dynamics amplify fp32 round-off (a toe that touches down one substep apart is a transient of
        ~1e-2 rad), so the max is looser than the median; the divergence curve goes to gpurun_out/ for
        profiles/r06_parity.json (tools/parity_report.py), with the event-trace split of the window.
    Round 5: the window's error figures come from the PRODUCT kernels (tests/parity_window.py: pass 1 without the event
    trace); the `_trace` instantiations run a second pass for the event split and must reproduce pass 1 bit for bit."""
    import json
    import os
    import parity_window as pw
    if epw == 64 and name == "mixed_arm_2048":
        pytest.skip("mark arm / mixed tasks: lane-group kernels only")
    if name == "walk_ik_8192" and epw not in (8, 16):
        pytest.skip("north_star's shard: the variant the host selects at 8 192 envs (8 per wave) and the next one")
    monkeypatch.setenv("REX_ENVS_PER_WAVE", str(epw))
    steps, seed = 200, 23
    env = pw.make_env(name, seed=seed)
    assert env._L.rex_envs_per_wave(env._h) == epw
    acts, oq, opos, odone, ostates, (otrace, pre_trace) = pw.oracle_trajectory(name, env, steps, seed)
    nm = env.num_motors
    # --- single steps from common states, with the event trace on: the envs whose step took every discrete decision of the
    #     contact set-up as the oracle's did (same toe points in reach, same facets, same bounds reached in every substep) are
    #     held to the round-off tolerance WITHOUT exception; the quantile bounds below are for the others
    single = []
    for k0, st in ostates.items():
        env.reset()
        env.state.copy_(numeric_to_product_state(st, torch, env.state.device))
        kt = env.set_event_trace(True)
        kt.copy_(torch.from_numpy(pre_trace[k0].view(np.int32)))      # the oracle's chain before this step
        env.step(torch.as_tensor(acts[k0], device="cuda"))
        same = (kt.cpu().numpy().view(np.uint32)[0] == otrace[k0, 0])
        env.set_event_trace(False)
        ps = product_state_to_numeric(env.state)
        live = ~odone[k0]
        eq = np.abs(ps[orclib.S_Q:orclib.S_Q + nm] - oq[k0])[:, live]
        ep = np.abs(ps[0:3] - opos[k0])[:, live]
        sm = same[live]
        single.append(dict(from_step=k0, envs=int(live.sum()), same_events=int(sm.sum()),
                           joint_err_max_same_events=float(eq[:, sm].max()) if sm.any() else None,
                           joint_err_max_other_events=float(eq[:, ~sm].max()) if (~sm).any() else None))
        assert sm.mean() >= (0.5 if "arm" in name else 0.97), (epw, k0, sm.mean())
        assert eq[:, sm].max() <= _SINGLE_STEP_SAME_EVENTS_RAD[name] and ep[:, sm].max() <= 2e-4, (epw, k0, eq[:, sm].max(), ep[:, sm].max())
        if "heightfield" in name:
            # a toe point within float32 resolution of a triangle edge of the field (45-degree facets between the 2 x 2
            # blocks) takes the neighbouring facet's normal in one of the two precisions: a jump of several 1e-3 rad for
            # that env in that step -- measured for 5 of 4 096 envs per step
            assert (eq.max(0) <= 2e-4).mean() >= 0.995 and (ep.max(0) <= 2e-4).mean() >= 0.995, (epw, k0, eq.max(), ep.max())
        elif "arm" in name:
            # three arm joints are commanded beyond their bounds (ARM_POSES['rest'] = -1.6 against -1.5 rad) and sit ON
            # them: Bullet's limit row exists only while the bound is violated, so it switches on and off with the last
            # bit of the joint angle -- in float32 on other substeps than in float64 -- and every switch moves the arm
            # joints by ~1e-3 rad.  The 12 leg joints and the base are held to the single-step tolerance, the arm joints
            # to that jitter.
            assert (eq[:12].max(0) <= 2e-4).mean() >= 0.95 and eq[:12].max() <= 2e-3 and (ep.max(0) <= 2e-4).mean() >= 0.995, (epw, k0, eq[:12].max(), ep.max())
            assert eq[12:].max() <= 5e-3, (epw, k0, eq[12:].max())    # (measured: legs 96 % / 5.6e-4 -- the 1 kg arm pushes the base when a row switches --, arm 2.3e-3)
        else:
            assert eq.max() <= 2e-4 and ep.max() <= 2e-4, (epw, k0, eq.max(), ep.max())
    # --- 200-step window from reset (episode counters back to 0: the oracle's rollout is every env's FIRST episode, and the
    #     episode number keys the Philox draws of target / direction)
    env.state.zero_()
    rec = pw.window(name, env, steps=steps, seed=seed)      # error figures: the PRODUCT kernels; event split: a second, traced pass
    rec["envs_per_wave"] = epw
    rec["single_steps_from_common_states"] = single
    floor = rec["float32_floor"] = pw.float32_floor(name, env, steps, seed)     # the oracle's fp32 build against its fp64 build, this window
    # the ABSOLUTE verdict of north_star's bar, recorded next to the floor-relative one and never relaxed by it
    rec["meets_1e-3_rad_absolute"] = dict(median=bool(rec["median_rad"] <= 1e-3), p99=bool(rec["p99_rad"] <= 1e-3), max=bool(rec["max_rad"] <= 1e-3))
    os.makedirs(os.path.join(os.path.dirname(__file__), "..", "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "parity_windows.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")
    print(json.dumps(rec))
    # the traced pass reproduced the product pass bit for bit after every one of the 200 steps: the event split below annotates
    # the product kernels' own trajectory
    assert rec["trace_pass_bit_identical"], (name, epw, rec["trace_pass_identical_steps"])
    med, p99, mx = _PARITY_BOUNDS[name]
    assert rec["median_rad"] <= 1e-3 and (name not in _MEETS_THE_BAR_AT_P99 or rec["p99_rad"] <= 1e-3), rec      # BASELINE.json's bar
    assert rec["meets_1e-3_rad_absolute"]["p99"] == (name in _MEETS_THE_BAR_AT_P99), (name, rec["p99_rad"])      # (which workloads meet it is itself pinned)
    assert rec["median_rad"] <= med and rec["p99_rad"] <= p99 and rec["max_rad"] <= mx, rec
    # the stated fp32 tolerance where 1e-3 rad at the 99th percentile is below what float32 itself gives on the workload (heightfield,
    # mark arm): the whole-batch p99 within the FLOAT32 FLOOR of this very window + 20 % -- measured in the same run, not a constant
    assert rec["p99_rad"] <= max(1e-3, 1.2 * floor["p99_rad"]), (name, epw, rec["p99_rad"], floor)
    assert rec["median_rad"] <= max(1e-5, 1.2 * floor["median_rad"]), (name, epw, rec["median_rad"], floor)
    # the envs whose whole 200-step event sequence was the oracle's: BASELINE.json's bar at the 99th percentile, every workload
    ev = rec["events"]
    share = _SAME_EVENTS_SHARE[name]
    assert ev["share_same_event_sequence"] >= share, ev
    if share > 0:
        assert ev["joint_rmse_same_events"]["p99_rad"] <= 1e-3 and ev["joint_rmse_same_events"]["median_rad"] <= 2e-5, ev
    # ... and every env, for as long as its event sequence is the oracle's (mark arm: 5-10 control steps, then a bound row switches)
    u = ev["joint_rmse_until_first_divergence"]
    assert u["envs"] >= 0.95 * env.num_envs and u["p99_rad"] <= 1e-3, ev
    env.close()


def test_sensor_noise_model(torch):
    """Rex(observation_noise_stdev=...) (rex.py:22,765-769): Gaussian noise in the sensor getters.  Every getter call
    site of a step draws from the env's Philox stream, in the kernels as in the oracle, so the two agree sample for
    sample; and the noise has the configured standard deviation, is independent across envs and steps, and leaves the
    physics untouched."""
    from rex_gym_amd import RexBatchEnv
    n = 512
    stdev = (0.01, 0.3, 0.2, 0.02, 0.5)
    env = RexBatchEnv(n, task="gallop", signal_type="ol", seed=8, observation_noise_stdev=stdev)
    clean = RexBatchEnv(n, task="gallop", signal_type="ol", seed=8)
    cfg = orclib.default_config("gallop", "ol", n, seed=8)
    for k, v in enumerate(stdev):
        cfg.noise_stdev[k] = v
    orc = orclib.OracleEnv(cfg, np.float32)
    o0, c0, oo0 = env.reset().cpu().numpy(), clean.reset().cpu().numpy(), orc.reset()
    np.testing.assert_allclose(o0, oo0, atol=3e-3)
    assert 0.015 < (o0[:, 0] - c0[:, 0]).std() < 0.025 and 0.4 < (o0[:, 2] - c0[:, 2]).std() < 0.6
    rng = np.random.RandomState(2)
    d_obs, d_rew = [], []
    for k in range(12):
        a = rng.uniform(-0.3, 0.3, (n, 4)).astype(np.float32)
        o, r, d, _ = env.step(torch.as_tensor(a, device="cuda"))
        co, cr, cd, _ = clean.step(torch.as_tensor(a, device="cuda"))
        oo, orr, od, _ = orc.step(a)
        np.testing.assert_allclose(o.cpu().numpy(), oo, atol=3e-2)
        np.testing.assert_allclose(r.cpu().numpy(), orr, atol=5e-3)
        d_obs.append((o - co).cpu().numpy()); d_rew.append((r - cr).cpu().numpy())
    np.testing.assert_array_equal(env.state[:37].cpu().numpy(), clean.state[:37].cpu().numpy())    # the physics never sees the noise
    d_obs = np.stack(d_obs)
    for col, sd in ((0, 0.02), (1, 0.02), (2, 0.5), (3, 0.5), (4, 0.01), (15, 0.01)):
        x = d_obs[:, :, col].ravel()
        assert abs(x.mean()) < 4 * sd / np.sqrt(x.size) + 1e-4 and 0.93 * sd < x.std() < 1.07 * sd, (col, x.mean(), x.std())
    assert abs(np.corrcoef(d_obs[0, :, 0], d_obs[1, :, 0])[0, 1]) < 0.15 and abs(np.corrcoef(d_obs[0, :-1, 0], d_obs[0, 1:, 0])[0, 1]) < 0.15
    assert np.std(np.stack(d_rew)) > 0           # energy term: noisy torques . noisy velocities
    env.close(); clean.close()


def test_caller_supplied_heightfield_terrain(torch):
    """Heightfields other than the reference's random one (model/terrain.py:55-78 'hills' / 'mounts' / 'maze', whose data
    files are not part of rex-gym): any nx x ny grid, cell size, placement and drop height through rex_set_heightfield.
    A synthetic field of rolling bumps (96 x 64 vertices, 8 x 6 cm cells, off-centre, raised by 0.5 m): the robots settle ON
    it, and the HIP path follows the fp32 oracle on the same field."""
    from rex_gym_amd import RexBatchEnv
    ny, nx, cell = 64, 96, (0.08, 0.06)
    yy, xx = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    field = (0.03 * np.sin(0.35 * xx) * np.cos(0.3 * yy) + 0.03).astype(np.float32)        # 0 .. 0.06 m
    fields = np.stack([field, field[::-1].copy()])
    origin = (0.3, -0.2, 0.5)
    n = 32
    env = RexBatchEnv(n, task="turn", signal_type="ik", seed=2, terrain_type="custom", heightfield=fields, heightfield_cell=cell,
                      heightfield_origin=origin, init_height=0.75)
    cfg = orclib.default_config("turn", "ik", n, seed=2, init_height=0.75)
    orc = orclib.OracleEnv(cfg, np.float32)
    mids = np.array([0.5 * (f.min() + f.max()) - origin[2] for f in fields], np.float32)
    orc.set_heightfield(fields, mids, cell, origin[:2])
    # (the robots are still rocking on the bumps when the 600 settle substeps end -- pitch rate -1.3 rad/s --, so the two float32
    #  paths have parted by up to 8e-3 in the rate words by then: measured 8.2e-3 since the contraction fix of round 5, 2e-3 before)
    hip_reset, o32_reset = env.reset().cpu().numpy(), orc.reset()
    np.testing.assert_allclose(hip_reset, o32_reset, atol=1.5e-2)
    # the float32 floor of this scenario, measured here: the oracle's fp32 build against its fp64 build through the same reset and steps
    # (reset observation 0.13 apart, joint angles after the 25 steps 5.6e-3 at the median: tumbling robots amplify the last bit)
    orc64 = orclib.OracleEnv(cfg, np.float64)
    orc64.set_heightfield(fields, mids, cell, origin[:2])
    o64_reset = orc64.reset()
    # ... and the gate that does not move with the kernels' contraction mode: the HIP path against the FP64 oracle, held to the float32
    # floor of the same reset (the fp32 oracle -- built with contraction off -- against the fp64 one): per observation word, the median
    # over the envs of the HIP error within 1.2 x the floor's median, the worst env within 1.2 x the floor's worst
    e_hip, e_floor = np.abs(hip_reset - o64_reset), np.abs(o32_reset - o64_reset)
    assert np.all(np.median(e_hip, 0) <= 1.2 * np.median(e_floor, 0) + 1e-4), (np.median(e_hip, 0), np.median(e_floor, 0))
    assert e_hip.max() <= 1.2 * e_floor.max() + 1e-3, (e_hip.max(), e_floor.max())
    ps, os_ = product_state_to_numeric(env.state), orc.get_state()
    # shape centred on (min + max) / 2 = 0.03 and placed at z = 0.5: the surface spans 0.47 .. 0.53, the base stands ~0.2 above it
    assert np.all(ps[2] > 0.62) and np.all(ps[2] < 0.76)
    np.testing.assert_allclose(ps[:7], os_[:7], atol=3e-4)
    rng = np.random.RandomState(4)
    same = np.ones(n, bool)
    for k in range(25):
        a = rng.uniform(-0.01, 0.01, (n, 2)).astype(np.float32)
        o, r, d, info = env.step(torch.as_tensor(a, device="cuda"))
        oo, orr, od, ocmd = orc.step(a)
        orc64.step(a)
        same &= np.abs(info["action"].cpu().numpy() - ocmd).max(1) < 5e-5      # (a yaw goal test may flip one step apart)
    ps, os_ = product_state_to_numeric(env.state), orc.get_state()
    err = np.abs(ps[13:25] - os_[13:25]).max(0)
    floor = np.median(np.abs(os_[13:25] - orc64.get_state()[13:25]).max(0))
    # HIP vs the fp32 oracle: within half the float32 floor of the scenario at the median (measured 1.2e-3 against a floor of 5.6e-3)
    assert same.mean() > 0.9 and np.median(err) < max(1e-3, 0.5 * floor) and (err[same] < 2e-2).mean() > 0.9, (np.median(err), floor)
    env.close()
    with pytest.raises(NotImplementedError, match="pybullet_data"):
        RexBatchEnv(4, task="walk", terrain_type="mounts")


@pytest.mark.parametrize("mark,n", [("base", 64), ("arm", 24), ("base", 9000)])
def test_link_box_ground_contacts(torch, mark, n):
    """RexConfig.body_contacts: the link collision boxes of rex.urdf (base, chassis, shoulder, leg, foot) against the
    ground.  Limp robots (kp = kd = 0: only the back-EMF term is left) dropped ON THEIR SIDE from 0.25 m: the toes cannot
    catch them; without the rows the base falls through the floor, with them the robot comes to rest on its boxes -- and
    the HIP path follows the fp32 oracle through the fall (the first 40 steps in lock step; the resting height at the end)."""
    from rex_gym_amd import RexBatchEnv
    kw = dict(seed=5, motor_kp=0.0, motor_kd=0.0, mark=mark)

    def on_its_side(env_state_numeric):
        st = env_state_numeric.copy()
        st[2] = 0.25
        st[3], st[4], st[5], st[6] = np.sin(0.785), 0.0, 0.0, np.cos(0.785)
        st[7:13] = 0.0
        return st

    env, orc = make_pair("poses", "ik", n, np.float32, body_contacts=1, **kw)     # poses: the env that never terminates
    env.reset(); orc.reset()
    st = on_its_side(orc.get_state())
    orc.set_state(st)
    env.state.copy_(numeric_to_product_state(st, torch, env.state.device))
    zero = np.zeros((n, 1), np.float32)
    for k in range(300):
        env.step(torch.as_tensor(zero, device="cuda"))
        if n <= 64:
            orc.step(zero)
            if k < 40:
                ps, os_ = product_state_to_numeric(env.state), orc.get_state()
                np.testing.assert_allclose(ps[:3], os_[:3], atol=3e-3, err_msg=f"step {k}")
    ps, os_ = product_state_to_numeric(env.state), orc.get_state()
    assert np.isfinite(ps[:37]).all()
    assert np.all(ps[2] > 0.03) and np.all(ps[2] < 0.14), (ps[2].min(), ps[2].max())   # resting on the boxes, above the floor
    assert np.abs(ps[7:10]).max() < 0.05                                                # and at rest
    if n <= 64:
        np.testing.assert_allclose(ps[2], os_[2], atol=1.5e-2)
    env.close()
    if n <= 64:
        sunk = RexBatchEnv(n, task="poses", signal_type="ik", body_contacts=False, **kw)
        sunk.reset()
        sunk.state.copy_(numeric_to_product_state(st, torch, sunk.state.device))
        for k in range(300):
            sunk.step(torch.as_tensor(zero, device="cuda"))
        assert float(sunk.state[2].max()) < 0.0          # toes only: the base goes through the floor
        sunk.close()


def test_link_box_rows_change_nothing_while_no_box_touches_the_ground(torch):
    """With body_contacts on, a robot that walks on its toes must step exactly as with the toes-only kernel of the same
    lane layout (the rows exist only for boxes below the ground)."""
    from rex_gym_amd import RexBatchEnv
    n = 128
    a = RexBatchEnv(n, task="walk", signal_type="ik", seed=3, body_contacts=True)
    b = RexBatchEnv(n, task="walk", signal_type="ik", seed=3)
    oa, ob = a.reset(), b.reset()
    # (the two template instantiations round differently: the settled robot's residual rates, 3e-5 rad/s, differ in sign)
    np.testing.assert_allclose(oa.cpu().numpy(), ob.cpu().numpy(), atol=1e-4)
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    for k in range(40):
        act = torch.rand((n, 2), device="cuda", generator=g) * 0.8 - 0.4
        xa, ra, da, _ = a.step(act)
        xb, rb, db, _ = b.step(act)
        assert torch.equal(da, db)
        np.testing.assert_allclose(xa.cpu().numpy()[:, :2], xb.cpu().numpy()[:, :2], atol=2e-4)
    np.testing.assert_allclose(a.state[13:25].cpu().numpy(), b.state[13:25].cpu().numpy(), atol=2e-3)
    a.close(); b.close()


@pytest.mark.parametrize("n", [3000, 9000, 70000])    # the one-workgroup sort (<= 4 096 envs) and the many-workgroup one (9 and 69 chunks, ragged last chunk)
def test_regrouped_batch_is_bit_identical(torch, monkeypatch, n):
    """REX_REGROUP=1: envs are regrouped into waves by the solver sweeps of the previous step; the wave slot -> env
    permutation changes every step, an env's results must not."""
    from rex_gym_amd import RexBatchEnv
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("REX_REGROUP", flag)
        monkeypatch.setenv("REX_ENVS_PER_WAVE", "16")
        env = RexBatchEnv(n, task="walk", signal_type="ik", seed=9, auto_reset=True, max_episode_steps=40)
        env.reset()
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        rec = []
        for k in range(60):
            a = torch.rand((n, 2), device="cuda", generator=g) * 0.8 - 0.4
            o, r, d, info = env.step(a)
            rec.append((o.clone(), r.clone(), d.clone(), info["action"].clone()))
        outs[flag] = (rec, env.state.clone())
        env.close()
    for (o0, r0, d0, c0), (o1, r1, d1, c1) in zip(outs["0"][0], outs["1"][0]):
        assert torch.equal(o0, o1) and torch.equal(r0, r1) and torch.equal(d0, d1) and torch.equal(c0, c1)
    assert torch.equal(outs["0"][1], outs["1"][1])


@pytest.mark.parametrize("n,epw,mark", [(700, 4, "arm"), (5000, 16, "arm"), (3000, 8, "base"), (5000, 8, "arm")])    # one chunk of the count / scatter pair and several, ragged; both marks
def test_regrouped_mixed_batch_is_bit_identical(torch, monkeypatch, n, epw, mark):
    """A REX_TASK_MIXED batch under REX_REGROUP=1: every task owns a static region of whole waves of the slot map and is
    sorted inside it by the solver sweeps of the previous step (rex_regroup_mixed_*); against the chunked static map
    (REX_REGROUP=0) the waves change every step, an env's results must not -- through falls, in-launch resets and the
    per-reset mass / friction draws."""
    from rex_gym_amd import RexMixedBatchEnv
    kw = dict(mark=mark, seed=6, auto_reset=True, max_episode_steps=30, mass_scale_range=(0.8, 1.2), friction_range=(0.25, 0.625))
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("REX_REGROUP", flag)
        monkeypatch.setenv("REX_ENVS_PER_WAVE", str(epw))
        env = RexMixedBatchEnv(n, **kw)
        env.reset()
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        rec = []
        for k in range(45):
            a = torch.rand((n, 2), device="cuda", generator=g) * 0.02 - 0.01
            o, r, d, info = env.step(a)
            rec.append((o.clone(), r.clone(), d.clone(), info["action"].clone()))
        outs[flag] = (rec, env.state.clone())
        env.close()
    assert sum(int(x[2].sum()) for x in outs["0"][0]) >= n       # every env ended an episode
    for (o0, r0, d0, c0), (o1, r1, d1, c1) in zip(outs["0"][0], outs["1"][0]):
        assert torch.equal(o0, o1) and torch.equal(r0, r1) and torch.equal(d0, d1) and torch.equal(c0, c1)
    assert torch.equal(outs["0"][1], outs["1"][1])


@pytest.mark.parametrize("task,signal,mark", [("walk", "ik", "base"), ("turn", "ik", "base"), ("gallop", "ol", "arm")])
def test_on_rack_debug_mode_against_the_oracle(torch, task, signal, mark):
    """on_rack=True (loadURDF(useFixedBase=True) at [0, 0, 1], rex.py:269-287): the base stays exactly where the rack holds
    it, the legs follow the oracle's fixed-base articulated-body dynamics.  Tolerance: joints 1e-4 rad over 40 control
    steps (no contacts: the chain is smooth)."""
    n = 96
    env, orc = make_pair(task, signal, n, np.float32, seed=4, on_rack=1, mark=mark)
    env.reset(); orc.reset()
    nm = 18 if mark == "arm" else 12
    s0 = product_state_to_numeric(env.state)
    yaw = 2.1 if task == "turn" else 0.0
    np.testing.assert_allclose(s0[:7], np.tile(np.array([0, 0, 1, 0, 0, np.sin(yaw / 2), np.cos(yaw / 2)], np.float32)[:, None], (1, n)),
                               atol=1e-7)
    np.testing.assert_allclose(s0[orclib.S_Q:orclib.S_Q + nm], orc.get_state()[orclib.S_Q:orclib.S_Q + nm], atol=1e-4)
    rng = np.random.RandomState(2)
    lo, hi = np.minimum(env.action_space.low, env.action_space.high), np.maximum(env.action_space.low, env.action_space.high)
    for _ in range(40):
        a = rng.uniform(lo, hi, (n, env.action_dim)).astype(np.float32)
        po, pr, pd, _ = env.step(torch.as_tensor(a, device="cuda"))
        oo, orw, od, _ = orc.step(a)
        np.testing.assert_allclose(po.cpu().numpy(), oo, atol=2e-4)
        np.testing.assert_array_equal(pd.cpu().numpy(), od)
    ps, os_ = product_state_to_numeric(env.state), orc.get_state()
    np.testing.assert_array_equal(ps[:7], s0[:7])                                    # the rack holds
    np.testing.assert_array_equal(ps[7:13], np.zeros((6, n), np.float32))
    np.testing.assert_allclose(ps[orclib.S_Q:orclib.S_Q + nm], os_[orclib.S_Q:orclib.S_Q + nm], atol=1e-4)
    assert np.ptp(ps[orclib.S_Q:orclib.S_Q + 12], axis=1).max() > 0 or task == "turn"
    env.close()


def test_env_randomizer_hooks_of_the_single_env_classes(torch):
    """An EnvRandomizer in the reference's protocol (rex_gym_env.py:345-346,400-401): `randomize_env(env)` after every
    reset's Rex.Reset, `randomize_step(env)` before every step, reaching the robot through env.rex (rex.py:643-692)."""
    from rex_gym_amd.envs.gym import RexWalkEnv

    class Randomizer:
        def __init__(self):
            self.resets = self.steps = 0
            self.rng = np.random.RandomState(0)

        def randomize_env(self, env):
            self.resets += 1
            self.ratio = self.rng.uniform(0.8, 1.2)
            env.rex.SetBaseMasses([m * self.ratio for m in env.rex.GetBaseMassesFromURDF()])
            env.rex.SetLegMasses([m * 1.1 for m in env.rex.GetLegMassesFromURDF()])
            env.rex.SetFootFriction(0.4)

        def randomize_step(self, env):
            self.steps += 1

    r = Randomizer()
    env = RexWalkEnv(env_randomizer=r)
    assert isinstance(env, RexWalkEnv)
    for episode in range(2):
        env.reset()
        np.testing.assert_allclose(env._batch.body_params.cpu().numpy()[:, 0], [r.ratio, 1.1, 0.4], rtol=1e-6)
        for _ in range(3):
            env.step(np.zeros(2, np.float32))
    assert (r.resets, r.steps) == (2, 6)
    # and the knobs act: a heavier robot sags more on the same motors
    z = []
    for scale in (0.8, 1.2):
        class Fixed:
            def randomize_env(self, env, scale=scale):
                env.rex.SetBaseMasses([m * scale for m in env.rex.GetBaseMassesFromURDF()])
        e = RexWalkEnv(env_randomizer=[Fixed()])
        e.reset()
        for _ in range(40):
            e.step(np.zeros(2, np.float32))
        z.append(float(e._batch.state[2, 0]))
        e.close()
    assert z[1] < z[0] - 1e-4
    env.close()


@pytest.mark.parametrize("n,epw", [(64, 4), (4096, 4), (2048, 8)])
def test_self_collision_rows_of_the_rolled_pose(torch, n, epw, monkeypatch):
    """URDF_USE_SELF_COLLISION (rex.py:276-281) where it acts: RexPosesEnv rolling the base by 0.74 rad drives the edge of
    the base box 3 mm into the upper-leg boxes of the low side (profiles/r02_contact_census.md).  With body_contacts the
    leg boxes carry vertex-face rows against the base body's boxes: same rows in the oracle and in the kernels, and the
    overlap is gone."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import contact_census as cc
    monkeypatch.setenv("REX_ENVS_PER_WAVE", str(epw))
    env, orc = make_pair("poses", "ik", n, np.float32, seed=6, base_roll=-0.74, body_contacts=1)
    assert env._L.rex_envs_per_wave(env._h) == epw          # both link-box instantiations, at BASELINE batch sizes
    free, _ = make_pair("poses", "ik", n, np.float32, seed=6, base_roll=-0.74, body_contacts=0)
    env.reset(); orc.reset(); free.reset()
    rng = np.random.RandomState(4)
    acts = rng.uniform(-0.1, 0.1, (260, n, env.action_dim)).astype(np.float32)
    for a in acts[:200]:
        orc.step(a)
    # single-step parity from the oracle's states while the rows act; tolerance 1e-4 rad / 1e-4 m, 2 % of the envs may
    # differ (a corner that is inside its box in one precision and outside in the other, at zero depth)
    bad = np.zeros(n, bool)
    for a in acts[200:215]:
        env.state.copy_(numeric_to_product_state(orc.get_state(), torch, env.state.device))
        env.step(torch.as_tensor(a, device="cuda")); orc.step(a)
        ps, os_ = product_state_to_numeric(env.state), orc.get_state()
        bad |= np.abs(ps[orclib.S_Q:orclib.S_Q + 12] - os_[orclib.S_Q:orclib.S_Q + 12]).max(axis=0) > 1e-4
        bad |= np.abs(ps[:7] - os_[:7]).max(axis=0) > 1e-4
    assert bad.mean() <= 0.02, bad.mean()
    # whole rollouts on the GPU: with the rows the boxes stay out of each other, without them they do not
    env.reset()
    for a in acts:
        ta = torch.as_tensor(a, device="cuda")
        env.step(ta); free.step(ta)
    t = cc.model_tables(); boxes = cc.link_boxes(t); names = [b[0] for b in boxes]

    def deepest(state):
        worst = []
        for i in range(0, n, n // 8):
            R, p = cc.fk(t, state[:, i].astype(np.float64))
            world = [(R[b[1]], p[b[1]] + R[b[1]] @ b[2], b[3]) for b in boxes]
            worst.append(min(cc.sat_separation(*world[names.index(A)], *world[names.index(leg + "_leg")])
                             for A in ("base", "chassis_rear", "chassis_front") for leg in ("FL", "FR", "RL", "RR")))
        return min(worst)
    with_rows, without = deepest(product_state_to_numeric(env.state)), deepest(product_state_to_numeric(free.state))
    print("deepest leg-box / base-box overlap [mm]: with rows %.3f, without %.3f" % (1e3 * with_rows, 1e3 * without))
    assert without < -2e-3 and with_rows > -3e-4
    env.close(); free.close()


# ------------------------------------------------------------------ the debug (_trace) kernels are the product kernels
_MIX_KW = dict(mass_scale_range=(0.8, 1.2), friction_range=(0.25, 0.625))
_TIE_CASES = [
    # id, envs per wave the host can select for the group, envs (the BASELINE shard size of the group's workload), env keywords
    ("base-walk_ik", (4, 8, 16, 64), 4096, dict(task="walk", signal_type="ik")),
    ("base-gallop_ol", (4, 8, 16, 64), 8192, dict(task="gallop", signal_type="ol")),
    ("base-turn_ik_heightfield", (4, 8, 16, 64), 4096, dict(task="turn", signal_type="ik", terrain_type="random")),
    ("arm-walk_ik", (4, 8, 16), 4096, dict(task="walk", signal_type="ik", mark="arm")),
    ("mixed_base", (4, 8, 16), 4096, dict(task="mixed", signal_type="ik", **_MIX_KW)),
    ("mixed_arm", (4, 8, 16), 2048, dict(task="mixed", signal_type="ik", mark="arm", **_MIX_KW)),
    ("body-poses", (4, 8), 4096, dict(task="poses", signal_type="ik")),
    ("body-poses_rolled", (4, 8), 4096, dict(task="poses", signal_type="ik", base_roll=-0.74)),      # the link-link rows act
    ("arm_body-poses_rolled", (4,), 2048, dict(task="poses", signal_type="ik", mark="arm", base_roll=-0.74)),
]


_SEG_EXTRA_CASES = [
    # the observation-history ring (latency model): a later step of a segment reads what an earlier one stored, from other lanes
    ("base-walk_ik_latency", (4, 16), 4096, dict(task="walk", signal_type="ik", control_latency=0.02, pd_latency=0.003)),
    ("base-turn_ik_latency_noise", (4,), 2048, dict(task="turn", signal_type="ik", control_latency=0.02, observation_noise_stdev=(0.01, 0.01, 0.01, 0.01, 0.01))),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case,epw", [(c[0], e) for c in _TIE_CASES + _SEG_EXTRA_CASES for e in c[1]])
def test_segment_launch_is_bit_identical_to_single_steps(torch, case, epw, monkeypatch):
    """rex_step_segment (RexBatchEnv.step_segment) runs T consecutive env.step() calls in one launch -- the `_seg` instantiations of
    the step kernel, the step body inside a loop, an env's state in registers from step to step.  For every variant group x envs
    per wave, at the BASELINE shard size of the group's workload: 3 segments of 23 steps from reset, with in-launch auto-resets
    (episode cap 25), against the same 69 actions through step(): observation, reward, done and info['action'] of EVERY step and
    the state block after every segment BIT FOR BIT.  Also: a segment of one step, and the event trace set (step by step through
    the _trace kernels) give the same."""
    from rex_gym_amd import RexBatchEnv
    _, _, n, kw = next(c for c in _TIE_CASES + _SEG_EXTRA_CASES if c[0] == case)
    monkeypatch.setenv("REX_ENVS_PER_WAVE", str(epw))
    mk = lambda: RexBatchEnv(n, seed=17, auto_reset=True, max_episode_steps=25, check_actions=False, **kw)
    one, seg = mk(), mk()
    assert seg._L.rex_envs_per_wave(seg._h) == epw
    assert torch.equal(one.reset(), seg.reset()) and torch.equal(one.state, seg.state)
    lo = torch.as_tensor(np.minimum(one.action_space.low, one.action_space.high), device="cuda", dtype=torch.float32)
    hi = torch.as_tensor(np.maximum(one.action_space.low, one.action_space.high), device="cuda", dtype=torch.float32)
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    T, nm = 23, one._info["action"].shape[1]
    ended = 0
    for s in range(3):
        a = torch.rand((T, n, one.action_dim), device="cuda", generator=g) * (hi - lo) + lo
        cmd = torch.zeros((T, n, nm), device="cuda")
        so, sr, sd, si = seg.step_segment(a, motor_cmd=cmd)
        assert so.shape == (T, n, one.obs_dim) and sr.shape == (T, n) and sd.shape == (T, n) and sd.dtype == torch.bool and si["action"] is cmd
        for t in range(T):
            oo, orw, od, oi = one.step(a[t])
            for name, x, y in (("obs", oo, so[t]), ("reward", orw, sr[t]), ("done", od, sd[t]), ("action", oi["action"], cmd[t])):
                if not torch.equal(x, y):
                    bad = (x != y)
                    pytest.fail(f"{case} epw {epw}: segment {s} step {t}: {name} of the segment launch differs from the single steps' in "
                                f"{int(bad.sum())} words (first at {tuple(int(v) for v in torch.nonzero(bad)[0])})")
            ended += int(od.sum())
        assert torch.equal(one.state, seg.state), (case, epw, s)
    assert ended >= 2 * n                       # the comparison ran through in-launch resets in every env
    # a segment of one step; a segment under the event trace
    a = torch.rand((1, n, one.action_dim), device="cuda", generator=g) * (hi - lo) + lo
    so, sr, sd, _ = seg.step_segment(a)
    oo, orw, od, _ = one.step(a[0])
    assert torch.equal(so[0], oo) and torch.equal(sr[0], orw) and torch.equal(sd[0], od) and torch.equal(one.state, seg.state)
    seg.set_event_trace(True)
    a = torch.rand((5, n, one.action_dim), device="cuda", generator=g) * (hi - lo) + lo
    so, sr, sd, _ = seg.step_segment(a)
    for t in range(5):
        oo, orw, od, _ = one.step(a[t])
        assert torch.equal(so[t], oo) and torch.equal(sr[t], orw) and torch.equal(sd[t], od), (case, epw, "trace", t)
    assert torch.equal(one.state, seg.state)
    one.close(); seg.close()


def test_segment_launch_argument_checks(torch):
    from rex_gym_amd import RexBatchEnv
    env = RexBatchEnv(8, task="walk", signal_type="ik", seed=1)
    with pytest.raises(RuntimeError):
        env.step_segment(torch.zeros((2, 8, env.action_dim), device="cuda"))      # before reset
    env.reset()
    with pytest.raises(ValueError):
        env.step_segment(torch.zeros((8, env.action_dim), device="cuda"))         # not a segment
    with pytest.raises(ValueError):
        env.step_segment(torch.full((2, 8, env.action_dim), 9.0, device="cuda"))  # outside the Box (check_actions is on)
    with pytest.raises(ValueError):
        env.step_segment(torch.zeros((2, 8, env.action_dim), device="cuda"), out=(torch.zeros((2, 8, env.obs_dim), device="cuda"),
                                                                                   torch.zeros((2, 8), device="cuda"), torch.zeros((3, 8), dtype=torch.uint8, device="cuda")))
    o, r, d, info = env.step_segment(torch.zeros((4, 8, env.action_dim), device="cuda"))
    assert o.shape == (4, 8, env.obs_dim) and bool(torch.isfinite(o).all()) and info["action"] is None
    env.close()


@pytest.mark.parametrize("case,epw", [(c[0], e) for c in _TIE_CASES for e in c[1]])
def test_trace_kernels_are_bit_identical_to_the_product_kernels(torch, case, epw, monkeypatch):
    """rex_set_event_trace switches rex_step to the `_trace` instantiations -- separate code objects, compiled from the same
    step units with -DREX_TU_TRACE=1 (rex_gym_amd/build.py) -- and the event split of every long parity window is taken from
    them.  This ties the two together: for every variant group (base / arm / mixed_base / mixed_arm / body) x every envs-per-wave
    the host can select, at the BASELINE shard size of the group's workload, 70 steps from reset with in-launch auto-resets
    (episode cap 25: every env goes through two resets inside a launch, besides the falls), one env with the trace and one
    without, same seed and actions: state block, observation, reward, done and info['action'] are compared BIT FOR BIT after
    every step.  (tests/parity_window.py repeats the comparison over each 200-step oracle window.)"""
    from rex_gym_amd import RexBatchEnv
    _, _, n, kw = next(c for c in _TIE_CASES if c[0] == case)
    monkeypatch.setenv("REX_ENVS_PER_WAVE", str(epw))
    mk = lambda: RexBatchEnv(n, seed=17, auto_reset=True, max_episode_steps=25, **kw)
    prod, dbg = mk(), mk()
    assert prod._L.rex_envs_per_wave(prod._h) == epw and dbg._L.rex_envs_per_wave(dbg._h) == epw
    o0, o1 = prod.reset(), dbg.reset()
    tr = dbg.set_event_trace(True)
    assert torch.equal(o0, o1) and torch.equal(prod.state, dbg.state)
    lo = torch.as_tensor(np.minimum(prod.action_space.low, prod.action_space.high), device="cuda", dtype=torch.float32)
    hi = torch.as_tensor(np.maximum(prod.action_space.low, prod.action_space.high), device="cuda", dtype=torch.float32)
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    ended = 0
    for k in range(70):
        a = torch.rand((n, prod.action_dim), device="cuda", generator=g) * (hi - lo) + lo
        po, pr, pd, pi = prod.step(a)
        do, dr, dd, di = dbg.step(a)
        for name, x, y in (("state", prod.state, dbg.state), ("obs", po, do), ("reward", pr, dr), ("done", pd, dd), ("action", pi["action"], di["action"])):
            if not torch.equal(x, y):
                bad = (x != y)
                pytest.fail(f"{case} epw {epw}: step {k}: the _trace kernel's {name} differs from the product kernel's in {int(bad.sum())} words "
                            f"(first at {tuple(int(v) for v in torch.nonzero(bad)[0])}): a miscompile of one of the two -- diff their ISA")
        ended += int(pd.sum())
    assert ended >= 2 * n                       # the comparison ran through in-launch resets in every env
    assert int((tr[0] != 0).sum()) == n         # ... and the debug env really ran the trace instantiation
    prod.close(); dbg.close()

"""Unit tests of the oracle's physics restatement against analytic facts:
free fall, momentum/energy conservation with contacts out of reach, static equilibrium on the ground,
solver termination rule.  These pin the oracle where no PyBullet golden data exists."""
import numpy as np
import pytest

from orclib import Oracle, OracleEnv, default_config

STAND = np.array([0.0, -0.88643435, 1.30197369] * 4)


@pytest.fixture(scope="module")
def o():
    return Oracle(np.float64)


def _state(pos=(0, 0, 5.0), quat=(0, 0, 0, 1), lin=(0, 0, 0), ang=(0, 0, 0), q=STAND, qd=np.zeros(12)):
    return np.concatenate([pos, quat, lin, ang, q, qd]).astype(np.float64)


def test_free_fall_accelerations(o):
    a = o.forward_dynamics(_state(), np.zeros(12))
    np.testing.assert_allclose(a[:3], 0, atol=1e-12)
    np.testing.assert_allclose(a[3:6], [0, 0, -10.0], atol=1e-12)     # rex_gym_env.py:314
    np.testing.assert_allclose(a[6:], 0, atol=1e-10)


def test_internal_torques_do_not_move_the_centre_of_mass(o):
    o.set_damping(0, 0)
    try:
        rng = np.random.RandomState(3)
        st = _state(quat=rng.randn(4) / 2, q=STAND + 0.2 * rng.randn(12), qd=rng.randn(12))
        st[3:7] /= np.linalg.norm(st[3:7])
        p0 = o.energy_momentum(st)[1:]
        st2 = o.physics_substep(st, rng.uniform(-0.02, 0.02, 12), dt=1e-4, nsteps=500)  # small: stay below the 100 rad/s clamp
        p1 = o.energy_momentum(st2)[1:]
        np.testing.assert_allclose(p1[:2], p0[:2], atol=2e-4)                       # horizontal momentum
        np.testing.assert_allclose(p1[2], p0[2] - 4.52 * 10.0 * 0.05, atol=2e-4)    # m g t
    finally:
        o.set_damping(0.04, 0.04)


def test_energy_drift_is_first_order_in_dt(o):
    o.set_damping(0, 0)
    try:
        rng = np.random.RandomState(0)
        st = _state(quat=rng.randn(4), lin=rng.randn(3), ang=2 * rng.randn(3), q=STAND + 0.2 * rng.randn(12), qd=3 * rng.randn(12))
        st[3:7] /= np.linalg.norm(st[3:7])
        e0 = o.energy_momentum(st)[0]
        drift = []
        for dt in (1e-3, 1e-4):
            s = o.physics_substep(st, np.zeros(12), dt=dt, nsteps=int(round(0.1 / dt)))
            drift.append(abs(o.energy_momentum(s)[0] - e0))
        assert drift[0] < 0.05 and drift[1] < drift[0] / 5      # symplectic Euler: O(dt), converging
    finally:
        o.set_damping(0.04, 0.04)


def test_settled_stand_is_static_and_on_the_ground():
    env = OracleEnv(default_config("walk", "ik", 1, backwards=0, target_position=2.0))
    env.reset()
    st = env.get_state()[:, 0]
    assert 0.19 < st[2] < 0.215                       # base height: legs + toe radius
    assert np.abs(st[7:13]).max() < 0.05              # base at rest after the 600-substep reset motion
    assert np.abs(st[25:37]).max() < 0.1
    assert abs(st[6]) > 0.999                         # upright
    assert int(st[46]) == 0xFFF                       # all motors still enabled


def test_residual_threshold_only_trims_converged_sweeps():
    """Bullet's early exit (m_leastSquaresResidualThreshold = 1e-7) must not change the trajectory
    beyond the threshold's own scale."""
    a = OracleEnv(default_config("walk", "ik", 1, backwards=0, target_position=2.0, solver_residual_threshold=1e-7))
    b = OracleEnv(default_config("walk", "ik", 1, backwards=0, target_position=2.0, solver_residual_threshold=0.0))
    a.reset(); b.reset()
    for _ in range(60):
        a.step(np.array([[0.1, 0.0]])); b.step(np.array([[0.1, 0.0]]))
    sa, sb = a.get_state()[:, 0], b.get_state()[:, 0]
    np.testing.assert_allclose(sa[13:25], sb[13:25], atol=5e-4)
    np.testing.assert_allclose(sa[:3], sb[:3], atol=5e-4)


def test_walk_reward_terms_and_termination_flags():
    env = OracleEnv(default_config("walk", "ik", 2, backwards=0, target_position=2.0, max_episode_steps=7))
    env.reset()
    done_at = None
    for k in range(7):
        obs, rew, done, cmd = env.step(np.zeros((2, 2)))
        assert obs.shape == (2, 4) and cmd.shape == (2, 12) and np.isfinite(rew).all()
        if done.all() and done_at is None:
            done_at = k
    assert done_at == 6                                # LimitDuration semantics (wrappers.py:268-291)
    # standing start: forward term 0 (x <= 0.05), small negative drift/shake/energy terms
    assert -0.05 < rew[0] <= 0.0


# ---- mark='arm' (rex_arm.urdf: 19 bodies, 18 motors; model/mark_constants.py:14-27) ----
ARM_REST = np.array([-1.6, -1.6, 0.0, 0.0, 1.6, 0.0])     # mark_constants.py ARM_POSES['rest']


def _arm_state(rng=None, pos=(0, 0, 5.0)):
    q = np.concatenate([STAND, 0.5 * ARM_REST])
    q[17] = -0.5                                      # keep the last joint (bounds -1.3 .. 0.2) away from its limit rows
    qd = np.zeros(18)
    quat = np.array([0, 0, 0, 1.0])
    if rng is not None:
        q = q + 0.15 * rng.randn(18)
        qd = rng.randn(18)
        quat = rng.randn(4)
        quat /= np.linalg.norm(quat)
    return np.concatenate([pos, quat, np.zeros(6), q, qd]).astype(np.float64)


def test_arm_mark_free_fall_and_momentum():
    o = Oracle(np.float64, "arm")
    assert o.num_motors == 18 and o.state_words == 69
    a = o.forward_dynamics(_arm_state(), np.zeros(18))
    np.testing.assert_allclose(a[:3], 0, atol=1e-12)
    np.testing.assert_allclose(a[3:6], [0, 0, -10.0], atol=1e-12)
    np.testing.assert_allclose(a[6:], 0, atol=1e-10)
    o.set_damping(0, 0)
    try:
        rng = np.random.RandomState(5)
        st = _arm_state(rng)
        p0 = o.energy_momentum(st)[1:]
        st2 = o.physics_substep(st, rng.uniform(-0.02, 0.02, 18), dt=1e-4, nsteps=500)
        p1 = o.energy_momentum(st2)[1:]
        mass = 4.52 + 2.1                                                  # rex.urdf + the arm links of rex_arm.urdf
        np.testing.assert_allclose(p1[:2], p0[:2], atol=3e-4)
        np.testing.assert_allclose(p1[2], p0[2] - mass * 10.0 * 0.05, atol=3e-4)
    finally:
        o.set_damping(0.04, 0.04)


def test_arm_mark_reset_holds_the_arm_on_its_limits():
    """ARM_POSES['rest'] asks m1, m2, m5 for +-1.6 rad, beyond the +-1.5 rad URDF bounds: after the reset motion
    the three joints rest on their limit rows, the robot stands, and the env API is 18 / 22 wide."""
    env = OracleEnv(default_config("gallop", "ol", 2, mark=1), np.float64, "arm")
    assert env.obs_dim == 22
    obs = env.reset()
    st = env.get_state()
    q = st[13:31, 0]
    np.testing.assert_allclose(q[[12, 13, 16]], [-1.5, -1.5, 1.5], atol=5e-3)
    np.testing.assert_allclose(q[[14, 15, 17]], 0.0, atol=2e-2)
    assert 0.12 < st[2, 0] < 0.25 and np.abs(st[7:13, 0]).max() < 0.1     # standing, nearly at rest
    o, r, d, cmd = env.step(np.zeros((2, 4)))
    assert cmd.shape == (2, 18)
    np.testing.assert_allclose(cmd[:, 12:], np.tile(ARM_REST, (2, 1)))
    np.testing.assert_allclose(o[:, 4:], env.get_state()[13:31].T, atol=1e-12)   # gallop obs = rpy rates + 18 angles
    env.close()


def test_standup_env_starts_crouched_clear_of_the_ground_and_rises():
    """RexStandupEnv: the reset motion folds the legs onto the foot joint bound (INIT_POSES['rest_position'] asks for
    6 rad, the URDF allows 2.59); the base ends 66 mm above the ground -- the chassis box (35 mm half height) stays
    clear, which is why this env needs no body-ground rows -- and the 'brake' signal then lifts it past 0.15 m."""
    env = OracleEnv(default_config("standup", "ol", 1))
    env.reset()
    st = env.get_state()
    assert 0.055 < st[2, 0] < 0.08 and st[2, 0] - 0.035 > 0.02
    np.testing.assert_allclose(st[[15, 18, 21, 24], 0], 2.59, atol=5e-3)
    zs, rewards = [], []
    for k in range(40):
        o, r, d, cmd = env.step(np.array([[0.05]]))
        zs.append(env.get_state()[2, 0]); rewards.append(r[0])
        if k == 0:   # t = 0: stand * ((0.1 + a) / 1 + 1.5)
            np.testing.assert_allclose(cmd[0, :3], np.array([0.0, -0.88643435, 1.30197369]) * 1.65, atol=1e-12)
        assert not d[0]
    np.testing.assert_allclose(cmd[0, :3], [0.0, -0.88643435, 1.30197369], atol=1e-12)      # t > 0.1: the stand pose
    assert max(zs) > 0.15 and max(rewards) > 0.9                                               # within 0.1 of (0, 0, 0.21)
    env.close()

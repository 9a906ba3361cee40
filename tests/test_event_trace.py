"""The event trace of the oracle (orc_set_event_trace; the kernels fold the same words, rex_set_event_trace in include/rexsim.h --
compared on the GPU by tests/test_gpu_parity.py): per substep the toe points in reach, the heightfield facets under them, the
joint / arm bounds reached, per control step the controller's flags -- a chained hash per env; the sweep counts in a second word;
the events without the arm's bounds in a third.  CPU checks of the checker itself."""
import numpy as np

import orclib


def _run(task, signal, n, dtype, steps, mark="base", trace=True, terrain=False, seed=5, **kw):
    cfg = orclib.default_config(task, signal, n, seed=seed, mark=1 if mark == "arm" else 0, **kw)
    env = orclib.OracleEnv(cfg, dtype, mark)
    if terrain:
        from rex_gym_amd.terrain import random_terrain_pool
        env.set_terrain(*random_terrain_pool(4, 10))
    env.reset()
    t = env.set_event_trace(True) if trace else None
    rng = np.random.RandomState(1)
    b = 0.01 if task == "turn" else 0.4
    hist = []
    for _ in range(steps):
        env.step(rng.uniform(-b, b, (n, env.action_dim)).astype(np.float32))
        if t is not None:
            hist.append(t.copy())
    st = env.get_state()
    env.close()
    return (np.stack(hist) if hist else None), st


def test_trace_is_deterministic_chained_and_leaves_the_physics_alone():
    a, sa = _run("walk", "ik", 16, np.float64, 12)
    b, sb = _run("walk", "ik", 16, np.float64, 12)
    _, sc = _run("walk", "ik", 16, np.float64, 12, trace=False)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(sa, sc)                         # tracing changes nothing it traces
    assert a.shape == (12, 3, 16) and (a[0] != 0).all()
    assert (a[1:] != a[:-1]).all()                                # every step folds something in (flags / latches at least)
    # mark base has no arm bounds: words [0] and [2] (one folding step apart) separate the envs the same way
    assert ((a[-1, 0][:, None] == a[-1, 0][None, :]) == (a[-1, 2][:, None] == a[-1, 2][None, :])).all()


def test_float32_and_float64_take_the_same_decisions_on_the_plane_at_first():
    """On the flat plane, from the settled stand, the two precisions agree on every contact / bound / flag decision of the first
    control steps in nearly every env (the GPU tests measure 98-99 % over 200 steps); their sweep counts do not."""
    a, _ = _run("walk", "ik", 64, np.float32, 40)
    b, _ = _run("walk", "ik", 64, np.float64, 40)
    assert (a[-1, 0] == b[-1, 0]).mean() >= 0.9
    assert (a[-1, 1] == b[-1, 1]).mean() <= (a[-1, 0] == b[-1, 0]).mean()     # sweep counts part first: whether sweep 17 or 18 crosses the 1e-7 residual is a float32 question


def test_heightfield_facets_and_arm_bounds_enter_the_hash():
    plane, _ = _run("turn", "ik", 8, np.float64, 3)
    field, _ = _run("turn", "ik", 8, np.float64, 3, terrain=True)
    assert (plane[-1, 0] != field[-1, 0]).all()                   # another ground, other events
    arm, _ = _run("walk", "ik", 8, np.float64, 3, mark="arm")
    assert (arm[-1, 0] != arm[-1, 2]).all()                       # the arm sits on its bounds (ARM_POSES['rest'] beyond +-1.5): word [2] leaves them out

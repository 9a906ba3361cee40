#!/usr/bin/env python3
"""Golden ROLLOUTS: the reference's complete env / robot code running over the oracle's rigid-body step.

PyBullet is not installable here, so the physics half of the oracle cannot be pinned.  Everything
AROUND `stepSimulation`, however, is plain Python in the reference: RexGymEnv.__init__/reset/step, the five env
subclasses, Rex.Reset (drop, 100 + 500 settle substeps), Rex.Step (action repeat), ApplyAction (PD observation, motor
model, overheat protection, torque application), ReceiveObservation (history deque, latency blend), the getters,
_reward, _termination, _get_observation.  This script runs THAT code unmodified, constructors included, against a
stand-in for the pybullet client whose `stepSimulation` is the oracle's `orc_physics_substep` (one 1 ms rigid-body
step for the torques the reference just applied) and whose state getters read the same 37/49-word body state.  The
resulting (action -> observation, reward, done, motor command, body state) sequences are what the reference's
orchestration produces for this physics; the oracle's own orchestration (`orc_reset` / `orc_step`) must reproduce them
(tests/test_oracle_rollouts.py), which pins every row of the hot path except the contents of `stepSimulation`.

The stand-in implements only what a physics server would: joint table parsed from the reference's URDF, body state
get / set, torque accumulation, Bullet's quaternion <-> Euler / matrix conventions.  No reference logic lives in it.

Run in the build container:  PYTHONPATH=/root/reference python tests/golden/make_rollout_golden.py
"""
import json
import math
import os
import random
import sys
import types
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orclib                                                    # noqa: E402

np.math = math   # gait_planner.py:24 uses the numpy.math alias removed in numpy 2 (SURVEY.md 8c shim 1)
sys.path.insert(0, os.environ.get("REX_REFERENCE", "/root/reference"))


# ------------------------------------------------------------------ Bullet's math conventions (b3Quaternion / pybullet.c)
def euler_from_quaternion(q):
    x, y, z, w = (float(v) for v in q)
    sqx, sqy, sqz, squ = x * x, y * y, z * z, w * w
    sarg = -2.0 * (x * z - w * y)
    if sarg <= -0.99999:
        return (0.0, -0.5 * math.pi, 2 * math.atan2(x, -y))
    if sarg >= 0.99999:
        return (0.0, 0.5 * math.pi, 2 * math.atan2(-x, y))
    return (math.atan2(2 * (y * z + w * x), squ - sqx - sqy + sqz), math.asin(sarg),
            math.atan2(2 * (x * y + w * z), squ + sqx - sqy - sqz))


def quaternion_from_euler(e):
    r, p, y = (float(v) for v in e)
    cr, sr, cp, sp, cy, sy = math.cos(r / 2), math.sin(r / 2), math.cos(p / 2), math.sin(p / 2), math.cos(y / 2), math.sin(y / 2)
    return (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy)


def matrix_from_quaternion(q):
    x, y, z, w = (float(v) for v in q)
    d = x * x + y * y + z * z + w * w
    s = 2.0 / d
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz, xx, xy, xz, yy, yz, zz = w * xs, w * ys, w * zs, x * xs, x * ys, x * zs, y * ys, y * zs, z * zs
    return (1 - (yy + zz), xy - wz, xz + wy, xy + wz, 1 - (xx + zz), yz - wx, xz - wy, yz + wx, 1 - (xx + yy))


# ------------------------------------------------------------------ the physics-server stand-in
class OraclePhysicsClient:
    """What rex_gym talks to instead of pybullet: body state + one rigid-body step, both the oracle's."""
    COV_ENABLE_RENDERING = COV_ENABLE_PLANAR_REFLECTION = COV_ENABLE_GUI = 0
    TORQUE_CONTROL, VELOCITY_CONTROL, POSITION_CONTROL = 2, 0, 1
    URDF_USE_SELF_COLLISION = 8
    GUI, DIRECT, SHARED_MEMORY = 1, 2, 3
    unknown_calls = set()

    def __init__(self, connection_mode=None):
        self.robot = None
        self.dt, self.iterations = 1.0 / 240.0, 50
        self.calls = 0

    def __getattr__(self, name):                     # rendering / debug-visualiser calls: nothing to do
        OraclePhysicsClient.unknown_calls.add(name)
        return lambda *a, **k: None

    # -- world
    def resetSimulation(self): self.robot = None
    def setTimeStep(self, dt): self.dt = dt
    def setGravity(self, x, y, z): assert (x, y, z) == (0, 0, -10)
    def setPhysicsEngineParameter(self, numSolverIterations=None, **kw):
        if numSolverIterations is not None:
            self.iterations = int(numSolverIterations)

    def loadURDF(self, path, position=None, orientation=None, useFixedBase=False, flags=0):
        if not path.endswith(("rex.urdf", "rex_arm.urdf")):
            return 0                                 # plane.urdf: the ground of orc_physics_substep
        self.fixed_base = bool(useFixedBase)         # on_rack: the base hangs at INIT_RACK_POSITION (rex.py:269-287)
        mark = "arm" if path.endswith("rex_arm.urdf") else "base"
        self.orc = orclib.Oracle(np.float64, mark)
        nm = self.nm = self.orc.num_motors
        joints = [j for j in ET.parse(path).getroot().findall("joint")]
        self.joint_names = [j.get("name") for j in joints]
        import rex_gym.model.mark_constants as mc
        motor_names = mc.MARK_DETAILS["motors_names"][mark]
        self.motor_of_joint = [motor_names.index(n) if n in motor_names else -1 for n in self.joint_names]
        assert sorted(m for m in self.motor_of_joint if m >= 0) == list(range(nm))
        self.st = np.zeros(13 + 2 * nm)
        self.st[0:3], self.st[3:7] = position, orientation
        self.tau = np.zeros(nm)
        self.robot = 1
        return 1

    def getNumJoints(self, body): return len(self.joint_names)
    def getJointInfo(self, body, i): return (i, self.joint_names[i].encode())
    def getDynamicsInfo(self, body, link): return (1.0, 0.5, (1e-3, 1e-3, 1e-3))    # only read by the randomisers
    def changeDynamics(self, *a, **k): raise AssertionError("randomisers are not part of these rollouts")

    # -- body state
    def resetBasePositionAndOrientation(self, body, pos, orn):
        if body == self.robot:
            self.st[0:3], self.st[3:7] = pos, orn
    def resetBaseVelocity(self, body, lin, ang):
        self.st[7:10], self.st[10:13] = lin, ang
    def resetJointState(self, body, joint, value, targetVelocity=0):
        m = self.motor_of_joint[joint]
        self.st[13 + m], self.st[13 + self.nm + m] = value, targetVelocity
    def getBasePositionAndOrientation(self, body):
        return tuple(self.st[0:3].tolist()), tuple(self.st[3:7].tolist())
    def getBaseVelocity(self, body):
        return tuple(self.st[7:10].tolist()), tuple(self.st[10:13].tolist())
    def getJointState(self, body, joint):
        m = self.motor_of_joint[joint]
        return (float(self.st[13 + m]), float(self.st[13 + self.nm + m]), (0.0,) * 6, 0.0)

    # -- actuation and stepping
    def setJointMotorControl2(self, bodyIndex, jointIndex, controlMode, force=0, targetVelocity=0, **kw):
        if controlMode == self.TORQUE_CONTROL:
            self.tau[self.motor_of_joint[jointIndex]] = force
        else:
            assert controlMode == self.VELOCITY_CONTROL and force == 0   # "disable the default motor", rex.py:369-388
    def stepSimulation(self):
        self.st = self.orc.physics_substep(self.st, self.tau, dt=self.dt, iterations=self.iterations, nsteps=1,
                                           residual_threshold=float(np.float32(1e-7)),   # pybullet's default, never changed
                                           fixed_base=self.fixed_base)
        self.tau[:] = 0                                                  # external torques last one step
        self.calls += 1

    getEulerFromQuaternion = staticmethod(euler_from_quaternion)
    getQuaternionFromEuler = staticmethod(quaternion_from_euler)
    getMatrixFromQuaternion = staticmethod(matrix_from_quaternion)


# `--real-pybullet` (on a machine where `import pybullet, pybullet_data, gym` work; not possible in the build container, so
# this branch has never run): nothing is substituted, the very same scenarios run on the real engine and go to
# pybullet_rollout_golden.json -- the fixture that would pin the physics half (tests/test_oracle_rollouts.py picks it up
# and reports the oracle's trajectory error against it).
REAL = "--real-pybullet" in sys.argv
for name in ["tensorflow"] if REAL else ["pybullet", "pybullet_data", "gym", "gym.spaces", "gym.utils", "gym.utils.seeding", "tensorflow"]:
    sys.modules[name] = types.ModuleType(name)
_quiet = types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None, error=lambda *a, **k: None)
sys.modules["tensorflow"].logging = _quiet               # agents/tools/wrappers.py only logs through tensorflow
sys.modules["tensorflow"].compat = types.SimpleNamespace(v1=types.SimpleNamespace(logging=_quiet))
if not REAL:
    pb = sys.modules["pybullet"]
    pb.getMatrixFromQuaternion = matrix_from_quaternion      # rex_gym_env.py:527 calls the module, not the client
    pb.GUI, pb.DIRECT, pb.SHARED_MEMORY = 1, 2, 3
    sys.modules["gym"].Env = type("Env", (), {})
    class _Box:                                          # what rex_gym and its wrappers use of gym.spaces.Box
        def __init__(self, low, high, dtype=None):
            self.low, self.high = np.asarray(low), np.asarray(high)
            self.shape = self.low.shape
        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))
        def __eq__(self, other):
            return np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high)
    sys.modules["gym.spaces"].Box = _Box
    sys.modules["gym"].spaces = sys.modules["gym.spaces"]
    sys.modules["gym"].utils = sys.modules["gym.utils"]
    sys.modules["gym.utils"].seeding = sys.modules["gym.utils.seeding"]
    sys.modules["gym.utils.seeding"].np_random = lambda seed=None: (np.random.RandomState(seed), seed)
    sys.modules["pybullet_data"].getDataPath = lambda: "/nonexistent"

import rex_gym.model.gait_planner as gp                          # noqa: E402
import rex_gym.util.bullet_client as bullet_client               # noqa: E402
if not REAL:
    bullet_client.BulletClient = OraclePhysicsClient             # the one substitution: the physics server
from rex_gym.envs.gym import gallop_env, poses_env, standup_env, turn_env, walk_env   # noqa: E402
from rex_gym.agents.tools import wrappers                        # noqa: E402
from rex_gym.agents.tools.batch_env import BatchEnv              # noqa: E402

SCENARIOS = [
    # name, env class, constructor kwargs, oracle config kwargs, steps, episodes
    ("walk_ik", walk_env.RexWalkEnv, dict(target_position=1.0, backwards=False, signal_type="ik"),
     dict(task="walk", signal="ik", target_position=1.0, backwards=0), 60, 2),
    ("walk_ik_backwards", walk_env.RexWalkEnv, dict(target_position=1.5, backwards=True, signal_type="ik"),
     dict(task="walk", signal="ik", target_position=1.5, backwards=1), 40, 1),
    ("walk_ol", walk_env.RexWalkEnv, dict(target_position=0.75, backwards=False, signal_type="ol"),
     dict(task="walk", signal="ol", target_position=0.75, backwards=0), 60, 1),
    ("walk_ik_latency", walk_env.RexWalkEnv, dict(target_position=1.0, backwards=False, signal_type="ik",
                                                  control_latency=0.0125, pd_latency=0.003),
     dict(task="walk", signal="ik", target_position=1.0, backwards=0, control_latency=0.0125, pd_latency=0.003), 50, 2),
    ("walk_ik_arm", walk_env.RexWalkEnv, dict(target_position=1.0, backwards=False, signal_type="ik", mark="arm"),
     dict(task="walk", signal="ik", target_position=1.0, backwards=0, mark=1), 30, 1),
    ("gallop_ik", gallop_env.RexReactiveEnv, dict(target_position=2.0, signal_type="ik"),
     dict(task="gallop", signal="ik", target_position=2.0), 50, 2),
    ("gallop_ol", gallop_env.RexReactiveEnv, dict(target_position=2.0, signal_type="ol"),
     dict(task="gallop", signal="ol", target_position=2.0), 50, 1),
    ("turn_ik", turn_env.RexTurnEnv, dict(target_orient=2.0, init_orient=0.5, signal_type="ik"),
     dict(task="turn", signal="ik", target_orient=2.0, init_orient=0.5, orient_fixed=3), 60, 2),
    ("turn_ol", turn_env.RexTurnEnv, dict(target_orient=0.5, init_orient=2.5, signal_type="ol"),
     dict(task="turn", signal="ol", target_orient=0.5, init_orient=2.5, orient_fixed=3), 40, 1),
    ("poses_pitch", poses_env.RexPosesEnv, dict(base_y=0.0, base_z=0.0, base_roll=0.0, base_pitch=0.3, base_yaw=0.0),
     dict(task="poses", signal="ik", pose_index=3, pose_value=0.3), 60, 1),
    ("standup", standup_env.RexStandupEnv, dict(),
     dict(task="standup", signal="ol"), 60, 2),
    # mark 'arm' under the other observation / reset shapes: gallop returns 4 + 18 motor angles, standup starts crouched
    ("gallop_ik_arm", gallop_env.RexReactiveEnv, dict(target_position=2.0, signal_type="ik", mark="arm"),
     dict(task="gallop", signal="ik", target_position=2.0, mark=1), 25, 1),
    ("standup_arm", standup_env.RexStandupEnv, dict(mark="arm"),
     dict(task="standup", signal="ol", mark=1), 25, 1),
    # a forward walk until the robot falls (is_fallen -> done, rex_gym_env.py:490-499), then the next episode
    ("walk_ik_until_fallen", walk_env.RexWalkEnv, dict(target_position=2.0, backwards=False, signal_type="ik"),
     dict(task="walk", signal="ik", target_position=2.0, backwards=0), 400, 2),
    # three turn-IK envs behind the reference's BatchEnv, each inside the training stack; per-env actions, resets by index
    ("turn_ik_batch", turn_env.RexTurnEnv, dict(target_orient=2.0, init_orient=0.5, signal_type="ik", wrap=15, batch=3),
     dict(task="turn", signal="ik", target_orient=2.0, init_orient=0.5, orient_fixed=3, range_normalize=1, max_episode_steps=15), 50, 1),
    # the training stack of playground/trainer.py:47-52 around the env: LimitDuration, RangeNormalize, ClipAction,
    # ConvertTo32Bit; actions are drawn beyond [-1, 1] so that the clip is exercised
    ("walk_ik_wrapped", walk_env.RexWalkEnv, dict(target_position=1.0, backwards=False, signal_type="ik", wrap=25),
     dict(task="walk", signal="ik", target_position=1.0, backwards=0, range_normalize=1, max_episode_steps=25), 40, 3),
    ("gallop_ol_wrapped", gallop_env.RexReactiveEnv, dict(target_position=2.0, signal_type="ol", wrap=20),
     dict(task="gallop", signal="ol", target_position=2.0, range_normalize=1, max_episode_steps=20), 40, 2),
    # (appended last: a scenario's seed is 100 + its index)
    # the overheat shut-down (rex.py:601-608,617-623): on the debug rack, an open-loop command 3 rad below the front-left
    # foot joint's lower bound keeps that motor at its torque limit (> 2.45 N m) for more than 1 000 substeps; the robot
    # then switches it off until the next Reset, which re-enables it (rex.py:301-302).  Every event also records
    # Rex._motor_enabled_list and Rex._overheat_counter.
    ("walk_ol_on_rack_overheat", walk_env.RexWalkEnv, dict(target_position=1.0, backwards=False, signal_type="ol", on_rack=True,
                                                          action_bias=[0, -3.0, 0, 0, 0, 0, 0, 0]),
     dict(task="walk", signal="ol", target_position=1.0, backwards=0, on_rack=1), 215, 2),
]


def action_space(env):
    sp = env.action_space
    lo, hi = np.minimum(sp.low, sp.high), np.maximum(sp.low, sp.high)     # the gallop Box is inverted in the reference
    return lo, hi


def body(client):
    if not REAL:
        return client.st.tolist()
    rex = body.env.rex                                # the same 13 + 2 x motors words read back from the real engine
    pos, orn = client.getBasePositionAndOrientation(rex.quadruped)
    lin, ang = client.getBaseVelocity(rex.quadruped)
    js = [client.getJointState(rex.quadruped, j) for j in rex._motor_id_list]
    return list(pos) + list(orn) + list(lin) + list(ang) + [j[0] for j in js] + [j[1] for j in js]


ENV_OF_PLANNER = {}


def planner_clock():
    """time.time() as GaitPlanner.loop sees it: the simulation time of the env that owns the calling planner."""
    planner = sys._getframe(1).f_locals.get("self")
    env = ENV_OF_PLANNER.get(id(planner))
    return env.rex.GetTimeSinceReset() if env is not None and hasattr(env, "rex") else 0.0


def restart_phase(env):
    """The reference's planner phase runs on the wall clock (gait_planner.py:108-110) and simply continues across
    episodes; here the phase clock is simulation time since reset ("clock deviation"), so the phase
    bookkeeping restarts with it.  The arc angle `_alpha` is NOT touched: it carries over, as in the reference."""
    planner = getattr(env, "_gait_planner", None)
    if planner is not None:
        planner._phi, planner._last_time = 0.0, 0.0


def build_env(cls, kwargs):
    kwargs = dict(kwargs)
    wrap = kwargs.pop("wrap", None)
    kwargs.pop("batch", None)
    kwargs.pop("action_bias", None)
    env = cls(render=False, terrain_id="plane", **kwargs)   # the reference's constructor (terrain_id as its CLI passes it): hard reset, drop, settle
    if getattr(env, "_gait_planner", None) is not None:
        ENV_OF_PLANNER[id(env._gait_planner)] = env
    inner = env
    if wrap:                                          # playground/trainer.py:47-52
        env = wrappers.ConvertTo32Bit(wrappers.ClipAction(wrappers.RangeNormalize(wrappers.LimitDuration(env, wrap))))
    return env, inner


def run_batch(name, cls, kwargs, steps, seed):
    """N envs behind the reference's BatchEnv (agents/tools/batch_env.py), stepped with per-env actions; finished envs are
    reset by index the way simulate() does (agents/tools/simulate.py:70-81), and one env is reset early by index."""
    gp.time.time = planner_clock
    random.seed(seed)
    n = kwargs["batch"]
    pairs = [build_env(cls, kwargs) for _ in range(n)]
    batch = BatchEnv([e for e, _ in pairs], blocking=True)
    inners = [i for _, i in pairs]
    rng = np.random.RandomState(seed)
    events = []

    def snapshot():
        return [i._pybullet_client.st.tolist() for i in inners]

    def reset(indices):
        obs = batch.reset(None if indices is None else np.asarray(indices))
        for k in (range(n) if indices is None else indices):
            restart_phase(inners[k])
        events.append(dict(kind="reset", indices=None if indices is None else [int(k) for k in indices],
                           obs=np.asarray(obs, float).tolist(), body=snapshot()))

    reset(None)
    done = np.zeros(n, bool)
    for k in range(steps):
        if done.any():
            reset(np.nonzero(done)[0].tolist())
        if k == 7:
            reset([1])                                 # an env taken out of its episode early
        a = rng.uniform(-1.5, 1.5, (n, len(inners[0].action_space.low)))
        obs, reward, done, info = batch.step(a)
        events.append(dict(kind="step", action=a.tolist(), obs=np.asarray(obs, float).tolist(),
                           reward=np.asarray(reward, float).tolist(), done=np.asarray(done, bool).tolist(),
                           cmd=[np.asarray(i["action"], float).tolist() for i in info], body=snapshot()))
    return events


def run(name, cls, kwargs, steps, episodes, seed):
    if "batch" in kwargs:
        return run_batch(name, cls, kwargs, steps, seed)
    gp.time.time = planner_clock
    random.seed(seed)
    env, client_env = build_env(cls, kwargs)
    rng = np.random.RandomState(seed)
    lo, hi = action_space(client_env)
    client = client_env._pybullet_client
    body.env = client_env
    if "wrap" in kwargs:
        lo, hi = -1.5 * np.ones_like(lo), 1.5 * np.ones_like(hi)
    events = []
    bias = np.asarray(kwargs.get("action_bias", 0.0), float)

    def motors():   # the overheat bookkeeping of Rex.ApplyAction (rex.py:601-608)
        rex = client_env.rex
        return dict(motor_enabled=[bool(b) for b in rex._motor_enabled_list], overheat=[int(c) for c in rex._overheat_counter])

    for ep in range(episodes):
        obs = env.reset()                             # soft reset (hard_reset=False after the constructor)
        restart_phase(client_env)
        events.append(dict(kind="reset", obs=np.asarray(obs, float).tolist(), body=body(client), **motors()))
        for k in range(steps):
            a = rng.uniform(lo, hi) + bias
            obs, reward, done, info = env.step(a)
            events.append(dict(kind="step", action=a.tolist(), obs=np.asarray(obs, float).tolist(), reward=float(reward),
                               done=bool(done), cmd=np.asarray(info["action"], float).tolist(), body=body(client), **motors()))
            if done:
                break
    return events


def main():
    out = dict(description=("reference env/robot code on real pybullet" if REAL else
                            "reference env/robot code over the oracle's rigid-body step") + "; see make_rollout_golden.py",
               scenarios=[])
    for i, (name, cls, kwargs, ocfg, steps, episodes) in enumerate(SCENARIOS):
        events = run(name, cls, kwargs, steps, episodes, seed=100 + i)
        nstep = sum(e["kind"] == "step" for e in events)
        ndone = sum(int(np.sum(e.get("done", False))) for e in events)
        print(f"{name}: {nstep} steps, {ndone} done")
        out["scenarios"].append(dict(name=name, env_class=cls.__name__, env_kwargs=kwargs, oracle_config=ocfg, events=events))
    print("client calls with nothing to do:", sorted(OraclePhysicsClient.unknown_calls))
    path = os.path.join(HERE, "pybullet_rollout_golden.json" if REAL else "rollout_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()

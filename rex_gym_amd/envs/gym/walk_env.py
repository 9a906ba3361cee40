"""RexWalkEnv -- the reference's per-env Gym surface (rex_gym/envs/gym/walk_env.py:17-378) on top of the
batched HIP simulator: one env is simply a batch of one. numpy in / numpy out, like the original."""
import numpy as np

from ..batch_env import RexBatchEnv


class _Forward:
    def __init__(self, randomizer, env):
        self._r, self._env = randomizer, env
        if hasattr(randomizer, "randomize_step"):
            self.randomize_step = lambda _batch: randomizer.randomize_step(env)

    def randomize_env(self, _batch):
        self._r.randomize_env(self._env)


class _SingleEnv:
    _TASK = None
    metadata = {"render.modes": ["human", "rgb_array"], "video.frames_per_second": 66}

    def __init__(self, signal_type="ik", **kw):
        kw.setdefault("check_actions", False)   # RexGymEnv.step takes any action (rex_gym_env.py:369-414); BatchEnv is what checks
        self._batch = RexBatchEnv(1, task=self._TASK, signal_type=signal_type, **kw)
        self.action_space = self._batch.action_space
        self.observation_space = self._batch.observation_space
        self.control_time_step = self._batch.control_time_step
        self.rex = self._batch.rex     # what an env_randomizer's randomize_env(env) reaches for (envs/rex_knobs.py)
        # the hooks get THIS env object, as in the reference (rex_gym_env.py:345-346)
        self._batch._env_randomizers = [_Forward(r, self) for r in self._batch._env_randomizers]

    def add_env_randomizer(self, env_randomizer):
        self._batch.add_env_randomizer(_Forward(env_randomizer, self))

    def seed(self, seed=None):
        return [seed]

    def reset(self):
        return self._batch.reset()[0].cpu().numpy().astype(np.float64)

    def step(self, action):
        """-> (obs ndarray, float reward, bool done, {'action': 12-d motor command}) (rex_gym_env.py:369-414)"""
        a = np.asarray(action, dtype=np.float32).reshape(1, -1)
        obs, rew, done, info = self._batch.step(a)
        return (obs[0].cpu().numpy().astype(np.float64), float(rew[0].item()), bool(done[0].item()),
                {"action": info["action"][0].cpu().numpy().astype(np.float64)})

    def render(self, mode="rgb_array", close=False):
        return np.array([])

    def close(self):
        self._batch.close()


class RexWalkEnv(_SingleEnv):
    """Same constructor keywords as the reference (walk_env.py:31-50); GUI/debug/logging ones are accepted
    and ignored, terrain/mark other than plane/base raise."""
    _TASK = "walk"

    def __init__(self, debug=False, urdf_version=None, control_time_step=0.005, action_repeat=5, control_latency=0,
                 pd_latency=0, on_rack=False, motor_kp=1.0, motor_kd=0.02, render=False, num_steps_to_log=2000,
                 env_randomizer=None, log_path=None, target_position=None, backwards=None, signal_type="ik",
                 terrain_type="plane", terrain_id=None, mark="base", **kw):
        super().__init__(signal_type=signal_type, control_time_step=control_time_step, action_repeat=action_repeat,
                         motor_kp=motor_kp, motor_kd=motor_kd, control_latency=control_latency, pd_latency=pd_latency,
                         render=render, on_rack=on_rack, env_randomizer=env_randomizer, target_position=target_position,
                         backwards=backwards, terrain_type=terrain_type, mark=mark, **kw)

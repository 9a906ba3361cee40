"""RexReactiveEnv (gallop) -- rex_gym/envs/gym/gallop_env.py:25-381 on the batched HIP simulator."""
from .walk_env import _SingleEnv


class RexReactiveEnv(_SingleEnv):
    _TASK = "gallop"

    def __init__(self, debug=False, urdf_version=None, energy_weight=0.005, control_time_step=0.006, action_repeat=6,
                 control_latency=0.0, pd_latency=0.0, on_rack=False, motor_kp=1.0, motor_kd=0.02, render=False,
                 num_steps_to_log=2000, use_angle_in_observation=True, env_randomizer=None, log_path=None,
                 target_position=None, signal_type="ik", terrain_type="plane", terrain_id=None, mark="base", **kw):
        super().__init__(use_angle_in_observation=use_angle_in_observation, signal_type=signal_type, control_time_step=control_time_step, action_repeat=action_repeat,
                         motor_kp=motor_kp, motor_kd=motor_kd, control_latency=control_latency, pd_latency=pd_latency,
                         render=render, on_rack=on_rack, env_randomizer=env_randomizer, target_position=target_position, energy_weight=energy_weight,
                         terrain_type=terrain_type, mark=mark, **kw)

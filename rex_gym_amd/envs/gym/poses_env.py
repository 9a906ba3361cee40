"""RexPosesEnv -- rex_gym/envs/gym/poses_env.py:25-309 on the batched HIP simulator."""
from .walk_env import _SingleEnv


class RexPosesEnv(_SingleEnv):
    _TASK = "poses"

    def __init__(self, debug=False, urdf_version=None, control_time_step=0.006, action_repeat=6, control_latency=0,
                 pd_latency=0, on_rack=False, motor_kp=1.0, motor_kd=0.02, remove_default_joint_damping=False,
                 render=False, num_steps_to_log=1000, env_randomizer=None, log_path=None, base_y=None, base_z=None,
                 base_roll=None, base_pitch=None, base_yaw=None, signal_type="ik", terrain_type="plane",
                 terrain_id=None, mark="base", **kw):
        super().__init__(signal_type=signal_type, control_time_step=control_time_step, action_repeat=action_repeat,
                         motor_kp=motor_kp, motor_kd=motor_kd, control_latency=control_latency, pd_latency=pd_latency,
                         render=render, on_rack=on_rack, env_randomizer=env_randomizer, base_y=base_y, base_z=base_z,
                         base_roll=base_roll, base_pitch=base_pitch, base_yaw=base_yaw, terrain_type=terrain_type,
                         mark=mark, **kw)

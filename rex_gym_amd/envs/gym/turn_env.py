"""RexTurnEnv -- rex_gym/envs/gym/turn_env.py:19-418 on the batched HIP simulator (the companion cube the
reference drops 1 m away at the target heading, turn_env.py:369-378, is a visual marker and is not simulated)."""
from .walk_env import _SingleEnv


class RexTurnEnv(_SingleEnv):
    _TASK = "turn"

    def __init__(self, debug=False, urdf_version=None, control_time_step=0.005, action_repeat=5, control_latency=0,
                 pd_latency=0, on_rack=False, motor_kp=1.0, motor_kd=0.02, render=False, num_steps_to_log=1000,
                 env_randomizer=None, log_path=None, target_orient=None, init_orient=None, signal_type="ik",
                 terrain_type="plane", terrain_id=None, mark="base", **kw):
        super().__init__(signal_type=signal_type, control_time_step=control_time_step, action_repeat=action_repeat,
                         motor_kp=motor_kp, motor_kd=motor_kd, control_latency=control_latency, pd_latency=pd_latency,
                         render=render, on_rack=on_rack, env_randomizer=env_randomizer, target_orient=target_orient,
                         init_orient=init_orient, terrain_type=terrain_type, mark=mark, **kw)

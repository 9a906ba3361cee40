"""RexStandupEnv -- rex_gym/envs/gym/standup_env.py:19-209 on the batched HIP simulator.

The episode starts crouched: reset() runs the reset motion towards INIT_POSES['rest_position'] (standup_env.py:108-110;
its foot target of 6 rad is held at the URDF bound 2.59 by the joint-limit rows), then the env commands the 'stand' pose
with a short 'brake' overshoot (standup_env.py:113-120).

Known discrepancy with PyBullet: on recorded PyBullet episodes of this env the robot stays up after standing; here, at the
default toe friction 0.5, it tips over its front feet at control step ~100.  `RexStandupEnv(friction_range=(0.25, 0.25))` reproduces the record."""
from .walk_env import _SingleEnv


class RexStandupEnv(_SingleEnv):
    _TASK = "standup"

    def __init__(self, debug=False, urdf_version=None, control_time_step=0.005, action_repeat=5, control_latency=0,
                 pd_latency=0, on_rack=False, motor_kp=1.0, motor_kd=0.02, remove_default_joint_damping=False,
                 render=False, num_steps_to_log=1000, env_randomizer=None, log_path=None, signal_type="ol",
                 terrain_type="plane", terrain_id=None, mark="base", **kw):
        super().__init__(signal_type=signal_type, control_time_step=control_time_step, action_repeat=action_repeat,
                         motor_kp=motor_kp, motor_kd=motor_kd, control_latency=control_latency, pd_latency=pd_latency,
                         render=render, on_rack=on_rack, env_randomizer=env_randomizer, terrain_type=terrain_type, mark=mark, **kw)

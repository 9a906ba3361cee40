"""A batch whose envs run different tasks (BASELINE.json configs[4]: "mixed gaits + mass/friction domain
randomisation").  The reference gets a mixed batch by handing `BatchEnv` a list of different env objects
(`agents/tools/batch_env.py:21-44` only asks that they share the spaces).  Here it is ONE launch per step
(`REX_TASK_MIXED`, include/rexsim.h): every env runs one task of the mix for its whole life, drawn per env from its
Philox stream (global env index), with that task's own action_repeat, solver sweep cap, action bounds and reward; mass
and friction are drawn afresh on every reset inside the launch.  Env g of the mixed batch equals env g of a
single-task batch of its task with the same seed to fp32 round-off (tests/test_gpu_parity.py; bit for bit in the oracle).

Rows are as wide as the widest task of the mix: narrower tasks ignore the tail of their action row and leave the tail
of their observation row zero -- the convention a learner needs anyway to feed one policy.
"""
from .batch_env import RexBatchEnv


class RexMixedBatchEnv(RexBatchEnv):
    def __init__(self, num_envs, tasks=(("walk", "ik"), ("gallop", "ik"), ("turn", "ik")), mass_scale_range=None,
                 friction_range=None, **kwargs):
        """tasks: sequence of (task, signal_type) -- one signal type for the whole batch; mass_scale_range /
        friction_range: (lo, hi) of the per-reset uniform draws, e.g. (0.8, 1.2) and (0.25, 0.625) (foot friction 0.5 x
        U(0.5, 1.25)); None leaves the URDF values."""
        names = [t if isinstance(t, str) else t[0] for t in tasks]
        signals = {"ik" if isinstance(t, str) else t[1] for t in tasks}
        if len(signals) != 1:
            raise ValueError("a mixed batch runs one signal type")
        super().__init__(num_envs, task="mixed", signal_type=signals.pop(), tasks=names, mass_scale_range=mass_scale_range,
                         friction_range=friction_range, **kwargs)

    def task_of(self, g):
        """(task name, signal) of env g."""
        inv = {v: k for k, v in __import__("rex_gym_amd")._lib.TASKS.items()}
        return inv[int(self.task_ids()[g])], self.signal_type

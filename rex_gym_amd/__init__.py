"""rex_gym_amd -- MI355X-native batched Rex quadruped simulator + gait engine.

Drop-in for ONE hot path of nicrusso7/rex-gym: `rex_gym.envs` step()/reset() (PyBullet
stepSimulation + Bezier-IK controller + motor model), vectorised one-env-per-lane in hand-written
gfx950 HIP kernels behind the C ABI of include/rexsim.h.  See DESIGN.md.
"""
__version__ = "0.1.0"

from .envs.batch_env import RexBatchEnv  # noqa: F401
from .envs.mixed_env import RexMixedBatchEnv  # noqa: F401
from .envs.spaces import Box  # noqa: F401
from .envs.registry import ENV_IDS, make  # noqa: F401

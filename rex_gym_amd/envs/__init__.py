from .batch_env import RexBatchEnv  # noqa: F401
from .spaces import Box  # noqa: F401

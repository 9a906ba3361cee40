from .batch_env import RexBatchEnv  # noqa: F401
from .spaces import Box  # noqa: F401
from .mixed_env import RexMixedBatchEnv  # noqa: F401
from .registry import ENV_IDS, make  # noqa: F401

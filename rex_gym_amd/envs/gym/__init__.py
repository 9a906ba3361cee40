from .walk_env import RexWalkEnv  # noqa: F401
from .gallop_env import RexReactiveEnv  # noqa: F401

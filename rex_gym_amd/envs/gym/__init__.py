from .walk_env import RexWalkEnv  # noqa: F401
from .gallop_env import RexReactiveEnv  # noqa: F401
from .turn_env import RexTurnEnv  # noqa: F401
from .poses_env import RexPosesEnv  # noqa: F401
from .standup_env import RexStandupEnv  # noqa: F401

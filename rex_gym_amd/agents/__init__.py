"""PPO learner that consumes RexBatchEnv's device tensors directly."""
from .ppo import PPOAgent, PPOConfig, StreamingNormalize, ForwardGaussianPolicy, RecurrentGaussianPolicy, train, train_segments  # noqa: F401
from .fused_actor import FusedActor  # noqa: F401
from .policy_player import SimplePPOPolicy, play, play_segments, save_policy  # noqa: F401
from .tf_checkpoint import Checkpoint, CheckpointError, write_checkpoint  # noqa: F401

"""PPO learner that consumes RexBatchEnv's device tensors directly."""
from .ppo import PPOAgent, PPOConfig, StreamingNormalize, ForwardGaussianPolicy, train  # noqa: F401

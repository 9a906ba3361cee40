"""The Gym ids the reference registers (rex_gym/playground/__init__.py:19-58) and `make()` for them.

`gym.make(id, **kwargs)` in the reference builds ONE env and wraps it in gym's TimeLimit(max_episode_steps)
(playground/trainer.py:47); here the same id builds a batch of `num_envs` robots on the GPU with the step limit folded
into the launch (`max_episode_steps`).  `RexGo-v0` is registered by the reference but its module (`go_env.py`) does not
exist there; asking for it fails here as it does there.
"""
from .batch_env import RexBatchEnv

#        id                 task       max_episode_steps   (reward_threshold 5.0 for all; unused by the agents)
ENV_IDS = {
    "RexGalloping-v0": ("gallop", 1000),
    "RexWalk-v0": ("walk", 2500),
    "RexTurn-v0": ("turn", 1000),
    "RexStandup-v0": ("standup", 400),
    "RexPoses-v0": ("poses", 400),
}
# gym.make(id) without signal_type uses the env constructor's own default (walk_env.py:46, gallop_env.py:60,
# turn_env.py:46, standup_env.py:45, poses_env.py:55)
DEFAULT_SIGNAL = {"gallop": "ik", "walk": "ik", "turn": "ik", "standup": "ol", "poses": "ik"}


def make(env_id, num_envs=1, **kwargs):
    """-> RexBatchEnv for a registered id; kwargs are the env constructor's (signal_type, target_position, mark ...)."""
    if env_id == "RexGo-v0":
        raise ModuleNotFoundError("RexGo-v0 is registered by rex_gym but rex_gym.envs.gym.go_env does not exist")
    if env_id not in ENV_IDS:
        raise KeyError(f"unknown env id {env_id!r}; registered: {sorted(ENV_IDS)}")
    task, limit = ENV_IDS[env_id]
    kwargs.setdefault("signal_type", DEFAULT_SIGNAL[task])
    kwargs.setdefault("max_episode_steps", limit)
    return RexBatchEnv(num_envs, task=task, **kwargs)

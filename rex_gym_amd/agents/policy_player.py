"""Run a policy trained by the reference (or by rex_gym_amd.agents.ppo) on the batched HIP env.

Mirrors `SimplePPOPolicy` (agents/ppo/simple_ppo_agent.py:9-88) and `PolicyPlayer.play`
(playground/policy_player.py:22-56): the observation is mapped onto [-1, 1] with the env's observation bounds, passed
through the checkpoint's streaming observation filter (centre, scale, clip 5), the policy network's MEAN action is taken
(no sampling) and mapped back onto the env's action bounds.  The reference restores the TensorFlow-1 checkpoint with
tf.train.Saver; here `tf_checkpoint.Checkpoint` reads the same files and the weights are loaded into the torch
`ForwardGaussianPolicy`, so the shipped policies (rex_gym/policies/<env>/<signal>/model.ckpt-N) run unchanged -- on
N envs at a time, observations and actions staying on the GPU.

    python -m rex_gym_amd.agents.policy_player --env walk --signal-type ik \
        --checkpoint /path/to/rex_gym/policies/walk/ik/model.ckpt-2000000 --num-envs 1024
"""
import argparse
import json

import numpy as np
import torch

from .ppo import ForwardGaussianPolicy, PPOConfig, StreamingNormalize
from .tf_checkpoint import Checkpoint, CheckpointError, write_checkpoint

SCOPE = "network/rnn"          # simple_ppo_agent.py:47 builds the cell under variable_scope("network/rnn")


def _layer_names(ckpt, branch):
    """fully_connected, fully_connected_1, ... of one branch in creation order (tf.contrib.layers naming)."""
    names, k = [], 0
    while True:
        base = f"{SCOPE}/{branch}/fully_connected" + (f"_{k}" if k else "")
        if base + "/weights" not in ckpt.entries:
            return names
        names.append(base)
        k += 1


def restore_network(ckpt, device="cpu"):
    """ForwardGaussianPolicy with the checkpoint's layer sizes and weights (networks.py:69-112).

    TF's fully_connected stores weights [in, out]; torch.nn.Linear holds [out, in]."""
    if isinstance(ckpt, str):
        ckpt = Checkpoint(ckpt)
    pol, val = _layer_names(ckpt, "policy"), _layer_names(ckpt, "value")
    if len(pol) < 1 or len(val) < 1 or f"{SCOPE}/policy/logstd" not in ckpt.entries:
        raise CheckpointError(f"{ckpt.prefix}: no {SCOPE}/policy|value/fully_connected* variables -- not a "
                              "ForwardGaussianPolicy checkpoint")
    obs_dim = ckpt.shape(pol[0] + "/weights")[0]
    action_dim = ckpt.shape(pol[-1] + "/weights")[1]
    cfg = PPOConfig(policy_layers=tuple(ckpt.shape(n + "/weights")[1] for n in pol[:-1]),
                    value_layers=tuple(ckpt.shape(n + "/weights")[1] for n in val[:-1]))
    net = ForwardGaussianPolicy(obs_dim, action_dim, cfg)

    def put(linear, base):
        w, b = ckpt.tensor(base + "/weights"), ckpt.tensor(base + "/biases")
        if tuple(linear.weight.shape) != (w.shape[1], w.shape[0]):
            raise CheckpointError(f"{base}: shape {w.shape} does not fit {tuple(linear.weight.shape)}")
        linear.weight.copy_(torch.from_numpy(np.ascontiguousarray(w.T)))
        linear.bias.copy_(torch.from_numpy(b))

    with torch.no_grad():
        for lin, base in zip([m for m in net.policy if isinstance(m, torch.nn.Linear)], pol[:-1]):
            put(lin, base)
        put(net.mean, pol[-1])
        for lin, base in zip([m for m in net.value if isinstance(m, torch.nn.Linear)], val[:-1]):
            put(lin, base)
        put(net.value_out, val[-1])
        net.logstd.copy_(torch.from_numpy(ckpt.tensor(f"{SCOPE}/policy/logstd").reshape(-1)))
    return net.to(device)


def restore_normalizer(ckpt, name, clip, device="cpu"):
    """StreamingNormalize from `<name>/Variable{,_1,_2}` = count, mean, var_sum (normalize.py:38-41)."""
    if isinstance(ckpt, str):
        ckpt = Checkpoint(ckpt)
    mean = ckpt.tensor(f"{name}/Variable_1")
    f = StreamingNormalize(mean.shape, center=True, scale=True, clip=clip, device=device)
    f.count = int(ckpt.tensor(f"{name}/Variable").reshape(-1)[0])
    f.mean = torch.as_tensor(mean, dtype=torch.float32, device=device)
    f.var_sum = torch.as_tensor(ckpt.tensor(f"{name}/Variable_2"), dtype=torch.float32, device=device)
    return f


def save_policy(prefix, network, observ_filter, global_step=None):
    """The inverse of restore_network / restore_normalizer: write `network` (a ForwardGaussianPolicy) and its observation filter
    as a TensorFlow-1 checkpoint `<prefix>.index` / `.data-00000-of-00001` under the variable names of the reference's graph,
    plus the `checkpoint` state file tf.train.get_checkpoint_state reads (agents/scripts/utility.py:138-142).  These are the
    variables SimplePPOPolicy's Saver restores (simple_ppo_agent.py:23-27,46-63: `normalize_observ/*` and `network/rnn/*`; its
    `temporary/*` state is excluded), so a policy trained with rex_gym_amd.agents.ppo plays under the reference's
    `rex-gym policy` next to the config.yaml of its env.  (Resuming the reference's TRAINING from it needs the optimizer slots
    and episode memory of that graph as well, which are not state of this learner: not written.)"""
    import os
    tensors = {}

    def put(linear, base):
        tensors[base + "/weights"] = np.ascontiguousarray(linear.weight.detach().cpu().numpy().astype(np.float32).T)   # TF: [in, out]
        tensors[base + "/biases"] = linear.bias.detach().cpu().numpy().astype(np.float32)

    def branch(layers, last, name):
        for k, lin in enumerate(layers + [last]):
            put(lin, f"{SCOPE}/{name}/fully_connected" + (f"_{k}" if k else ""))

    branch([m for m in network.policy if isinstance(m, torch.nn.Linear)], network.mean, "policy")
    branch([m for m in network.value if isinstance(m, torch.nn.Linear)], network.value_out, "value")
    tensors[f"{SCOPE}/policy/logstd"] = network.logstd.detach().cpu().numpy().astype(np.float32).reshape(-1)
    tensors["normalize_observ/Variable"] = np.asarray(int(observ_filter.count), np.int32)            # normalize.py:38-41
    tensors["normalize_observ/Variable_1"] = observ_filter.mean.detach().cpu().numpy().astype(np.float32)
    tensors["normalize_observ/Variable_2"] = observ_filter.var_sum.detach().cpu().numpy().astype(np.float32)
    if global_step is not None:
        tensors["global_step"] = np.asarray(int(global_step), np.int32)
    write_checkpoint(prefix, tensors)
    with open(os.path.join(os.path.dirname(prefix) or ".", "checkpoint"), "w") as f:
        base = os.path.basename(prefix)
        f.write(f'model_checkpoint_path: "{base}"\nall_model_checkpoint_paths: "{base}"\n')
    return prefix


class SimplePPOPolicy:
    """simple_ppo_agent.py:9-88 for a batch of envs: get_action(observation[N, O]) -> action[N, A], on `device`."""

    def __init__(self, env, checkpoint, device=None):
        self.env = env
        self.device = torch.device(device if device is not None else getattr(env, "device", "cpu"))
        ckpt = Checkpoint(checkpoint) if isinstance(checkpoint, str) else checkpoint
        self.network = restore_network(ckpt, self.device).eval()
        self._observ_filter = restore_normalizer(ckpt, "normalize_observ", clip=5.0, device=self.device)  # :23-27
        t = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=self.device)   # noqa: E731
        self._olo, self._ohi = t(env.observation_space.low), t(env.observation_space.high)
        self._alo, self._ahi = t(env.action_space.low), t(env.action_space.high)
        obs_dim, act_dim = self.network.policy[0].in_features, self.network.mean.out_features
        if self._olo.numel() != obs_dim or self._alo.numel() != act_dim:
            raise CheckpointError(f"checkpoint maps {obs_dim} observations to {act_dim} actions; the env has "
                                  f"{self._olo.numel()} and {self._alo.numel()}")

    @torch.no_grad()
    def get_action(self, observation):
        observ = torch.as_tensor(observation, dtype=torch.float32, device=self.device)
        observ = 2 * (observ - self._olo) / (self._ohi - self._olo) - 1            # _normalize_observ, :83-88
        mean, _, _ = self.network(self._observ_filter.transform(observ))          # mean action, :53-60
        return (mean + 1) / 2 * (self._ahi - self._alo) + self._alo               # _denormalize_action, :77-81


@torch.no_grad()
def play(env, policy, max_steps=2500):
    """PolicyPlayer.play (policy_player.py:44-56) on every env of the batch at once: one episode each, until done.

    Returns per-env (sum of rewards, episode length, whether the env ended by itself before max_steps)."""
    observ = env.reset()
    n = observ.shape[0]
    total = torch.zeros(n, device=observ.device)
    length = torch.zeros(n, dtype=torch.int32, device=observ.device)
    alive = torch.ones(n, dtype=torch.bool, device=observ.device)
    for _ in range(max_steps):
        action = policy.get_action(observ)
        observ, reward, done, _ = env.step(action)
        total += torch.where(alive, reward.to(total.dtype), torch.zeros_like(total))
        length += alive.to(length.dtype)
        alive &= ~done.bool()
        if not bool(alive.any()):
            break
    return total, length, ~alive


@torch.no_grad()
def play_segments(env, policy, max_steps=2500, segment=100):
    """play() with the ACTOR INSIDE THE LAUNCH: the checkpoint's network and observation filter go to the library once
    (FusedActor -> rex_set_policy, evaluation mode: the mean action) and the episodes run `segment` closed-loop steps per launch
    (RexBatchEnv.step_segment_policy).  `env` must fold the wrapper stack the reference plays its policies through
    (RexBatchEnv(range_normalize=True, auto_reset=True): `_normalize_observ` / `_denormalize_action` of simple_ppo_agent.py:77-88 are
    RangeNormalize's formulas, applied inside the launch) -- `policy` is a SimplePPOPolicy built on that env.  One episode per env, as
    play(): what an env does after its first `done` (it is reset inside the launch) is masked out.  One host synchronisation per segment.
    Returns per-env (sum of rewards, episode length, whether the env ended by itself before max_steps)."""
    from .fused_actor import FusedActor
    if not env.config.range_normalize or not env.config.auto_reset:
        raise ValueError("play_segments: create the env with range_normalize=True, auto_reset=True")
    FusedActor(env, policy.network, policy._observ_filter, sample=False)
    observ = env.reset()
    n, dev = observ.shape[0], observ.device
    total = torch.zeros(n, device=dev)
    length = torch.zeros(n, dtype=torch.int32, device=dev)
    alive = torch.ones(n, dtype=torch.bool, device=dev)
    steps = 0
    while steps < max_steps:
        T = min(int(segment), max_steps - steps)
        obs, reward, done, _ = env.step_segment_policy(T, observ)
        ended_before = torch.cat([torch.zeros((1, n), dtype=torch.bool, device=dev), done[:-1].cumsum(0) > 0], 0)     # [T, n]: a done at an earlier step
        counted = alive[None, :] & ~ended_before
        total += (reward * counted).sum(0)
        length += counted.sum(0).to(length.dtype)
        alive &= ~done.any(0)
        observ = obs[-1].clone()
        steps += T
        if not bool(alive.any()):
            break
    return total, length, ~alive


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--env", default="walk", choices=["walk", "gallop", "turn", "poses", "standup"])
    p.add_argument("--signal-type", default="ik", choices=["ik", "ol"])
    p.add_argument("--checkpoint", required=True, help="checkpoint prefix, e.g. .../model.ckpt-2000000")
    p.add_argument("--num-envs", type=int, default=64)
    p.add_argument("--max-steps", type=int, default=2500)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--fused", action="store_true", help="the actor inside the launch: one launch per --segment steps (play_segments)")
    p.add_argument("--segment", type=int, default=100)
    args = p.parse_args(argv)
    from ..envs import RexBatchEnv
    # (PolicyPlayer.play steps the bare env, policy_player.py:44-56: BatchEnv's per-step Box test -- a host sync per step here -- is not in that loop)
    env = RexBatchEnv(args.num_envs, task=args.env, signal_type=args.signal_type, seed=args.seed, check_actions=False,
                      range_normalize=args.fused, auto_reset=args.fused)
    policy = SimplePPOPolicy(env, args.checkpoint)
    total, length, ended = play_segments(env, policy, args.max_steps, args.segment) if args.fused else play(env, policy, args.max_steps)
    x = env.state[0].float()
    print(json.dumps(dict(env=args.env, signal=args.signal_type, num_envs=args.num_envs,
                          mean_return=float(total.mean()), mean_length=float(length.float().mean()),
                          ended_before_limit=float(ended.float().mean()), mean_final_x=float(x.mean()))))
    env.close()


if __name__ == "__main__":
    main()

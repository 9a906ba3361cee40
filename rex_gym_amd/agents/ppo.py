"""KL-penalty PPO of the reference (rex_gym/agents/ppo/algorithm.py, a TF1 graph) restated in PyTorch on the device.

The reference learner pulls numpy arrays out of N environment processes every step; here observations, rewards and
done flags are the device tensors RexBatchEnv returns, the episode buffers live in HBM and nothing crosses PCIe.

What is restated, piece by piece:
  StreamingNormalize     agents/ppo/normalize.py:18-153   (Welford mean / variance, centre, scale, clip)
  discounted_return ...  agents/ppo/utility.py:71-143     (Monte-Carlo return, TD-lambda return, diagonal normals)
  ForwardGaussianPolicy  agents/scripts/networks.py:69-112, scripts/utility.py:100-117 (initialisers)
  RecurrentGaussianPolicy  agents/scripts/networks.py:113-159 (GRU last policy layer; no shipped config selects it)
  PPOAgent               agents/ppo/algorithm.py:26-531   (episode buffers, memory of `update_every` episodes, policy /
                                                           value updates, KL penalty adaptation)
  train                  agents/tools/simulate.py:15-131  (reset done envs, act, step, hand transitions to the agent)
Defaults are agents/scripts/configs.py:24-52.  Multi-GPU (PPOAgent(sync_gradients=True)): every rank keeps the
episodes of its own shard in its own memory; the update is collective -- decided by an all-reduce once every rank's
memory is full, gradients and the KL statistic averaged over the ranks, the observation / reward filters merged
(torch.distributed = RCCL over xGMI).  `sharding.gather_rollout` is the hand-off for a CENTRAL learner (one rank
training on everybody's rollouts) and is not used by this gradient-averaged loop.
"""
import math
from dataclasses import dataclass

import torch


@dataclass
class PPOConfig:
    """agents/scripts/configs.py:24-52 (default()) + the per-task max_length."""
    policy_layers: tuple = (200, 100)
    value_layers: tuple = (200, 100)
    init_mean_factor: float = 0.05
    init_logstd: float = -1.0
    update_every: int = 25
    update_epochs_policy: int = 50
    update_epochs_value: int = 50
    policy_lr: float = 1e-4
    value_lr: float = 3e-4
    discount: float = 0.985
    gae_lambda: float = None
    kl_target: float = 1e-2
    kl_cutoff_factor: float = 2.0
    kl_cutoff_coef: float = 1000.0
    kl_init_penalty: float = 1.0
    max_length: int = 2000
    network: str = "forward"        # "forward": ForwardGaussianPolicy (every shipped config); "recurrent": RecurrentGaussianPolicy


class StreamingNormalize:
    """normalize.py:18-153.  Running mean / variance over everything seen by update(); transform() centres, scales by
    std + 1e-8 (std = sqrt(var_sum / (count - 1) + 1e-4), 1 until two samples were seen) and clips."""

    def __init__(self, shape, center=True, scale=True, clip=10.0, device="cpu"):
        self.center, self.scale, self.clip = center, scale, clip
        self.count = 0
        self.mean = torch.zeros(shape, device=device)
        self.var_sum = torch.zeros(shape, device=device)

    def std(self):
        return torch.sqrt(self.var_sum / (self.count - 1) + 1e-4)

    def transform(self, value):
        if self.center:
            value = value - self.mean
        if self.scale:
            value = value / (self.std() + 1e-8 if self.count > 1 else torch.ones_like(self.var_sum))
        if self.clip:
            value = value.clamp(-self.clip, self.clip)
        return value

    def update(self, value):
        """value: a batch [B, *shape] (normalize.py:70-95: a batch update of Welford's recurrence)."""
        first = self.count == 0
        self.count += value.shape[0]
        mean_delta = (value - self.mean).sum(0)
        new_mean = self.mean + mean_delta / self.count
        if self.count <= 1:
            new_mean = value[0].clone()
        self.var_sum = self.var_sum + ((value - self.mean) * (value - new_mean)).sum(0)
        self.mean = new_mean
        return first

    def merge_ranks(self):
        """Multi-rank learner: replace this rank's statistics by those of the union of what all ranks have seen since the
        last merge plus the common statistics of that merge (Chan's parallel mean / variance combination, one all-reduce).
        After the call every rank holds the same filter."""
        import torch.distributed as dist
        base = getattr(self, "_base", None)
        if base is None:
            base = (0, torch.zeros_like(self.mean), torch.zeros_like(self.var_sum))
        n0, m0, v0 = base
        # this rank's own samples since the last merge: (count, mean, var_sum) "minus" the common base
        n1 = self.count - n0
        if n1 > 0:
            m1 = (self.mean * self.count - m0 * n0) / n1
            v1 = self.var_sum - v0 - (m0 - self.mean) ** 2 * n0 - (m1 - self.mean) ** 2 * n1
        else:
            m1, v1 = torch.zeros_like(self.mean), torch.zeros_like(self.var_sum)
        pack = torch.cat([torch.tensor([float(n1)], device=self.mean.device, dtype=torch.float64),
                          (m1.double() * n1).reshape(-1), (v1.double() + m1.double() ** 2 * n1).reshape(-1)])
        dist.all_reduce(pack)
        k = self.mean.numel()
        n = float(pack[0]) + n0
        s1 = pack[1:1 + k].reshape(self.mean.shape) + m0.double() * n0
        s2 = pack[1 + k:].reshape(self.mean.shape) + v0.double() + m0.double() ** 2 * n0
        if n > 0:
            mean = s1 / n
            self.mean = mean.to(self.mean.dtype)
            self.var_sum = (s2 - mean ** 2 * n).clamp_min(0).to(self.var_sum.dtype)
            self.count = int(round(n))
        self._base = (self.count, self.mean.clone(), self.var_sum.clone())


# ---- utility.py:71-143 on [B, T] sequences with a length per row -------------------------------------------------
def _mask(length, steps, dtype=torch.float32):
    return (torch.arange(steps, device=length.device)[None, :] < length[:, None]).to(dtype)


def discounted_return(reward, length, discount):
    """utility.py:71-81: Monte-Carlo return of every step, zero beyond the episode's length."""
    r = reward * _mask(length, reward.shape[1], reward.dtype)
    out = torch.zeros_like(r)
    agg = torch.zeros_like(r[:, 0])
    for t in range(r.shape[1] - 1, -1, -1):
        agg = r[:, t] + discount * agg
        out[:, t] = agg
    return out


def lambda_return(reward, value, length, discount, lambda_):
    """utility.py:97-110 (TD-lambda return; used as the advantage when gae_lambda is set, algorithm.py:336-338)."""
    mask = _mask(length, reward.shape[1], reward.dtype)
    seq = mask * reward + discount * value * (1 - lambda_)
    disc = mask * discount * lambda_
    out = torch.zeros_like(seq)
    agg = torch.zeros_like(seq[:, 0])
    for t in range(seq.shape[1] - 1, -1, -1):
        agg = seq[:, t] + disc[:, t] * agg
        out[:, t] = agg
    return out


def diag_normal_kl(mean0, logstd0, mean1, logstd1):
    """utility.py:127-132."""
    l0, l1 = 2 * logstd0, 2 * logstd1
    return 0.5 * (torch.exp(l0 - l1).sum(-1) + ((mean1 - mean0) ** 2 / torch.exp(l1)).sum(-1) + l1.sum(-1) - l0.sum(-1)
                  - mean0.shape[-1])


def diag_normal_logpdf(mean, logstd, loc):
    """utility.py:135-139, as written there: the constant is -0.5 (log 2 pi + logstd), not -0.5 log 2 pi - logstd, so
    this is the log density plus 0.5 sum(logstd).  Kept: the importance ratio of the reference's PPO is built on it."""
    return (-0.5 * (math.log(2 * math.pi) + logstd) - 0.5 * ((loc - mean) / torch.exp(logstd)) ** 2).sum(-1)


def diag_normal_entropy(mean, logstd):
    """utility.py:142-145."""
    return (mean.shape[-1] * math.log(2 * math.pi * math.e) + (2 * logstd).sum(-1)) / 2


class ForwardGaussianPolicy(torch.nn.Module):
    """networks.py:69-112: separate ReLU MLPs for the policy mean (tanh output) and the value; the log standard
    deviation is a free parameter vector.  Initialisers: xavier-uniform weights / zero biases (tf.contrib.layers
    .fully_connected defaults), mean layer variance_scaling(factor=init_mean_factor) = truncated normal with std
    sqrt(1.3 factor / fan_in), logstd = init_logstd (scripts/utility.py:108-111)."""

    def __init__(self, obs_dim, action_dim, cfg):
        super().__init__()
        def mlp(sizes):
            layers, last = [], obs_dim
            for s in sizes:
                lin = torch.nn.Linear(last, s)
                torch.nn.init.xavier_uniform_(lin.weight)
                torch.nn.init.zeros_(lin.bias)
                layers += [lin, torch.nn.ReLU()]
                last = s
            return torch.nn.Sequential(*layers), last
        self.policy, last = mlp(cfg.policy_layers)
        self.mean = torch.nn.Linear(last, action_dim)
        std = math.sqrt(1.3 * cfg.init_mean_factor / last)
        torch.nn.init.trunc_normal_(self.mean.weight, std=std, a=-2 * std, b=2 * std)
        torch.nn.init.zeros_(self.mean.bias)
        self.logstd = torch.nn.Parameter(torch.full((action_dim,), float(cfg.init_logstd)))
        self.value, last = mlp(cfg.value_layers)
        self.value_out = torch.nn.Linear(last, 1)
        torch.nn.init.xavier_uniform_(self.value_out.weight)
        torch.nn.init.zeros_(self.value_out.bias)

    def policy_parameters(self):
        return list(self.policy.parameters()) + list(self.mean.parameters()) + [self.logstd]

    def value_parameters(self):
        return list(self.value.parameters()) + list(self.value_out.parameters())

    def forward(self, observ):
        mean = torch.tanh(self.mean(self.policy(observ)))
        logstd = self.logstd.expand_as(mean)
        value = self.value_out(self.value(observ))[..., 0]
        return mean, logstd, value


class RecurrentGaussianPolicy(ForwardGaussianPolicy):
    """networks.py:113-159: as ForwardGaussianPolicy, but the LAST policy layer is a GRU cell of 100 units
    (tf.contrib.rnn.GRUBlockCell(100), whatever size policy_layers[-1] names); the value network stays feed-forward.
    The cell is TensorFlow's: r, u = sigmoid(W_ru [x, h] + b_ru) (gate biases start at 1), c = tanh(W_c [x, r h] + b_c),
    h' = u h + (1 - u) c -- the reset gate acts BEFORE the candidate's matrix product, unlike torch.nn.GRU.
    The state starts at zero with every episode (tf.nn.dynamic_rnn over whole episodes, algorithm.py:513-531)."""
    state_size = 100

    def __init__(self, obs_dim, action_dim, cfg):
        import copy
        inner = copy.copy(cfg)
        inner.policy_layers = tuple(cfg.policy_layers[:-1])
        super().__init__(obs_dim, action_dim, inner)
        feat = inner.policy_layers[-1] if inner.policy_layers else obs_dim
        H = self.state_size
        self.gates = torch.nn.Linear(feat + H, 2 * H)
        self.candidate = torch.nn.Linear(feat + H, H)
        for lin, bias in ((self.gates, 1.0), (self.candidate, 0.0)):
            torch.nn.init.xavier_uniform_(lin.weight)
            torch.nn.init.constant_(lin.bias, bias)
        self.mean = torch.nn.Linear(H, action_dim)
        std = math.sqrt(1.3 * cfg.init_mean_factor / H)
        torch.nn.init.trunc_normal_(self.mean.weight, std=std, a=-2 * std, b=2 * std)
        torch.nn.init.zeros_(self.mean.bias)

    def policy_parameters(self):
        return super().policy_parameters() + list(self.gates.parameters()) + list(self.candidate.parameters())

    def cell(self, x, h):
        r, u = torch.sigmoid(self.gates(torch.cat([x, h], -1))).chunk(2, -1)
        c = torch.tanh(self.candidate(torch.cat([x, r * h], -1)))
        return u * h + (1.0 - u) * c

    def step(self, observ, state):
        """one control step of a batch: observ [N, O], state [N, 100] -> (mean, logstd, value), new state"""
        state = self.cell(self.policy(observ), state)
        mean = torch.tanh(self.mean(state))
        return (mean, self.logstd.expand_as(mean), self.value_out(self.value(observ))[..., 0]), state

    def forward(self, observ):
        """whole episodes: observ [R, T, O], zero initial state"""
        x = self.policy(observ)
        h = torch.zeros(observ.shape[0], self.state_size, device=observ.device, dtype=observ.dtype)
        out = []
        for t in range(observ.shape[1]):
            h = self.cell(x[:, t], h)
            out.append(h)
        mean = torch.tanh(self.mean(torch.stack(out, 1)))
        return mean, self.logstd.expand_as(mean), self.value_out(self.value(observ))[..., 0]


class PPOAgent:
    """algorithm.py:26-531 for `num_agents` parallel environments whose tensors live on `device`."""

    def __init__(self, num_agents, obs_dim, action_dim, cfg=None, device="cuda", seed=0, sync_gradients=False):
        self.cfg = cfg = cfg or PPOConfig()
        self.n, self.device = num_agents, torch.device(device)
        gen = torch.Generator(device="cpu"); gen.manual_seed(seed)
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(seed)
            if cfg.network not in ("forward", "recurrent"):
                raise ValueError("PPOConfig.network must be 'forward' or 'recurrent'")
            self.net = (RecurrentGaussianPolicy if cfg.network == "recurrent" else ForwardGaussianPolicy)(obs_dim, action_dim, cfg).to(self.device)
        self.state = torch.zeros((num_agents, RecurrentGaussianPolicy.state_size), device=self.device) if cfg.network == "recurrent" else None
        self._gen = torch.Generator(device=self.device); self._gen.manual_seed(seed + 1)
        self.observ_filter = StreamingNormalize((obs_dim,), center=True, scale=True, clip=5, device=self.device)   # :48-52
        self.reward_filter = StreamingNormalize((), center=False, scale=True, clip=10, device=self.device)         # :53-57
        T = cfg.max_length
        def buffers(rows):
            return [torch.zeros((rows, T, obs_dim), device=self.device), torch.zeros((rows, T, action_dim), device=self.device),
                    torch.zeros((rows, T, action_dim), device=self.device), torch.zeros((rows, T, action_dim), device=self.device),
                    torch.zeros((rows, T), device=self.device)]
        self.episodes, self.episode_length = buffers(num_agents), torch.zeros(num_agents, dtype=torch.long, device=self.device)
        self.memory, self.memory_length = buffers(cfg.update_every), torch.zeros(cfg.update_every, dtype=torch.long, device=self.device)
        self.memory_index = 0
        self.penalty = float(cfg.kl_init_penalty)
        self.policy_opt = torch.optim.Adam(self.net.policy_parameters(), lr=cfg.policy_lr, eps=1e-8)
        self.value_opt = torch.optim.Adam(self.net.value_parameters(), lr=cfg.value_lr, eps=1e-8)
        self.sync_gradients = sync_gradients
        self._full_episodes_unpolled, self._warned_unpolled = 0, False
        self.last = None
        self.updates = 0
        self.log = []

    # ---- acting (algorithm.py:100-136) ----
    def begin_episode(self, indices):
        self.episode_length[indices] = 0
        if self.state is not None:
            self.state[indices] = 0.0

    @torch.no_grad()
    def perform(self, observ, training=True):
        if self.state is not None:
            (mean, logstd, _), self.state = self.net.step(self.observ_filter.transform(observ), self.state)
        else:
            mean, logstd, _ = self.net(self.observ_filter.transform(observ))
        if training:
            action = mean + torch.exp(logstd) * torch.randn(mean.shape, device=self.device, generator=self._gen)
        else:
            action = mean
        self.last = (action, mean, logstd)
        return action

    @torch.no_grad()
    def experience(self, observ, action, reward):
        """algorithm.py:138-180: update the filters, append the transition to every agent's running episode."""
        self.observ_filter.update(observ)
        self.reward_filter.update(reward)
        _, mean, logstd = self.last
        rows = torch.arange(self.n, device=self.device)
        t = self.episode_length.clamp(max=self.cfg.max_length - 1)
        for buf, val in zip(self.episodes, (observ, action, mean, logstd, reward)):
            buf[rows, t] = val
        self.episode_length += 1

    def end_episode(self, indices):
        """algorithm.py:182-212: finished episodes go to the memory while it has room; train when it is full."""
        space = self.cfg.update_every - self.memory_index
        use = indices[:space]
        k = use.numel()
        if k:
            dst = torch.arange(self.memory_index, self.memory_index + k, device=self.device)
            for mem, epi in zip(self.memory, self.episodes):
                mem[dst] = epi[use]
            self.memory_length[dst] = self.episode_length[use].clamp(max=self.cfg.max_length)
            self.memory_index += k
        if self.memory_index >= self.cfg.update_every:
            if not self._distributed():
                return self._training()
            # multi-rank learner: the update is collective and runs from train_if_all_full(), which the caller's loop has to
            # poll (train() does); a custom loop that never does would silently stop learning
            self._full_episodes_unpolled += int(indices.numel())
            if self._full_episodes_unpolled > 4 * self.cfg.update_every and not self._warned_unpolled:
                import warnings
                warnings.warn("PPOAgent(sync_gradients=True): the memory is full but train_if_all_full() has not been called; "
                              "poll it at the same loop positions on every rank (agents.ppo.train does)")
                self._warned_unpolled = True
        return None

    def _distributed(self):
        return self.sync_gradients and torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1

    def train_if_all_full(self):
        """Multi-rank learner: episodes end at different steps on different ranks, and an update is a sequence of
        blocking gradient all-reduces, so the decision to update is itself collective -- every rank calls this at the
        same loop positions (train() does, every `sync_poll` steps); the update runs, on every rank at once, as soon as
        every rank's memory is full (a full memory ignores further episodes, as in the reference, algorithm.py:196-199)."""
        if not self._distributed():
            return None
        full = torch.tensor([1.0 if self.memory_index >= self.cfg.update_every else 0.0], device=self.device)
        torch.distributed.all_reduce(full, op=torch.distributed.ReduceOp.MIN)
        self._full_episodes_unpolled = 0
        if float(full) < 1.0:
            return None
        self.observ_filter.merge_ranks()      # the ranks' filters have seen different envs: one filter from here on
        self.reward_filter.merge_ranks()
        return self._training()

    # ---- learning (algorithm.py:214-470) ----
    def _training(self):
        cfg = self.cfg
        observ, action, old_mean, old_logstd, reward = [m.clone() for m in self.memory]
        length = self.memory_length.clone()
        observ = self.observ_filter.transform(observ)
        reward = self.reward_filter.transform(reward)
        stats = {}
        stats.update(self._update_policy(observ, action, old_mean, old_logstd, reward, length))
        stats.update(self._update_value(observ, reward, length))
        stats.update(self._adjust_penalty(observ, old_mean, old_logstd, length))
        self.memory_index = 0
        self.memory_length.zero_()
        self.updates += 1
        self.log.append(stats)
        return stats

    def _sync(self, params):
        if self.sync_gradients and torch.distributed.is_available() and torch.distributed.is_initialized():
            world = torch.distributed.get_world_size()
            for p in params:
                if p.grad is not None:
                    torch.distributed.all_reduce(p.grad)
                    p.grad /= world

    def _update_value(self, observ, reward, length):
        mask = _mask(length, reward.shape[1])
        return_ = discounted_return(reward, length, self.cfg.discount)
        losses = []
        for _ in range(self.cfg.update_epochs_value):
            value = self.net(observ)[2] * mask                                      # dynamic_rnn: zero beyond the length
            loss = (0.5 * mask * (return_ - value) ** 2).mean()                     # :289-301
            self.value_opt.zero_grad(set_to_none=True)
            loss.backward()
            self._sync(self.net.value_parameters())
            self.value_opt.step()
            losses.append(loss.detach())
        return {"value_loss": float(torch.stack(losses).mean())}

    def _update_policy(self, observ, action, old_mean, old_logstd, reward, length):
        cfg = self.cfg
        with torch.no_grad():
            return_ = discounted_return(reward, length, cfg.discount)
            # the reference runs the network through tf.nn.dynamic_rnn(cell, observ, length) (algorithm.py:521), which
            # returns ZERO outputs beyond an episode's length: the value (and with it the advantage) of a padded step is
            # 0 there, so padding never enters advantage.mean() / var() nor the TD-lambda bootstrap of the last valid step
            value = self.net(observ)[2] * _mask(length, reward.shape[1])
            if cfg.gae_lambda:
                advantage = lambda_return(reward, value, length, cfg.discount, cfg.gae_lambda)   # :336-338
            else:
                advantage = return_ - value
            advantage = (advantage - advantage.mean()) / (advantage.var(unbiased=False).sqrt() + 1e-8)   # :341-342
        mask = _mask(length, reward.shape[1])
        losses = []
        for _ in range(cfg.update_epochs_policy):
            mean, logstd, _ = self.net(observ)
            kl = (mask * diag_normal_kl(old_mean, old_logstd, mean, logstd)).mean(1)                      # :399-401
            ratio = torch.exp(diag_normal_logpdf(mean, logstd, action) - diag_normal_logpdf(old_mean, old_logstd, action))
            surrogate = -(mask * ratio * advantage).mean(1)                                               # :402-406
            cutoff = cfg.kl_target * cfg.kl_cutoff_factor
            kl_cutoff = cfg.kl_cutoff_coef * (kl > cutoff).float() * (kl - cutoff) ** 2                   # :407-414
            loss = (surrogate + self.penalty * kl + kl_cutoff).mean()
            self.policy_opt.zero_grad(set_to_none=True)
            loss.backward()
            self._sync(self.net.policy_parameters())
            self.policy_opt.step()
            losses.append(loss.detach())
        return {"policy_loss": float(torch.stack(losses).mean()), "return": float(return_[:, 0].mean())}

    @torch.no_grad()
    def _adjust_penalty(self, observ, old_mean, old_logstd, length):
        """algorithm.py:436-474: x1.5 when the policy moved more than 1.3 kl_target, /1.5 below 0.7 kl_target."""
        mean, logstd, _ = self.net(observ)
        kl = (_mask(length, observ.shape[1]) * diag_normal_kl(old_mean, old_logstd, mean, logstd)).mean()
        if self._distributed():               # one penalty for all ranks: the mean KL over every rank's memory
            torch.distributed.all_reduce(kl)
            kl = kl / torch.distributed.get_world_size()
        kl_change = float(kl)
        if kl_change > 1.3 * self.cfg.kl_target:
            self.penalty *= 1.5
        if kl_change < 0.7 * self.cfg.kl_target:
            self.penalty /= 1.5
        return {"kl_change": kl_change, "penalty": self.penalty}


def train(env, agent, steps, training=True, sync_poll=8):
    """agents/tools/simulate.py:15-131 for an env with the RexBatchEnv surface (reset(indices), step(actions) on device
    tensors; auto_reset must be off: finished envs are reset here, by index, as simulate() does).
    Returns the mean score (undiscounted episode reward) and length of the episodes that ended."""
    dev = agent.device
    if training and agent._distributed():
        # the update decision is collective (train_if_all_full): every rank must poll at the same loop positions, i.e. run
        # the same number of steps with the same poll period -- checked once here, where a mismatch would otherwise block
        # a rank in an all-reduce for ever
        want = torch.tensor([float(steps), -float(steps), float(sync_poll), -float(sync_poll)], device=dev)
        torch.distributed.all_reduce(want, op=torch.distributed.ReduceOp.MAX)
        if want[0] != -want[1] or want[2] != -want[3]:
            raise ValueError("train(): every rank of a sync_gradients learner must pass the same `steps` and `sync_poll`")
    observ = env.reset().to(dev).clone()
    agent.begin_episode(torch.arange(agent.n, device=dev))
    score = torch.zeros(agent.n, device=dev)
    length = torch.zeros(agent.n, dtype=torch.long, device=dev)
    done = torch.zeros(agent.n, dtype=torch.bool, device=dev)
    scores, lengths = [], []
    for it in range(int(steps)):   # every rank of a multi-rank job runs the same number of iterations
        idx = done.nonzero()[:, 0]
        if idx.numel():
            observ[idx] = env.reset(idx.to(torch.int32)).to(dev)
            score[idx] = 0
            length[idx] = 0
            agent.begin_episode(idx)
        action = agent.perform(observ, training)
        nobs, reward, done, _ = env.step(action)
        nobs, reward, done = nobs.to(dev), reward.to(dev), done.to(dev).bool()
        score += reward
        length += 1
        if training:
            agent.experience(observ, action, reward)
        timeout = agent.episode_length >= agent.cfg.max_length            # LimitDuration (wrappers.py:268-291)
        done = done | timeout
        idx = done.nonzero()[:, 0]
        if idx.numel():
            scores.append(score[idx].clone()); lengths.append(length[idx].clone())
            if training:
                agent.end_episode(idx)
        if training and (it + 1) % sync_poll == 0:
            agent.train_if_all_full()         # no-op for a single-rank learner
        observ = nobs.clone()
    if not scores:
        return float("nan"), float("nan")
    return float(torch.cat(scores).mean()), float(torch.cat(lengths).float().mean())


def train_segments(env, agent, steps, segment=25, training=True, actor=None, seed=0):
    """The same loop with the ACTOR INSIDE THE LAUNCH and no per-step host synchronisation: what simulate() does step by step
    (agents/tools/simulate.py:57-131: reset finished envs, perform, simulate, experience, end_episode) done a rollout SEGMENT at a
    time.  `env.step_segment_policy` (rex_step_segment_policy) runs `segment` closed-loop steps in one launch -- the policy of
    `agent.net` behind `agent.observ_filter`, sampled in the kernel; finished envs are reset inside the launch (the env must have
    been created with auto_reset=True, range_normalize=True and max_episode_steps=agent.cfg.max_length: LimitDuration folded) -- and
    hands back the segment's blocks obs [T + 1, N, O], action / mean [T, N, A], reward, done [T, N].  The episode bookkeeping then
    runs on those blocks on the device: the filters are updated with the segment's observations and rewards (one batched Welford
    step instead of T), every transition is scattered to its place in its env's running episode, finished episodes go to the
    agent's memory (end_episode -> the update when it is full), in the order (episode ordinal within the segment, env index).
    Host synchronisations: one per segment (how many episodes ended) plus one per ordinal with finished episodes -- not three per
    step.  The actor's weights and filter statistics are refreshed (FusedActor.sync) before every segment, so a segment is rolled
    out with the policy and the filter as they stood at its start (the reference's perform() sees the filter of the step before).

    steps: control steps of every env (rounded up to whole segments).  Returns the mean score and length of the episodes that ended.
    `actor`: a FusedActor to reuse between calls (else one is made and left installed in the env)."""
    from .fused_actor import FusedActor
    dev, cfg, n = agent.device, agent.cfg, agent.n
    T = int(segment)
    if not env.config.auto_reset or not env.config.range_normalize or env.config.max_episode_steps != cfg.max_length:
        raise ValueError("train_segments: create the env with auto_reset=True, range_normalize=True, max_episode_steps=agent.cfg.max_length")
    if agent.state is not None:
        raise NotImplementedError("the recurrent policy runs through train(): its GRU state is not carried by the fused actor")
    if actor is None:
        actor = FusedActor(env, agent.net, agent.observ_filter, sample=training, seed=seed)
    O, A, Tmax = env.obs_dim, env.action_dim, cfg.max_length
    obs = torch.zeros((T + 1, n, O), device=dev)
    action, mean = torch.zeros((T, n, A), device=dev), torch.zeros((T, n, A), device=dev)
    reward, done = torch.zeros((T, n), device=dev), torch.zeros((T, n), dtype=torch.uint8, device=dev)
    obs[0].copy_(env.reset())
    agent.begin_episode(torch.arange(n, device=dev))
    score0 = torch.zeros(n, device=dev)
    tt = torch.arange(T, device=dev)[:, None]
    rows = torch.arange(n, device=dev)
    scores, lengths = [], []
    for _ in range((int(steps) + T - 1) // T):
        actor.sync()
        logstd = agent.net.logstd.detach().clone()
        env.step_segment_policy(T, obs[0], out=(obs[1:], reward, done), action=action, mean=mean)
        d = done.bool()
        if training:
            agent.observ_filter.update(obs[:T].reshape(T * n, O))
            agent.reward_filter.update(reward.reshape(T * n))
        # every transition's episode ordinal within the segment and its position inside its episode
        cnt = d.long().cumsum(0)
        ordinal = cnt - d.long()
        last_done = torch.where(d, tt, torch.full_like(cnt, -1)).cummax(0).values          # latest done step <= t
        prev_done = torch.cat([torch.full((1, n), -1, device=dev, dtype=torch.long), last_done[:-1]], 0)   # latest done step < t
        len0 = agent.episode_length.clone()
        pos = torch.where(ordinal == 0, len0[None, :] + tt, tt - prev_done - 1)
        csum = reward.cumsum(0)
        before = torch.where(prev_done >= 0, csum.gather(0, prev_done.clamp(min=0)), -score0[None, :].expand(T, n))
        ep_score = csum - before                                                             # of the episode (t, n) belongs to, up to t
        total = cnt[-1]
        rounds = int(total.max())                                                            # (the segment's one unconditional sync)
        for r in range(rounds + 1):
            m = ordinal == r
            if training:
                t_idx, n_idx = m.nonzero(as_tuple=True)
                p_idx = pos[t_idx, n_idx].clamp(max=Tmax - 1)
                for buf, val in zip(agent.episodes, (obs[:T], action, mean, None, reward)):
                    buf[n_idx, p_idx] = val[t_idx, n_idx] if val is not None else logstd.expand(t_idx.numel(), A)
            if r < rounds:
                ends = m & d                                                                 # at most one per env
                fin = (total > r).nonzero()[:, 0]
                ep_len = torch.where(ends, pos + 1, torch.zeros_like(pos)).sum(0)
                scores.append(torch.where(ends, ep_score, torch.zeros_like(ep_score)).sum(0)[fin])
                lengths.append(ep_len[fin])
                if training:
                    agent.episode_length[fin] = ep_len[fin]
                    agent.end_episode(fin)
                agent.begin_episode(fin)
        if training:
            agent.train_if_all_full()               # no-op for a single-rank learner
        # what the running episodes hold after the segment
        agent.episode_length.copy_(torch.where(total == 0, len0 + T, T - 1 - last_done[-1]))
        score0 = torch.where(total == 0, score0 + csum[-1], csum[-1] - csum.gather(0, last_done[-1:].clamp(min=0))[0])
        obs[0].copy_(obs[T])
    if not scores:
        return float("nan"), float("nan")
    return float(torch.cat(scores).mean()), float(torch.cat(lengths).float().mean())


if __name__ == "__main__":   # python -m rex_gym_amd.agents.ppo --task walk --envs 1024 --iterations 20
    import argparse
    from ..envs.batch_env import RexBatchEnv
    ap = argparse.ArgumentParser(description="train the reference's PPO on the batched HIP simulator")
    ap.add_argument("--task", default="walk"); ap.add_argument("--signal", default="ik")
    ap.add_argument("--envs", type=int, default=1024); ap.add_argument("--iterations", type=int, default=20)
    ap.add_argument("--max-length", type=int, default=500); ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gait-clock-scale", type=float, default=1.0,
                    help="wall-clock seconds per simulated second seen by GaitPlanner.loop (gait_planner.py:108-110)")
    ap.add_argument("--loop", default="segments", choices=["segments", "steps"],
                    help="segments: the actor inside the launch, one launch per --segment steps (train_segments); steps: perform() in PyTorch, "
                         "one launch and three host synchronisations per step (train, the reference's loop shape)")
    ap.add_argument("--segment", type=int, default=25)
    ap.add_argument("--toe-friction", type=float, default=None, help="pin the toe friction (RexBatchEnv(friction_range=(f, f))); the standup task matches its "
                                                                     "PyBullet record at 0.25 (DESIGN.md section 2)")
    ap.add_argument("--logdir", default=None, help="write the trained policy there as a TensorFlow-1 checkpoint the reference's policy player "
                                                   "restores (model.ckpt-<env steps>.index / .data-00000-of-00001 + checkpoint: policy_player.save_policy)")
    a = ap.parse_args()
    import time
    # the reference trains through RangeNormalize + ClipAction (playground/trainer.py:48-52): actions in [-1, 1]
    env = RexBatchEnv(a.envs, task=a.task, signal_type=a.signal, seed=a.seed, max_episode_steps=a.max_length, range_normalize=True,
                      gait_clock_scale=a.gait_clock_scale, auto_reset=a.loop == "segments", check_actions=False,
                      **({"friction_range": (a.toe_friction, a.toe_friction)} if a.toe_friction is not None else {}))
    agent = PPOAgent(a.envs, env.obs_dim, env.action_dim, PPOConfig(update_every=a.envs, max_length=a.max_length), seed=a.seed)
    actor = None
    if a.loop == "segments":
        from .fused_actor import FusedActor
        actor = FusedActor(env, agent.net, agent.observ_filter, sample=True, seed=a.seed)
    t0 = time.perf_counter()
    for it in range(a.iterations):
        if a.loop == "segments":
            score, length = train_segments(env, agent, a.max_length, segment=a.segment, actor=actor)
        else:
            score, length = train(env, agent, a.max_length)
        print(f"iteration {it}: mean score {score:.3f}, mean length {length:.1f}, updates {agent.updates}, "
              f"penalty {agent.penalty:.3g}, {time.perf_counter() - t0:.1f} s", flush=True)
    if a.logdir:
        import os
        from .policy_player import save_policy
        os.makedirs(a.logdir, exist_ok=True)
        steps = a.iterations * a.max_length * a.envs
        print("saved", save_policy(os.path.join(a.logdir, f"model.ckpt-{steps}"), agent.net, agent.observ_filter, global_step=steps))

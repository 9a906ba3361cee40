"""The PyTorch PPO learner (rex_gym_amd/agents/ppo.py) against direct restatements of the reference's formulas
(agents/ppo/normalize.py, utility.py, algorithm.py) and on a toy batch env.  CPU only."""
import math

import numpy as np
import pytest
import torch

from rex_gym_amd.agents import PPOAgent, PPOConfig, StreamingNormalize, train
from rex_gym_amd.agents import ppo


def test_streaming_normalize_matches_welford_and_the_reference_constants():
    rng = np.random.RandomState(0)
    data = rng.randn(7, 5, 3) * np.array([1.0, 10.0, 0.1]) + np.array([0.0, -3.0, 7.0])
    f = StreamingNormalize((3,), center=True, scale=True, clip=5)
    x0 = torch.tensor(data[0, :1], dtype=torch.float32)
    assert torch.equal(f.transform(x0), x0.clamp(-5, 5))           # count 0: mean 0, scale 1 (normalize.py:58-60)
    for b in data:
        f.update(torch.tensor(b, dtype=torch.float32))
    flat = data.reshape(-1, 3)
    np.testing.assert_allclose(f.mean.numpy(), flat.mean(0), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(f.std().numpy(), np.sqrt(flat.var(0, ddof=1) + 1e-4), rtol=1e-4)   # normalize.py:131-141
    v = torch.tensor(flat[:4], dtype=torch.float32)
    want = np.clip((flat[:4] - flat.mean(0)) / (np.sqrt(flat.var(0, ddof=1) + 1e-4) + 1e-8), -5, 5)
    np.testing.assert_allclose(f.transform(v).numpy(), want, rtol=1e-4, atol=1e-5)
    g = StreamingNormalize((), center=False, scale=True, clip=10)  # the reward filter: scale only
    g.update(torch.tensor([2.0])); assert float(g.transform(torch.tensor(3.0))) == 3.0   # one sample: scale 1
    g.update(torch.tensor([4.0, 6.0]))
    np.testing.assert_allclose(float(g.transform(torch.tensor(3.0))), 3.0 / (math.sqrt(4.0 + 1e-4) + 1e-8), rtol=1e-6)


def test_returns_match_brute_force():
    rng = np.random.RandomState(1)
    B, T, disc, lam = 5, 12, 0.9, 0.7
    r, v = rng.randn(B, T), rng.randn(B, T)
    length = np.array([12, 1, 7, 3, 10])
    ret = ppo.discounted_return(torch.tensor(r), torch.tensor(length), disc).numpy()
    lr = ppo.lambda_return(torch.tensor(r), torch.tensor(v), torch.tensor(length), disc, lam).numpy()
    for b in range(B):
        for t in range(T):
            want = sum(disc ** (k - t) * r[b, k] for k in range(t, length[b])) if t < length[b] else 0.0
            assert abs(ret[b, t] - want) < 1e-10
        agg = 0.0                                                       # utility.py:97-110, written as its recurrence
        for t in range(T - 1, -1, -1):
            m = 1.0 if t < length[b] else 0.0
            agg = m * r[b, t] + disc * v[b, t] * (1 - lam) + m * disc * lam * agg
            assert abs(lr[b, t] - agg) < 1e-10


def test_diagonal_normal_formulas():
    from scipy import stats
    rng = np.random.RandomState(2)
    m0, m1, s0, s1, x = rng.randn(4, 3), rng.randn(4, 3), rng.randn(4, 3) * 0.3, rng.randn(4, 3) * 0.3, rng.randn(4, 3)
    t = lambda a: torch.tensor(a)
    # utility.py:135-139 as written: the normalising term is -0.5 (log 2 pi + logstd), i.e. the true log density
    # plus 0.5 sum(logstd) -- a quirk of the reference that the importance ratio of its PPO inherits, kept as is
    np.testing.assert_allclose(ppo.diag_normal_logpdf(t(m0), t(s0), t(x)).numpy(),
                               stats.norm.logpdf(x, m0, np.exp(s0)).sum(-1) + 0.5 * s0.sum(-1), rtol=1e-10)
    np.testing.assert_allclose(ppo.diag_normal_entropy(t(m0), t(s0)).numpy(), stats.norm.entropy(m0, np.exp(s0)).sum(-1), rtol=1e-10)
    kl = (s1 - s0 + (np.exp(2 * s0) + (m0 - m1) ** 2) / (2 * np.exp(2 * s1)) - 0.5).sum(-1)
    np.testing.assert_allclose(ppo.diag_normal_kl(t(m0), t(s0), t(m1), t(s1)).numpy(), kl, rtol=1e-10)
    assert float(ppo.diag_normal_kl(t(m0), t(s0), t(m0), t(s0)).abs().max()) < 1e-12


def test_network_shapes_and_initialisers():
    cfg = PPOConfig()
    net = ppo.ForwardGaussianPolicy(4, 2, cfg)
    assert [m.out_features for m in net.policy if isinstance(m, torch.nn.Linear)] == [200, 100]       # configs.py:31-32
    assert [m.out_features for m in net.value if isinstance(m, torch.nn.Linear)] == [200, 100]
    assert torch.all(net.logstd == -1.0)                                                              # init_logstd
    std = math.sqrt(1.3 * 0.05 / 100)
    w = net.mean.weight.detach()
    assert float(w.abs().max()) <= 2 * std + 1e-7 and 0.5 * std < float(w.std()) < 1.2 * std
    mean, logstd, value = net(torch.zeros(7, 4))
    assert mean.shape == (7, 2) and logstd.shape == (7, 2) and value.shape == (7,) and float(mean.detach().abs().max()) < 1.0


class _PointEnv:
    """N points on a line; action moves the point, reward = -|x|; episodes of fixed length -- or, with `fall` > 0, ending
    at random (each step with that probability), like robots that fall at different times."""
    def __init__(self, n, seed=0, fall=0.0):
        self.n, self.g = n, torch.Generator().manual_seed(seed)
        self.x = torch.zeros(n, 1)
        self.fall = fall
    def reset(self, indices=None):
        if indices is None:
            self.x = torch.rand((self.n, 1), generator=self.g) * 4 - 2
            return self.x.clone()
        idx = indices.long()
        self.x[idx] = torch.rand((idx.numel(), 1), generator=self.g) * 4 - 2
        return self.x[idx].clone()
    def step(self, a):
        self.x = self.x + 0.5 * a.clamp(-1, 1)
        done = torch.rand(self.n, generator=self.g) < self.fall
        return self.x.clone(), -self.x[:, 0].abs(), done, {}


def test_ppo_learns_a_toy_task_and_adapts_its_penalty():
    torch.manual_seed(0)
    n = 64
    cfg = PPOConfig(policy_layers=(32,), value_layers=(32,), update_every=n, update_epochs_policy=15, update_epochs_value=15,
                    policy_lr=3e-3, value_lr=3e-3, max_length=12, discount=0.9, init_mean_factor=0.1)
    env, agent = _PointEnv(n), PPOAgent(n, 1, 1, cfg, device="cpu", seed=3)
    first, _ = train(env, agent, 12 * 3)
    for _ in range(12):
        last, length = train(env, agent, 12 * 3)
    assert agent.updates >= 30 and length == 12
    assert last > first + 1.0, (first, last)                     # mean episode score rises (less distance from 0)
    kls = [s["kl_change"] for s in agent.log]
    assert all(np.isfinite(k) for k in kls) and max(kls) < 1.0
    pens = [s["penalty"] for s in agent.log]
    assert len(set(pens)) > 1                                     # x1.5 / /1.5 steps happened (algorithm.py:458-469)
    assert all(abs(math.log(pens[i + 1] / pens[i]) / math.log(1.5)) in (0.0, 1.0) or abs(abs(math.log(pens[i + 1] / pens[i]) / math.log(1.5)) - 1) < 1e-6
               for i in range(len(pens) - 1))


def test_memory_takes_only_the_episodes_it_has_room_for():
    cfg = PPOConfig(policy_layers=(8,), value_layers=(8,), update_every=4, update_epochs_policy=1, update_epochs_value=1, max_length=5)
    agent = PPOAgent(6, 2, 1, cfg, device="cpu")
    obs = torch.zeros(6, 2)
    for _ in range(3):
        a = agent.perform(obs)
        agent.experience(obs, a, torch.ones(6))
    assert agent.end_episode(torch.tensor([0, 2])) is None and agent.memory_index == 2
    assert list(agent.memory_length[:2]) == [3, 3]
    stats = agent.end_episode(torch.tensor([1, 3, 4, 5]))         # room for two more: trains, then the memory is empty
    assert stats is not None and agent.memory_index == 0 and agent.updates == 1


def test_padding_beyond_an_episode_never_enters_the_advantage():
    """The reference evaluates the network with tf.nn.dynamic_rnn(cell, observ, length) (algorithm.py:521): outputs
    beyond an episode's length are zero.  With short episodes and stale observations in the padding, the advantage
    statistics and the TD-lambda bootstrap must be those of the valid steps alone."""
    torch.manual_seed(1)
    for lam in (None, 0.8):
        cfg = PPOConfig(policy_layers=(8,), value_layers=(8,), update_every=4, update_epochs_policy=1, update_epochs_value=1,
                        max_length=6, gae_lambda=lam, policy_lr=0.0, value_lr=0.0)
        results = []
        for pad in (0.0, 50.0):                                   # same episodes, different garbage in the padding
            agent = PPOAgent(4, 2, 1, cfg, device="cpu", seed=5)
            g = torch.Generator().manual_seed(9)
            length = torch.tensor([6, 2, 3, 1])
            obs = torch.randn((4, 6, 2), generator=g)
            act = torch.randn((4, 6, 1), generator=g)
            rew = torch.randn((4, 6), generator=g)
            m = ppo._mask(length, 6)
            obs = obs * m[..., None] + pad * (1 - m[..., None])
            rew = rew * m + pad * (1 - m)
            mean, logstd, _ = agent.net(obs)
            agent.memory = [obs, act, mean.detach(), logstd.detach(), rew]
            agent.memory_length = length
            captured = {}
            orig = ppo.diag_normal_logpdf
            stats = agent._update_policy(obs, act, mean.detach(), logstd.detach(), rew, length)
            results.append(stats["policy_loss"])
        assert abs(results[0] - results[1]) < 1e-6, (lam, results)


def test_recurrent_policy_is_the_reference_gru_and_learns():
    """RecurrentGaussianPolicy (networks.py:113-159): last policy layer = TensorFlow's GRU block cell of 100 units (the
    reset gate acts before the candidate's product, gate biases start at 1); acting step by step with a carried state
    equals evaluating the whole episode from a zero state; the state is cleared at episode begin; PPO learns with it."""
    torch.manual_seed(2)
    cfg = PPOConfig(policy_layers=(16, 100), value_layers=(8,), network="recurrent")
    net = ppo.RecurrentGaussianPolicy(3, 2, cfg).double()
    assert [m.out_features for m in net.policy if isinstance(m, torch.nn.Linear)] == [16] and net.mean.in_features == 100
    assert torch.all(net.gates.bias == 1.0) and torch.all(net.candidate.bias == 0.0)
    obs = torch.randn(4, 9, 3, dtype=torch.float64)
    mean, logstd, value = net(obs)
    assert mean.shape == (4, 9, 2) and value.shape == (4, 9)
    # numpy restatement of the cell, step by step
    W = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    def dense(x, name):
        return x @ W[name + ".weight"].T + W[name + ".bias"]
    h = np.zeros((4, 100))
    state = torch.zeros(4, 100, dtype=torch.float64)
    for t in range(9):
        x = np.maximum(dense(obs[:, t].numpy(), "policy.0"), 0.0)
        ru = 1.0 / (1.0 + np.exp(-dense(np.concatenate([x, h], 1), "gates")))
        r, u = ru[:, :100], ru[:, 100:]
        c = np.tanh(dense(np.concatenate([x, r * h], 1), "candidate"))
        h = u * h + (1.0 - u) * c
        np.testing.assert_allclose(mean[:, t].detach().numpy(), np.tanh(dense(h, "mean")), rtol=1e-10, atol=1e-12)
        (m1, _, v1), state = net.step(obs[:, t], state)
        np.testing.assert_allclose(m1.detach().numpy(), mean[:, t].detach().numpy(), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(v1.detach().numpy(), value[:, t].detach().numpy(), rtol=1e-10, atol=1e-12)
    # the agent: state carried between perform() calls, cleared by begin_episode(); learning on the toy task
    n = 32
    cfg = PPOConfig(policy_layers=(16, 100), value_layers=(32,), update_every=n, update_epochs_policy=10, update_epochs_value=10,
                    policy_lr=3e-3, value_lr=3e-3, max_length=8, discount=0.9, init_mean_factor=0.1, network="recurrent")
    env, agent = _PointEnv(n), PPOAgent(n, 1, 1, cfg, device="cpu", seed=3)
    agent.perform(torch.ones(n, 1))
    assert float(agent.state.abs().max()) > 0
    agent.begin_episode(torch.tensor([0, 5]))
    assert float(agent.state[[0, 5]].abs().max()) == 0 and float(agent.state[1].abs().max()) > 0
    first, _ = train(env, agent, 8 * 3)
    for _ in range(10):
        last, _ = train(env, agent, 8 * 3)
    assert agent.updates >= 20 and last > first + 0.5, (first, last)
    with pytest.raises(ValueError):
        PPOAgent(2, 1, 1, PPOConfig(network="lstm"), device="cpu")


# ---------------------------------------------------------------- train_segments: the actor inside the launch, bookkeeping per segment
class _SegmentPointEnv:
    """The closed-loop surface of RexBatchEnv (set_policy / step_segment_policy, in-launch auto-reset, folded episode cap) on the toy
    task, on the CPU: the actor is evaluated here from the arrays FusedActor hands over, exactly as csrc/rex_policy.h defines it.
    Every transition is logged per env so that a test can rebuild the episodes independently."""
    def __init__(self, n, seed=0, fall=0.0, max_episode_steps=12):
        from types import SimpleNamespace
        self.n, self.g = n, torch.Generator().manual_seed(seed)
        self.device, self.obs_dim, self.action_dim = torch.device("cpu"), 1, 1
        self.config = SimpleNamespace(auto_reset=1, range_normalize=1, max_episode_steps=max_episode_steps)
        self.fall, self.x, self.steps = fall, torch.zeros(n, 1), torch.zeros(n, dtype=torch.long)
        self.log = [[] for _ in range(n)]
        self.pol = None

    def reset(self):
        self.x = torch.rand((self.n, 1), generator=self.g) * 4 - 2
        self.steps.zero_()
        return self.x.clone()

    def set_policy(self, w1, b1, w2, b2, w3, b3, logstd, obs_mean=None, obs_scale=None, obs_clip=5.0, sample=True, seed=0):
        self.pol = [t.clone() if t is not None else None for t in (w1, b1, w2, b2, w3, b3, logstd, obs_mean, obs_scale)] + [obs_clip, sample]

    def step_segment_policy(self, T, obs_in, out, action, mean):
        w1, b1, w2, b2, w3, b3, logstd, om, osc, clip, sample = self.pol
        obs, reward, done = out
        prev = obs_in
        for t in range(T):
            x = ((prev - om) * osc).clamp(-clip, clip) if om is not None else prev
            m = torch.tanh(torch.relu(torch.relu(x @ w1 + b1) @ w2 + b2) @ w3 + b3)
            a = m + torch.exp(logstd) * torch.randn(m.shape, generator=self.g) if sample else m
            self.x = self.x + 0.5 * a.clamp(-1, 1)
            self.steps += 1
            r = -self.x[:, 0].abs()
            d = (torch.rand(self.n, generator=self.g) < self.fall) | (self.steps >= self.config.max_episode_steps)
            for i in range(self.n):
                self.log[i].append((prev[i].clone(), a[i].clone(), m[i].clone(), float(r[i]), bool(d[i])))
            idx = d.nonzero()[:, 0]
            if idx.numel():                                         # in-launch auto-reset: the returned observation opens the new episode
                self.x[idx] = torch.rand((idx.numel(), 1), generator=self.g) * 4 - 2
                self.steps[idx] = 0
            action[t], mean[t], obs[t], reward[t], done[t] = a, m, self.x, r, d.to(done.dtype)
            prev = self.x.clone()
        return obs, reward, done.bool(), {}


@pytest.mark.parametrize("fall,T", [(0.0, 5), (0.15, 7), (0.6, 25)])     # no falls; some; several episodes of an env inside one segment
def test_train_segments_rebuilds_every_episode_from_the_segment_blocks(fall, T):
    """The device-side bookkeeping of train_segments (episode ordinal and position of every transition of a [T, N] block, scatter into the
    running episodes, hand-over of the finished ones) against episodes rebuilt step by step from the env's own log: every episode
    handed to end_episode is, transition for transition, the next unfinished episode of its env; the running episodes and scores
    after the call are the tails."""
    n, Tmax = 9, 6
    cfg = PPOConfig(policy_layers=(8, 8), value_layers=(8,), update_every=100000, max_length=Tmax)
    env, agent = _SegmentPointEnv(n, seed=4, fall=fall, max_episode_steps=Tmax), PPOAgent(n, 1, 1, cfg, device="cpu", seed=2)
    handed = [[] for _ in range(n)]
    orig = agent.end_episode

    def spy(indices):
        for i in indices.tolist():
            L = int(agent.episode_length[i])
            handed[i].append([(agent.episodes[0][i, t].clone(), agent.episodes[1][i, t].clone(), agent.episodes[2][i, t].clone(),
                               float(agent.episodes[4][i, t])) for t in range(L)])
        return orig(indices)
    agent.end_episode = spy
    score, length = ppo.train_segments(env, agent, 4 * T + 3, segment=T)
    steps = ((4 * T + 3 + T - 1) // T) * T
    total_eps, total_len, total_score = 0, 0, 0.0
    for i in range(n):
        assert len(env.log[i]) == steps
        want, cur = [], []
        for tr in env.log[i]:
            cur.append(tr)
            if tr[4]:
                want.append(cur); cur = []
        assert len(handed[i]) == len(want), (i, len(handed[i]), len(want))
        for got, exp in zip(handed[i], want):
            assert len(got) == len(exp)
            for (go, ga, gm, gr), (eo, ea, em, er, _) in zip(got, exp):
                assert torch.equal(go, eo) and torch.equal(ga, ea) and torch.equal(gm, em) and abs(gr - er) < 1e-6
            total_eps += 1; total_len += len(exp); total_score += sum(e[3] for e in exp)
        assert int(agent.episode_length[i]) == len(cur)             # the running episode: the tail
        for t, (eo, ea, em, er, _) in enumerate(cur):
            assert torch.equal(agent.episodes[0][i, t], eo) and torch.equal(agent.episodes[1][i, t], ea) and abs(float(agent.episodes[4][i, t]) - er) < 1e-6
    assert total_eps > 0 and abs(length - total_len / total_eps) < 1e-4 and abs(score - total_score / total_eps) < 1e-3
    assert int(agent.observ_filter.count) == steps * n and int(agent.reward_filter.count) == steps * n
    if fall == 0.6:
        assert max(len(h) for h in handed) >= 2 * (steps // T)       # (the multi-episode-per-segment path ran)


def test_train_segments_learns_the_toy_task():
    torch.manual_seed(0)
    n = 64
    cfg = PPOConfig(policy_layers=(32, 16), value_layers=(32,), update_every=n, update_epochs_policy=15, update_epochs_value=15,
                    policy_lr=3e-3, value_lr=3e-3, max_length=12, discount=0.9, init_mean_factor=0.1)
    env, agent = _SegmentPointEnv(n, max_episode_steps=12), PPOAgent(n, 1, 1, cfg, device="cpu", seed=3)
    first, _ = ppo.train_segments(env, agent, 12 * 3, segment=6)
    for _ in range(12):
        last, length = ppo.train_segments(env, agent, 12 * 3, segment=6)
    assert agent.updates >= 30 and length == 12
    assert last > first + 1.0, (first, last)
    with pytest.raises(ValueError):                                 # the env must fold auto-reset, the wrappers and the episode cap
        ppo.train_segments(_SegmentPointEnv(n, max_episode_steps=7), agent, 12)

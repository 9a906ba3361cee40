"""GPU tests of the fused actor: rex_set_policy / rex_step_policy / rex_step_segment_policy (include/rexsim.h, csrc/rex_policy.h) --
the closed-loop rollout of the reference's agents (agents/tools/simulate.py:57-76: perform(prevob) -> simulate(action) every step)
inside the step launch.  Every call goes through the C ABI (RexBatchEnv -> ctypes -> librexsim_hip.so)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "gpu tests need an MI355X"
    return t


def _actor(torch, env, seed=5, layers=(200, 100), with_filter=True, sample=True, logstd=-1.0, big=False):
    """A ForwardGaussianPolicy with the reference's initialisers (big=True: mean-layer weights large enough for the tanh to bend) and a
    filter that has seen a few observation batches, packed for the kernel."""
    from rex_gym_amd.agents.fused_actor import FusedActor
    from rex_gym_amd.agents.ppo import ForwardGaussianPolicy, PPOConfig, StreamingNormalize
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        net = ForwardGaussianPolicy(env.obs_dim, env.action_dim, PPOConfig(policy_layers=tuple(layers), init_logstd=logstd)).to(env.device)
        if big:
            with torch.no_grad():
                net.mean.weight.mul_(12.0); net.mean.bias.uniform_(-0.3, 0.3)
                for m in net.policy:
                    if isinstance(m, torch.nn.Linear):
                        m.bias.uniform_(-0.1, 0.1)
    flt = None
    if with_filter:
        flt = StreamingNormalize((env.obs_dim,), center=True, scale=True, clip=5, device=env.device)
        g = torch.Generator(device=env.device); g.manual_seed(seed)
        flt.update(torch.randn((64, env.obs_dim), device=env.device, generator=g) * 0.02)
        flt.update(torch.randn((64, env.obs_dim), device=env.device, generator=g) * 0.05 + 0.01)
    return FusedActor(env, net, flt, sample=sample, seed=seed)


_CASES = [
    # id, envs per wave, envs, env keywords
    ("base-walk_ik", (4, 8, 16), 1003, dict(task="walk", signal_type="ik")),
    ("base-walk_ol", (4,), 1003, dict(task="walk", signal_type="ol")),                   # 8 action words
    ("base-gallop_ol", (4, 8, 16), 1003, dict(task="gallop", signal_type="ol")),         # 16 observation words, 4 action words
    ("base-standup", (8,), 515, dict(task="standup", signal_type="ol")),
    ("base-turn_ik_heightfield", (4,), 515, dict(task="turn", signal_type="ik", terrain_type="random")),
    ("arm-walk_ik", (4, 8, 16), 515, dict(task="walk", signal_type="ik", mark="arm")),
    ("arm-gallop_ol", (4, 16), 515, dict(task="gallop", signal_type="ol", mark="arm")),  # 22 observation words
    ("base-walk_ik_latency", (4,), 515, dict(task="walk", signal_type="ik", control_latency=0.02, pd_latency=0.003)),
]


@pytest.mark.parametrize("case,epw", [(c[0], e) for c in _CASES for e in c[1]])
def test_policy_segment_is_bit_identical_to_single_policy_steps_and_matches_torch(torch, case, epw, monkeypatch):
    """For every variant group x envs per wave: 3 segments of 23 closed-loop steps from reset through in-launch auto-resets (episode cap
    25), one launch per segment (rex_step_segment_policy), against the same steps one launch each (rex_step_policy, chained through
    their observations) BIT FOR BIT -- action, mean, observation, reward, done, motor command of every step, the state block after every
    segment.  Then the network itself: the mean of every step against a float64 evaluation of the same weights on the observation
    the step acted on (<= 1e-5), the sample (action - mean) / exp(logstd) against N(0, 1), and the env half: a twin env fed the
    recorded actions through plain step() returns the same bits."""
    from rex_gym_amd import RexBatchEnv
    _, _, n, kw = next(c for c in _CASES if c[0] == case)
    monkeypatch.setenv("REX_ENVS_PER_WAVE", str(epw))
    mk = lambda: RexBatchEnv(n, seed=17, auto_reset=True, max_episode_steps=25, check_actions=False, range_normalize=True, **kw)
    one, seg, twin = mk(), mk(), mk()
    assert seg._L.rex_envs_per_wave(seg._h) == epw
    a1, a2 = _actor(torch, one, big=True), _actor(torch, seg, big=True)
    o1, o2, o3 = one.reset(), seg.reset(), twin.reset()
    assert torch.equal(o1, o2) and torch.equal(o1, o3)
    T, nm, A, O = 23, one.num_motors, one.action_dim, one.obs_dim
    ended = 0
    zs = []
    for s in range(3):
        cmd = torch.zeros((T, n, nm), device="cuda")
        so, sr, sd, si = seg.step_segment_policy(T, o2, motor_cmd=cmd)
        assert so.shape == (T, n, O) and sr.shape == (T, n) and sd.dtype == torch.bool and si["policy_action"].shape == (T, n, A)
        prev = o1
        for t in range(T):
            c1 = torch.zeros((n, nm), device="cuda")
            oo, orw, od, oi = one.step_policy(prev, motor_cmd=c1)
            for name, x, y in (("policy_action", oi["policy_action"], si["policy_action"][t]), ("policy_mean", oi["policy_mean"], si["policy_mean"][t]),
                               ("obs", oo, so[t]), ("reward", orw, sr[t]), ("done", od, sd[t]), ("motor command", c1, cmd[t])):
                if not torch.equal(x, y):
                    bad = (x != y)
                    pytest.fail(f"{case} epw {epw}: segment {s} step {t}: {name} of the segment launch differs from the single policy steps' in "
                                f"{int(bad.sum())} words (first at {tuple(int(v) for v in torch.nonzero(bad)[0])})")
            # the env half: the recorded action through the plain step of a twin env
            to, tr, td, ti = twin.step(oi["policy_action"])
            assert torch.equal(to, oo) and torch.equal(tr, orw) and torch.equal(td, od) and torch.equal(ti["action"], c1), (case, epw, s, t, "twin")
            ended += int(od.sum())
            prev = oo.clone()
        assert torch.equal(one.state, seg.state) and torch.equal(one.state, twin.state), (case, epw, s)
        # network parity over the whole segment at once: inputs = [o2, so[0..T-2]]
        x = torch.cat([o2[None], so[:-1]], 0).double().cpu()
        w = {k: getattr(a2, k).double().cpu() for k in ("w1", "b1", "w2", "b2", "w3", "b3", "logstd", "obs_mean", "obs_scale")}
        xf = ((x - w["obs_mean"]) * w["obs_scale"]).clamp(-a2.obs_clip, a2.obs_clip)
        mean64 = torch.tanh(torch.relu(torch.relu(xf @ w["w1"] + w["b1"]) @ w["w2"] + w["b2"]) @ w["w3"] + w["b3"])
        err = (si["policy_mean"].double().cpu() - mean64).abs().max().item()
        assert err <= 1e-5, (case, epw, s, err)
        assert mean64.abs().max() > 0.2 and mean64.abs().max() <= 1.0         # (the tanh is exercised)
        zs.append(((si["policy_action"] - si["policy_mean"]).double().cpu() / torch.exp(w["logstd"])).reshape(-1))
        o1, o2 = prev, so[-1].clone()
    assert ended >= 2 * n                       # the comparison ran through in-launch resets in every env
    z = torch.cat(zs)
    assert abs(z.mean().item()) < 0.02 and abs(z.std().item() - 1.0) < 0.02 and abs((z ** 3).mean().item()) < 0.06 and abs((z ** 4).mean().item() - 3.0) < 0.15, \
        (z.mean().item(), z.std().item(), (z ** 4).mean().item())
    assert len(torch.unique(z)) > 0.98 * z.numel()       # every (env, episode, step, action word) draws its own number
    one.close(); seg.close(); twin.close()


def test_policy_evaluation_mode_other_layer_sizes_and_in_place_weight_updates(torch):
    """sample=False gives the mean; hidden layers other than 200 / 100 (odd sizes, more than 128 units in the second layer, no filter);
    weights rewritten in place between launches are what the next launch evaluates; rex_step_policy needs no re-set."""
    from rex_gym_amd import RexBatchEnv
    n = 300
    env = RexBatchEnv(n, task="walk", signal_type="ik", seed=3, auto_reset=True, max_episode_steps=40, check_actions=False, range_normalize=True)
    for layers, flt in (((37, 150), False), ((64, 64), True), ((1, 1), True), ((230, 130), False)):
        act = _actor(torch, env, layers=layers, with_filter=flt, sample=False, big=True)
        obs = env.reset()
        o, r, d, info = env.step_segment_policy(7, obs)
        assert torch.equal(info["policy_action"], info["policy_mean"])
        x = torch.cat([obs[None], o[:-1]], 0)
        w = {k: (getattr(act, k).double().cpu() if getattr(act, k) is not None else None) for k in ("w1", "b1", "w2", "b2", "w3", "b3", "obs_mean", "obs_scale")}
        xf = x.double().cpu()
        if flt:
            xf = ((xf - w["obs_mean"]) * w["obs_scale"]).clamp(-5, 5)
        mean64 = torch.tanh(torch.relu(torch.relu(xf @ w["w1"] + w["b1"]) @ w["w2"] + w["b2"]) @ w["w3"] + w["b3"])
        assert (info["policy_mean"].double().cpu() - mean64).abs().max().item() <= 1e-5, layers
        # in place: flip the sign of the mean layer -> the next launch's means flip (same observation)
        with torch.no_grad():
            act.net.mean.weight.neg_(); act.net.mean.bias.neg_()
        act.sync()
        env.reset()
        _, _, _, info2 = env.step_policy(obs)
        assert torch.allclose(info2["policy_mean"], -info["policy_mean"][0], atol=1e-6)
    # too large for the LDS scratch of the kernel variant
    from rex_gym_amd import _lib
    with pytest.raises(_lib.RexSimError, match="LDS"):
        _actor(torch, env, layers=(300, 200))
    env.close()


def test_policy_rollouts_do_not_depend_on_the_sharding_or_the_envs_per_wave(torch, monkeypatch):
    """The sample of env g at (episode, step) is keyed by its GLOBAL index: two shards of n / 2 envs (env_index_base) reproduce the rows of
    one batch of n bit for bit, and so does the same batch at another envs-per-wave -- observation, action, reward, done over a 30-step
    closed-loop segment through in-launch resets."""
    from rex_gym_amd import RexBatchEnv
    n, T = 2048, 30
    kw = dict(task="walk", signal_type="ik", seed=9, auto_reset=True, max_episode_steps=12, check_actions=False, range_normalize=True)
    def rollout(count, base, epw):
        monkeypatch.setenv("REX_ENVS_PER_WAVE", str(epw))
        env = RexBatchEnv(count, env_index_base=base, **kw)
        _actor(torch, env, seed=21, big=True)
        o, r, d, info = env.step_segment_policy(T, env.reset())
        out = (o.clone(), r.clone(), d.clone(), info["policy_action"].clone(), env.state.clone())
        env.close()
        return out
    whole = rollout(n, 0, 4)
    for epw in (8, 16):
        other = rollout(n, 0, epw)
        # (the physics of different envs-per-wave variants agrees to rounding, not bit for bit -- DESIGN.md section 2 --, so only the first
        #  step's action, a function of the reset observation alone, is compared across variants)
        assert torch.equal(whole[3][0], other[3][0])
    lo, hi = rollout(n // 2, 0, 4), rollout(n // 2, n // 2, 4)
    for k in range(4):
        assert torch.equal(whole[k][:, : n // 2], lo[k]) and torch.equal(whole[k][:, n // 2:], hi[k]), k
    assert torch.equal(whole[4][:, : n // 2], lo[4]) and torch.equal(whole[4][:, n // 2:], hi[4])


def test_streamed_and_lds_resident_weights_give_the_same_bits(torch, monkeypatch):
    """Where the packed actor fits next to four waves' rows it is loaded into LDS once per launch, else streamed from L2 every step
    (csrc/rex_policy.h); the arithmetic is the same MFMA chains either way.  REX_POLICY_LDS=0 (read at rex_set_policy) forces the
    streamed path: a 40-step closed-loop segment through in-launch resets returns the same bits as the default, at 4 and 8 envs per wave,
    walk-IK and gallop-OL (16 observation words)."""
    from rex_gym_amd import RexBatchEnv
    for task, signal, epw in (("walk", "ik", 4), ("walk", "ik", 8), ("gallop", "ol", 4)):
        outs = []
        for lds in ("1", "0"):
            monkeypatch.setenv("REX_ENVS_PER_WAVE", str(epw)); monkeypatch.setenv("REX_POLICY_LDS", lds)
            env = RexBatchEnv(700, task=task, signal_type=signal, seed=4, auto_reset=True, max_episode_steps=15, check_actions=False, range_normalize=True)
            _actor(torch, env, seed=8, big=True)
            o, r, d, info = env.step_segment_policy(40, env.reset())
            outs.append((o.clone(), r.clone(), d.clone(), info["policy_action"].clone(), info["policy_mean"].clone(), env.state.clone()))
            env.close()
        for a, b in zip(*outs):
            assert torch.equal(a, b), (task, signal, epw)
        assert int(outs[0][2].sum()) >= 2 * 700


def test_policy_argument_checks(torch):
    from rex_gym_amd import RexBatchEnv, _lib
    plain = RexBatchEnv(8, task="walk", signal_type="ik", seed=1)                    # no range_normalize: the agents' wrapper stack is not folded
    with pytest.raises(_lib.RexSimError, match="range_normalize"):
        _actor(torch, plain)
    plain.close()
    mixed = RexBatchEnv(64, task="mixed", signal_type="ik", seed=1, range_normalize=True)
    with pytest.raises(_lib.RexSimError, match="single-task"):
        _actor(torch, mixed)
    mixed.close()
    env = RexBatchEnv(8, task="walk", signal_type="ik", seed=1, range_normalize=True, auto_reset=True)
    obs = env.reset()
    with pytest.raises(RuntimeError, match="no policy"):
        env.step_policy(obs)
    act = _actor(torch, env)
    with pytest.raises(ValueError):
        env.step_policy(obs[:4])                                                     # wrong shape
    with pytest.raises(ValueError):
        env.step_policy(obs.cpu())                                                   # host tensor
    with pytest.raises(ValueError):
        env.step_segment_policy(0, obs)
    with pytest.raises(ValueError):
        env.set_policy(act.w1.t().contiguous(), act.b1, act.w2, act.b2, act.w3, act.b3, act.logstd)   # output-major weights
    with pytest.raises(_lib.RexSimError, match="alias"):
        env.step_policy(obs, out=(obs, torch.zeros(8, device="cuda"), torch.zeros(8, dtype=torch.uint8, device="cuda")))
    env.set_event_trace(True)
    with pytest.raises(_lib.RexSimError, match="event trace"):
        env.step_policy(obs)
    env.set_event_trace(False)
    o, r, d, info = env.step_policy(obs)
    assert bool(torch.isfinite(o).all()) and info["action"] is None and info["policy_action"].shape == (8, 2)
    env.set_policy(None)
    with pytest.raises(RuntimeError, match="no policy"):
        env.step_policy(o)
    env.close()

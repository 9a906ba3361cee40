"""The agent's actor, handed to the simulator: closed-loop rollouts without a launch (or a host round trip) per step.

The reference's rollout loop is `action = algo.perform(prevob)` -> `batch_env.simulate(action)` for every step
(rex_gym/agents/tools/simulate.py:57-76, agents/ppo/algorithm.py:105-134).  `FusedActor` packs what perform() evaluates
-- the observ filter's statistics (agents/ppo/normalize.py:47-66) and the ForwardGaussianPolicy weights
(agents/scripts/networks.py:66-110) -- into the device buffers `RexBatchEnv.set_policy` hands to the library, in the
ABI's layout (input-major weight matrices), so that `env.step_policy` / `env.step_segment_policy` run perform() inside
the step launch (csrc/rex_policy.h).  `sync()` refreshes the buffers from the torch module and the filter and hands them to
the library again (which snapshots them: `rex_set_policy`): the learner calls it after every update (and whenever it wants the
filter statistics of the rollout refreshed); between two sync() calls the actor is frozen, as the reference's is between two
training phases.
"""
import torch


class FusedActor:
    def __init__(self, env, net, observ_filter=None, sample=True, seed=0):
        """env: a RexBatchEnv created with range_normalize=True; net: agents.ppo.ForwardGaussianPolicy (two hidden policy
        layers); observ_filter: agents.ppo.StreamingNormalize or None."""
        if getattr(net, "state_size", None):
            raise NotImplementedError("the fused actor evaluates ForwardGaussianPolicy (every shipped config); the recurrent "
                                      "policy keeps a per-env GRU state and runs through perform() / env.step()")
        lins = [m for m in net.policy if isinstance(m, torch.nn.Linear)]
        if len(lins) != 2:
            raise NotImplementedError("the fused actor is built for two hidden policy layers (configs.py:31: 200, 100)")
        self.env, self.net, self.filter = env, net, observ_filter
        self.l1, self.l2 = lins
        dev, O, A = env.device, env.obs_dim, env.action_dim
        h1, h2 = self.l1.out_features, self.l2.out_features
        f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        self.w1, self.b1, self.w2, self.b2, self.w3, self.b3, self.logstd = f(O, h1), f(h1), f(h1, h2), f(h2), f(h2, A), f(A), f(A)
        self.obs_mean, self.obs_scale = (f(O), f(O)) if observ_filter is not None else (None, None)
        self.obs_clip = float(observ_filter.clip) if observ_filter is not None and observ_filter.clip else 5.0
        self.sample, self.seed = bool(sample), int(seed)
        self.sync()

    @torch.no_grad()
    def sync(self):
        """copy the module's weights (transposed to input-major) and the filter's statistics into the kernel's buffers"""
        self.w1.copy_(self.l1.weight.t()); self.b1.copy_(self.l1.bias)
        self.w2.copy_(self.l2.weight.t()); self.b2.copy_(self.l2.bias)
        self.w3.copy_(self.net.mean.weight.t()); self.b3.copy_(self.net.mean.bias)
        self.logstd.copy_(self.net.logstd)
        if self.filter is not None:
            flt = self.filter
            self.obs_mean.copy_(flt.mean if flt.center else torch.zeros_like(flt.mean))
            if flt.scale and flt.count > 1:
                self.obs_scale.copy_(1.0 / (flt.std() + 1e-8))      # normalize.py:60-62
            else:
                self.obs_scale.fill_(1.0)
        # the library snapshots (packs) the arrays on the env's stream
        self.env.set_policy(self.w1, self.b1, self.w2, self.b2, self.w3, self.b3, self.logstd, self.obs_mean, self.obs_scale,
                            obs_clip=self.obs_clip, sample=self.sample, seed=self.seed)

    @torch.no_grad()
    def forward_reference(self, observ):
        """the kernel's arithmetic in plain torch fp32 on the packed buffers (tests): filtered observation -> mean"""
        x = observ
        if self.obs_mean is not None:
            x = ((x - self.obs_mean) * self.obs_scale).clamp(-self.obs_clip, self.obs_clip)
        h = torch.relu(x @ self.w1 + self.b1)
        h = torch.relu(h @ self.w2 + self.b2)
        return torch.tanh(h @ self.w3 + self.b3)

"""A batch whose envs run different tasks (BASELINE.json configs[4]: "mixed gaits + mass/friction domain
randomisation").  The reference gets a mixed batch by handing `BatchEnv` a list of different env objects
(`agents/tools/batch_env.py:21-44` only asks that they share the spaces); here the batch is partitioned into one
`RexBatchEnv` per task, each on its own HIP stream, so the per-task launches of a step overlap on the device.

Env g of the mixed batch is env `g - start[t]` of the sub-batch of its task t (contiguous ranges, in the order of
`tasks`), keeps the GLOBAL env index for its RNG stream, and pads its action / observation rows to the widest task
(actions beyond a task's own width are ignored, observation columns beyond it are zero) -- the same convention a
learner needs anyway to feed one policy.
"""
import numpy as np

from .batch_env import RexBatchEnv


class RexMixedBatchEnv:
    def __init__(self, num_envs, tasks=(("walk", "ik"), ("gallop", "ik"), ("turn", "ik")), weights=None, device=0, seed=0,
                 env_index_base=0, mass_scale_range=None, friction_range=None, **kwargs):
        """tasks: sequence of (task, signal_type); weights: fraction of the batch per task (default equal).
        mass_scale_range / friction_range: (lo, hi) of the per-env uniform draws applied with set_body_params, e.g.
        (0.8, 1.2) and (0.25, 0.625) (foot friction 0.5 x U(0.5, 1.25)); None leaves the URDF values."""
        import torch
        self._torch = torch
        self.num_envs = int(num_envs)
        self.tasks = tuple(tasks)
        w = np.ones(len(self.tasks)) if weights is None else np.asarray(weights, dtype=np.float64)
        counts = np.floor(w / w.sum() * self.num_envs).astype(int)
        counts[0] += self.num_envs - counts.sum()
        if np.any(counts <= 0):
            raise ValueError("every task needs at least one env")
        self.starts = np.concatenate([[0], np.cumsum(counts)]).astype(int)
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        self.streams = [torch.cuda.Stream(self.device) for _ in self.tasks]
        self.envs = []
        for (task, signal), a, n, st in zip(self.tasks, self.starts[:-1], counts, self.streams):
            self.envs.append(RexBatchEnv(int(n), task=task, signal_type=signal, device=device, seed=seed,
                                         env_index_base=int(env_index_base + a), stream=st, **kwargs))
        self.action_dim = max(e.action_dim for e in self.envs)
        self.obs_dim = max(e.obs_dim for e in self.envs)
        self.num_motors = self.envs[0].num_motors
        with torch.cuda.device(self.device):
            self._obs = torch.zeros((self.num_envs, self.obs_dim), device=self.device)
            self._reward = torch.zeros(self.num_envs, device=self.device)
            self._done = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
            self._cmd = torch.zeros((self.num_envs, self.num_motors), device=self.device)
        if mass_scale_range is not None or friction_range is not None:
            g = torch.Generator(device=self.device)
            g.manual_seed(int(seed) + 7919)
            for e in self.envs:
                def draw(rng):
                    return None if rng is None else torch.rand(e.num_envs, device=self.device, generator=g) * (rng[1] - rng[0]) + rng[0]
                m = draw(mass_scale_range)
                e.set_body_params(base_mass_scale=m, leg_mass_scale=m, foot_friction=draw(friction_range))
            torch.cuda.synchronize(self.device)

    def __len__(self):
        return self.num_envs

    def task_of(self, g):
        t = int(np.searchsorted(self.starts, g, side="right") - 1)
        return self.tasks[t], g - int(self.starts[t])

    def _join(self):
        cur = self._torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)

    def _fork(self):
        cur = self._torch.cuda.current_stream(self.device)
        for st in self.streams:
            st.wait_stream(cur)

    def reset(self):
        self._fork()
        for e, a, b, st in zip(self.envs, self.starts[:-1], self.starts[1:], self.streams):
            with self._torch.cuda.stream(st):
                self._obs[a:b, :e.obs_dim] = e.reset()
        self._join()
        return self._obs.clone()

    def step(self, actions):
        """actions [N, action_dim] device tensor; returns (obs [N, obs_dim], reward [N], done [N] bool, info)."""
        torch = self._torch
        a_all = torch.as_tensor(actions, dtype=torch.float32, device=self.device)
        if a_all.shape != (self.num_envs, self.action_dim):
            raise ValueError(f"actions must have shape {(self.num_envs, self.action_dim)}, got {tuple(a_all.shape)}")
        self._fork()
        for e, a, b, st in zip(self.envs, self.starts[:-1], self.starts[1:], self.streams):
            with torch.cuda.stream(st):
                o, r, d, info = e.step(a_all[a:b, :e.action_dim])
                self._obs[a:b, :e.obs_dim] = o
                self._reward[a:b] = r
                self._done[a:b] = d
                self._cmd[a:b] = info["action"]
        self._join()
        return self._obs, self._reward, self._done, {"action": self._cmd}

    def close(self):
        for e in self.envs:
            e.close()

"""Minimal `gym.spaces.Box` stand-in (gym is not a dependency of the hot path).

Only what the reference's callers use: low/high/shape/dtype, contains(), sample()
(agents/tools/batch_env.py:76-79, agents/ppo/simple_ppo_agent.py:78-88).
"""
import numpy as np


class Box:
    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        assert self.low.shape == self.high.shape
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)
        return [seed]

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    def sample(self):
        lo, hi = np.minimum(self.low, self.high), np.maximum(self.low, self.high)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    def __eq__(self, other):
        return isinstance(other, Box) and np.allclose(self.low, other.low) and np.allclose(self.high, other.high)

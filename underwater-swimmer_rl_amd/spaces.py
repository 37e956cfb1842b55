"""Minimal `Box` space (gymnasium is not a dependency of this package).

Same fields as `gymnasium.spaces.Box` for what the reference's consumers read:
`shape`, `dtype`, `low`, `high`, `sample()` (sb3_sac_agent.py:61-90 reads
`observation_space.shape[0]`, `action_space.shape[0]`, `action_space`).
If gymnasium is importable, `to_gymnasium()` returns the real thing.
"""
from __future__ import annotations

import math

import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    def __eq__(self, other):
        return (isinstance(other, Box) and self.shape == other.shape and self.dtype == other.dtype
                and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high))

    def to_gymnasium(self):
        from gymnasium import spaces  # raises ImportError when absent
        return spaces.Box(low=self.low, high=self.high, dtype=self.dtype.type)


def batch_space(space: Box, n: int) -> Box:
    return Box(np.broadcast_to(space.low, (n,) + space.shape), np.broadcast_to(space.high, (n,) + space.shape),
               shape=(n,) + space.shape, dtype=space.dtype)


def single_action_space(cfg) -> Box:
    """snake:69-74."""
    if cfg.forced_breathing:
        return Box(low=-1.0, high=1.0, shape=(1,), dtype=np.float32)
    return Box(low=np.array([0.0, -1.0]), high=np.array([1.0, 1.0]), dtype=np.float32)


def single_observation_space(cfg) -> Box:
    """snake:79-88."""
    food = cfg.max_observed_food * 4 + 2
    low = np.array([0, 0, -10, -10, -math.pi, -0.1, 0.5, 0, 0, -1] + [-1] * food)
    high = np.array([1, 1, 10, 10, math.pi, 0.1, 2.0, 2, 1, 1] + [1] * food)
    return Box(low=low, high=high, dtype=np.float32)

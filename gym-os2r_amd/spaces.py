"""Minimal ``Box`` space, duck-compatible with ``gym.spaces.Box`` for what the monopod task
uses (``low``, ``high``, ``dtype``, ``shape``, ``contains``, ``sample``, ``seed``).

The reference builds its spaces with ``gym.spaces.Box`` (gym_os2r/tasks/monopod.py:124,187,198);
gym is not a dependency of this package, so the few members the task touches live here.
"""
from __future__ import annotations

import numpy as np


class Box:
    def __init__(self, low, high, dtype=np.float64, seed=None):
        self.dtype = np.dtype(dtype)
        self.low = np.asarray(low, dtype=self.dtype).copy()
        self.high = np.asarray(high, dtype=self.dtype).copy()
        if self.low.shape != self.high.shape:
            raise ValueError("low and high must have the same shape")
        self.shape = self.low.shape
        self._rng = np.random.default_rng(seed)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]

    def contains(self, x) -> bool:
        if not isinstance(x, np.ndarray):
            try:
                x = np.asarray(x, dtype=self.dtype)
            except (ValueError, TypeError):
                return False
        return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self.shape
                    and np.all(x >= self.low) and np.all(x <= self.high))

    def __contains__(self, x) -> bool:
        return self.contains(x)

    def sample(self) -> np.ndarray:
        """Uniform inside bounded dimensions, normal where unbounded (as gym does)."""
        out = np.empty(self.shape, dtype=np.float64)
        lo_b, hi_b = np.isfinite(self.low), np.isfinite(self.high)
        both = lo_b & hi_b
        out[both] = self._rng.uniform(self.low[both], self.high[both])
        out[~lo_b & ~hi_b] = self._rng.normal(size=int((~lo_b & ~hi_b).sum()))
        only_lo = lo_b & ~hi_b
        out[only_lo] = self.low[only_lo] + self._rng.exponential(size=int(only_lo.sum()))
        only_hi = ~lo_b & hi_b
        out[only_hi] = self.high[only_hi] - self._rng.exponential(size=int(only_hi.sum()))
        return out.astype(self.dtype)

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"

    def __eq__(self, other):
        return (isinstance(other, Box) and self.shape == other.shape
                and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high))

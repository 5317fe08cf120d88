"""Minimal action/observation space descriptors.

The reference takes its spaces from gymnasium (`spaces.Discrete`, `spaces.Box`); gymnasium is not a
dependency of the hot path, and the wrappers in this package only ever need `.n`, `.shape`, `.sample()`
and equality (pettingzoo_env.py:50-67 of the reference compares the per-agent spaces).  Any object with
those attributes (e.g. a real gymnasium space) is accepted wherever these are.
"""
from __future__ import annotations

import numpy as np


class Discrete:
    def __init__(self, n: int, seed: int | None = None) -> None:
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64
        self._rng = np.random.default_rng(seed)

    def sample(self) -> int:
        return int(self._rng.integers(self.n))

    def contains(self, x) -> bool:
        return 0 <= int(x) < self.n

    def __eq__(self, o) -> bool:
        return hasattr(o, "n") and int(o.n) == self.n

    def __hash__(self) -> int:
        return hash(("Discrete", self.n))

    def __repr__(self) -> str:
        return f"Discrete({self.n})"


class Box:
    def __init__(self, low, high, shape, dtype=np.float32) -> None:
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def sample(self) -> np.ndarray:
        lo = np.broadcast_to(np.nan_to_num(np.asarray(self.low, np.float64), neginf=-1.0), self.shape)
        hi = np.broadcast_to(np.nan_to_num(np.asarray(self.high, np.float64), posinf=1.0), self.shape)
        return np.random.uniform(lo, hi).astype(self.dtype)

    def __eq__(self, o) -> bool:
        return hasattr(o, "shape") and not hasattr(o, "n") and tuple(o.shape) == self.shape

    def __hash__(self) -> int:
        return hash(("Box", self.shape))

    def __repr__(self) -> str:
        return f"Box({self.low}, {self.high}, {self.shape})"


def is_discrete(space) -> bool:
    """Duck-typed `isinstance(space, spaces.Discrete)` (pettingzoo_env.py:83, enhanced_pettingzoo_env.py:141)."""
    return hasattr(space, "n") and not hasattr(space, "nvec")

"""Gymnasium spaces when gymnasium is importable, otherwise tiny data holders
with gymnasium's field names (`low/high/shape/dtype`, `n`, mapping access,
`sample/seed/contains`).  The reference builds its spaces with gymnasium<1.0
(reference fov_env.py:125-142, atari_env.py:69-70)."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - gymnasium is absent from the build image
    from gymnasium.spaces import Box, Dict, Discrete, MultiDiscrete  # type: ignore
    from gymnasium.vector.utils import batch_space  # type: ignore
    HAVE_GYMNASIUM = True
except Exception:  # noqa: BLE001
    HAVE_GYMNASIUM = False

    class _Space:
        def __init__(self):
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            super().__init__()
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            self.shape = tuple(int(s) for s in shape)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

        def sample(self):
            if np.issubdtype(self.dtype, np.integer):
                return self._rng.integers(self.low, self.high + 1, size=self.shape).astype(self.dtype)
            return self._rng.uniform(self.low, self.high, size=self.shape).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Discrete(_Space):
        def __init__(self, n, start=0):
            super().__init__()
            self.n = int(n)
            self.start = int(start)
            self.shape = ()
            self.dtype = np.dtype(np.int64)

        def sample(self):
            return int(self.start + self._rng.integers(self.n))

        def contains(self, x):
            try:
                xi = int(x)
            except Exception:  # noqa: BLE001
                return False
            return self.start <= xi < self.start + self.n

        def __repr__(self):
            return f"Discrete({self.n})"

    class Dict(_Space, dict):
        def __init__(self, spaces=None, **kw):
            _Space.__init__(self)
            dict.__init__(self, spaces or {}, **kw)

        @property
        def spaces(self):
            return self

        def seed(self, seed=None):
            for i, s in enumerate(self.values()):
                s.seed(None if seed is None else seed + i)
            return [seed]

        def sample(self):
            return {k: s.sample() for k, s in self.items()}

        def contains(self, x):
            return isinstance(x, dict) and all(k in x and s.contains(x[k]) for k, s in self.items())

        def __repr__(self):
            return "Dict(" + ", ".join(f"{k!r}: {v!r}" for k, v in self.items()) + ")"

    class MultiDiscrete(_Space):
        def __init__(self, nvec, dtype=np.int64):
            super().__init__()
            self.nvec = np.asarray(nvec, dtype=dtype)
            self.shape = self.nvec.shape
            self.dtype = np.dtype(dtype)

        def sample(self):
            return (self._rng.random(self.shape) * self.nvec).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))

        def __repr__(self):
            return f"MultiDiscrete({self.nvec.tolist()})"

    def batch_space(space, n=1):
        """gymnasium.vector.utils.batch_space for the three space kinds used here: what SyncVectorEnv exposes as
        ``action_space`` / ``observation_space`` (the per-env spaces stay under ``single_*``)."""
        if isinstance(space, Box):
            rep = (n,) + (1,) * len(space.shape)
            return Box(low=np.tile(space.low, rep), high=np.tile(space.high, rep), dtype=space.dtype)
        if isinstance(space, Discrete):
            return MultiDiscrete(np.full((n,), space.n, dtype=np.int64))
        if isinstance(space, Dict):
            return Dict({k: batch_space(v, n) for k, v in space.items()})
        raise TypeError(f"cannot batch {space!r}")

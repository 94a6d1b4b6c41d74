"""Scripted stand-in for an ALE emulator, shared by the golden generator
(`tests/golden/make_golden.py`) and the parity tests.

It is NOT an Atari emulator: it is a deterministic event script with the
``atari_py.ALEInterface`` surface the reference touches (atari_env.py:44-52,
74,88-108,124,129,136,168), so that the reference's control flow (which frames
are sampled, zero fill, max, deque order, life-loss terminals, reset variants)
can be replayed identically by the oracle and by the product's host runner.
"""
from __future__ import annotations

import numpy as np


class ScriptedALE:
    """Every ``act`` advances one frame.  Rewards, life losses and game-overs
    are drawn from a private RNG keyed by ``seed``, so two instances with the
    same seed and the same call sequence behave identically."""

    def __init__(self, seed=0, screen_hw=(210, 160), n_actions=4, start_lives=3,
                 p_life=0.03, p_over=0.004, rgb=True):
        self.seed = seed
        self.hw = tuple(screen_hw)
        self.n_actions = n_actions
        self.start_lives = start_lives
        self.p_life, self.p_over = p_life, p_over
        self.rgb = rgb
        self._rng = np.random.default_rng(seed)
        self._lives = start_lives
        self._over = False
        self.frame = 0
        self.episode = 0
        self.calls = []            # trace of (name, arg) for debugging

    # --- configuration calls the reference makes (no-ops here) ---
    def setInt(self, *a):
        pass

    def setFloat(self, *a):
        pass

    def setBool(self, *a):
        pass

    def loadROM(self, *a):
        pass

    def getMinimalActionSet(self):
        return list(range(self.n_actions))

    # --- emulator surface ---
    def reset_game(self):
        self._lives = self.start_lives
        self._over = False
        self.episode += 1
        self.frame = 0

    def act(self, a):
        self.frame += 1
        u = self._rng.random(3)
        reward = 0
        if u[0] < 0.15:
            reward = int(self._rng.integers(-2, 8))
        if not self._over:
            if u[1] < self.p_life:
                self._lives -= 1
                if self._lives <= 0:
                    self._lives = 0
                    self._over = True
            if u[2] < self.p_over:
                self._over = True
        return reward

    def game_over(self):
        return self._over

    def lives(self):
        return self._lives

    def _screen_seed(self):
        return (self.seed * 1000003 + self.episode * 7919 + self.frame) & 0x7FFFFFFF

    def getScreenRGB(self):
        r = np.random.default_rng(self._screen_seed())
        h, w = self.hw
        # blocky content + noise so that resize/max are exercised non-trivially
        base = r.integers(0, 256, size=(h // 7 + 1, w // 5 + 1, 3), dtype=np.uint8)
        img = np.repeat(np.repeat(base, 7, axis=0), 5, axis=1)[:h, :w]
        noise = r.integers(0, 32, size=(h, w, 3), dtype=np.uint8)
        return (img // 2 + noise).astype(np.uint8)

    def getScreenGrayscale(self):
        """(H, W, 1) u8 like atari_py.  Uses a plain channel mean so it does
        not depend on the luminance restatement under test."""
        rgb = self.getScreenRGB().astype(np.uint16)
        return ((rgb[..., 0] + rgb[..., 1] + rgb[..., 2]) // 3).astype(np.uint8)[..., None]

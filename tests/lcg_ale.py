"""Python mirror of the native runner's "scripted" emulator (active-gym_amd/csrc/agx_runner.cpp: ScriptedEmu):
splitmix64 event script + arithmetic screens, so that the C++ runner and the Python runner can be compared
bit for bit on the CPU."""
import numpy as np

M64 = (1 << 64) - 1
_Y, _X = np.meshgrid(np.arange(210, dtype=np.uint32), np.arange(160, dtype=np.uint32), indexing="ij")
_BASE = _Y * 7 + _X * 13 + ((_Y * _X) >> 4)


class LcgALE:
    def __init__(self, seed, n_actions=4, start_lives=3, p_life=30, p_over=4):
        self.seed = int(seed)
        self.s = (self.seed * 0x9E3779B97F4A7C15 + 0x1234567) & M64
        self.n_actions, self.start_lives, self.p_life, self.p_over = n_actions, start_lives, p_life, p_over
        self._lives, self.frame, self.episode, self.over = start_lives, 0, 0, False

    def _rnd(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)

    def getMinimalActionSet(self):
        return list(range(self.n_actions))

    def act(self, a):
        self.frame += 1
        u0, u1, u2 = self._rnd(), self._rnd(), self._rnd()
        reward = 0
        if u0 % 100 < 15:
            reward = int(self._rnd() % 10) - 2
        if not self.over:
            if u1 % 1000 < self.p_life:
                self._lives -= 1
                if self._lives <= 0:
                    self._lives = 0
                    self.over = True
            if u2 % 1000 < self.p_over:
                self.over = True
        return reward

    def game_over(self):
        return self.over

    def lives(self):
        return self._lives

    def reset_game(self):
        self._lives, self.over, self.frame = self.start_lives, False, 0
        self.episode += 1

    def getScreenRGB(self):
        K = (self.seed * 1000003 + self.episode * 7919 + self.frame * 31) & 0xFFFF
        base = _BASE + np.uint32(K * 3)
        out = np.empty((210, 160, 3), np.uint8)
        out[..., 0] = base & 0xFF
        out[..., 1] = (base + 29) & 0xFF
        out[..., 2] = (base + 58 + (K >> 3)) & 0xFF
        return out

    def getScreenGrayscale(self):
        """(210, 160, 1) like atari_py: ALE's luminance of these RGB values, C-double arithmetic in ALE's order."""
        rgb = self.getScreenRGB().astype(np.float64)
        x = (rgb[..., 0] * 0.2989 + rgb[..., 1] * 0.5870) + rgb[..., 2] * 0.1140
        fl = np.floor(x)
        return (fl + ((x - fl) >= 0.5)).astype(np.uint8)[..., None]

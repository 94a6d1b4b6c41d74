"""Emulator backends behind the host runner.  An emulator is any object with
the slice of the ``atari_py.ALEInterface`` surface the reference touches
(reference atari_env.py:44-52,88-108,124,129,136,168):

    getMinimalActionSet() act(a) game_over() lives() reset_game() getScreenRGB()

* ``"ale"``       real ALE through ``atari_py`` (what the reference imports) or
                  ``ale_py`` when only that is installed; configured exactly as
                  reference atari_env.py:44-50.  Raises ImportError when
                  neither is importable — there is no silent substitute.
* ``"synthetic"`` :class:`SyntheticALE`, a deterministic procedural stand-in
                  (moving sprites, score events, lives) for plumbing tests and
                  benchmarks on machines without ALE/ROMs.  Not an Atari game.
* a callable      ``factory(args, index) -> emulator`` (tests inject scripted ones).
"""
from __future__ import annotations

import numpy as np

RAW_H, RAW_W = 210, 160


def resolve_frame_format(args, real=None) -> str:
    """"rgb" | "gray": which screens travel from the emulators to the device.  An explicit ``args.frame_format`` wins.
    Default: real emulators (``frame_source`` "ale" / "native:ale") hand over ALE's OWN grayscale screens - exactly what
    the reference reads (``getScreenGrayscale``, atari_env.py:74): ALE's palette table has done the luminance, a third of
    the PCIe bytes, and no restated luminance arithmetic on the device.  Everything else (synthetic / scripted sources,
    factories) defaults to raw RGB screens with the luminance on the GPU, the workload BASELINE.json's metric names."""
    fmt = getattr(args, "frame_format", None)
    if fmt is None:
        if real is None:                         # `real`: the caller already knows whether a real ALE sits behind it
            src = getattr(args, "frame_source", "ale")
            real = isinstance(src, str) and src in ("ale", "native:ale")
        fmt = "gray" if real else "rgb"
    if fmt not in ("rgb", "gray"):
        raise ValueError("frame_format must be 'rgb' (getScreenRGB, luminance on the device) or 'gray' (getScreenGrayscale)")
    return fmt


class SyntheticALE:
    """Procedural 210x160 RGB screens with the ALE call surface."""

    def __init__(self, game="synthetic", seed=0, n_actions=4, start_lives=3, max_frames=4000):
        self.game = game
        self._seed = int(seed)
        self.n_actions = n_actions
        self.start_lives = start_lives
        self.max_frames = max_frames
        self._rng = np.random.default_rng(self._seed)
        import zlib
        # a stable hash of the game name: Python's str hash is salted per process, and the ranks of a sharded run must
        # agree on the palette
        pal = np.random.default_rng(zlib.crc32(str(game).encode()) & 0xFFFF).integers(0, 256, size=(16, 3), dtype=np.uint8)
        self._palette = pal
        self._screen = np.zeros((RAW_H, RAW_W, 3), np.uint8)
        self.reset_game()

    # configuration calls of the reference: accepted and ignored
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

    def reset_game(self):
        self._lives = self.start_lives
        self._over = False
        self._frame = 0
        self._pos = self._rng.integers(0, [RAW_H - 24, RAW_W - 16], size=(6, 2)).astype(np.int64)
        self._vel = self._rng.integers(-3, 4, size=(6, 2)).astype(np.int64)

    def act(self, a):
        if self._over:
            return 0
        self._frame += 1
        self._vel[0] += np.array([(a % 3) - 1, (a // 3) - 1])
        self._pos += self._vel
        lim = np.array([RAW_H - 24, RAW_W - 16])
        bounce = (self._pos < 0) | (self._pos > lim)
        self._vel[bounce] *= -1
        self._pos = np.clip(self._pos, 0, lim)
        u = self._rng.random(2)
        reward = int(self._rng.integers(1, 11)) if u[0] < 0.05 else 0
        if u[1] < 0.004:
            self._lives -= 1
            if self._lives <= 0:
                self._lives = 0
                self._over = True
        if self._frame >= self.max_frames:
            self._over = True
        return reward

    def game_over(self):
        return self._over

    def lives(self):
        return self._lives

    def getScreenRGB(self, out=None):
        s = self._screen if out is None else out
        s[...] = self._palette[0]
        s[:16] = self._palette[1]
        for k in range(self._pos.shape[0]):
            r, c = self._pos[k]
            s[r:r + 24, c:c + 16] = self._palette[2 + k]
        return s

    def getScreenGrayscale(self):
        """(210, 160, 1) u8: ALE's palette luminance of the RGB screen, round(.2989 r + .5870 g + .1140 b)."""
        rgb = self.getScreenRGB().astype(np.float64)
        x = (rgb[..., 0] * 0.2989 + rgb[..., 1] * 0.5870) + rgb[..., 2] * 0.1140
        fl = np.floor(x)
        return (fl + ((x - fl) >= 0.5)).astype(np.uint8)[..., None]


def _make_ale(args, index):
    """Real ALE, configured as reference atari_env.py:44-50."""
    ale = None
    rom = None
    try:
        import atari_py  # type: ignore
        ale = atari_py.ALEInterface()
        rom = atari_py.get_game_path(args.game)
    except ImportError:
        try:
            import ale_py  # type: ignore
            ale = ale_py.ALEInterface()
            name = "".join(p.capitalize() for p in str(args.game).split("_"))
            rom = getattr(ale_py.roms, name, None) or ale_py.roms.get_rom_path(args.game)
        except ImportError as e:
            raise ImportError(
                "frame_source='ale' needs atari_py (as the reference does) or ale_py; neither is importable. "
                "Pass frame_source='synthetic' (procedural stand-in) or an emulator factory explicitly.") from e
    ale.setInt("random_seed", int(args.seed) + index)
    ale.setInt("max_num_frames_per_episode", int(args.max_episode_length))   # reference passes the float 108e3
    ale.setFloat("repeat_action_probability", 0)
    ale.setInt("frame_skip", 0)
    ale.setBool("color_averaging", False)
    ale.loadROM(rom)
    return ale


def make_emulator(args, index=0):
    src = getattr(args, "frame_source", "ale")
    if callable(src):
        return src(args, index)
    if src == "ale":
        return _make_ale(args, index)
    if src == "synthetic":
        return SyntheticALE(game=args.game, seed=int(args.seed) + index)
    raise ValueError(f"unknown frame_source {src!r}")

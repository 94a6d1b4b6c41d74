"""Single-env drop-in surface of the reference's ``active_gym/atari_env.py``:
``AtariEnvArgs``, ``AtariEnv`` and the four factories (reference
atari_env.py:25-39, 41-172, 174-192).  Every env is a thin view over a
one-env :class:`~active_gym.vector.AtariVecEnv`; the image work runs in the HIP
kernels of libagx.so.  Deviations from the reference are listed in DESIGN.md
(float32 observations instead of accidental float64, square obs_size check)."""
from __future__ import annotations

from typing import Tuple

import numpy as np

from .vector import AtariVecEnv


class AtariEnvArgs:
    """Same attribute bag as the reference (atari_env.py:25-39): every keyword becomes an attribute;
    ``fov_size, fov_init_loc, sensory_action_mode, resize_to_full`` (+ ``sensory_action_space`` for
    relative mode, ``peripheral_res`` for the peripheral env) have no defaults there either.
    Additional optional attributes understood here: ``frame_source``, ``antialias``, ``num_workers``;
    ``device`` (None in the reference and unused) selects NumPy (None) or device-tensor outputs."""

    def __init__(self, game, seed, obs_size: Tuple[int, int], **kwargs):
        self.env_backend = "atari_py"
        self.device = None
        self.seed = seed
        self.max_episode_length = 108e3
        self.game = game
        self.frame_stack = 4
        self.action_repeat = 4
        self.obs_size = obs_size
        self.mask_out = False
        self.record = False
        self.clip_reward = False
        for k, v in kwargs.items():
            self.__setattr__(k, v)


class _SingleEnv:
    """Shared single-env plumbing over an N=1 core."""

    metadata = {"render_modes": []}
    render_mode = None
    reward_range = (-float("inf"), float("inf"))
    spec = None

    def _scalar_info(self, infos, keys):
        out = {}
        for k in keys:
            if k in infos:
                v = infos[k][0]
                if hasattr(v, "detach"):                      # device outputs: a torch tensor row
                    v = v.detach().cpu().numpy()
                out[k] = v.copy() if isinstance(v, np.ndarray) else (v.item() if hasattr(v, "item") else v)
        return out

    @property
    def unwrapped(self):
        e = self
        while hasattr(e, "env"):
            e = e.env
        return e

    def close(self):
        pass


class AtariEnv(_SingleEnv):
    """``AtariEnv`` (reference atari_env.py:41-172)."""

    def __init__(self, args, _kind="base"):
        self.args = args
        self.seed_num = args.seed
        self._core = AtariVecEnv(args, 1, kind=_kind, autoreset=False)
        self.frame_stack = self._core.frame_stack
        self.action_repeat = self._core.action_repeat
        self.obs_size = self._core.obs_size
        self.clip_reward = bool(args.clip_reward)
        self.actions = dict(enumerate(self._core.runner.actions[0]))
        from .spaces import Box, Discrete
        self.action_space = Discrete(len(self.actions))
        self.observation_space = Box(low=-1., high=1., shape=(self.frame_stack,) + self.obs_size, dtype=np.float32)

    def _rekind(self, kind):
        """A fovea wrapper switches the core to its own kernel kind (before the first reset); the emulator is kept."""
        return self._core.rekind(kind)

    @property
    def training(self):
        return self._core.runner.training

    @property
    def lives(self):
        return int(self._core.runner.lives[0])

    @property
    def life_termination(self):
        return bool(self._core.runner.life_termination[0])

    def reset(self, seed=None, options=None):
        obs, infos = self._core.reset()
        return self._first(obs), self._scalar_info(infos, ("raw_reward",))

    def step(self, action):
        obs, r, d, t, infos = self._core.step(np.asarray([action]) if self._core.kind == "base" else action)
        return self._first(obs), r[0].item(), bool(d[0]), False, self._scalar_info(infos, ("raw_reward",))

    def _first(self, obs):
        return obs[0]

    def train(self):
        self._core.train()

    def eval(self):
        self._core.eval()

    def render(self, mode="rgb_array", obs_size=None):
        assert mode == "rgb_array", "only support rgb_array mode, given %s" % mode
        rgb = self._core.render(0)
        size = obs_size if obs_size else (256, 256)
        from .record import resize_rgb_linear
        return resize_rgb_linear(rgb, size)

    def close(self):
        self._core.close()


def AtariBaseEnv(args: AtariEnvArgs):
    from .fov_env import RecordWrapper
    return RecordWrapper(AtariEnv(args), args)


def AtariFixedFovealEnv(args: AtariEnvArgs):
    from .fov_env import FixedFovealEnv
    return FixedFovealEnv(AtariBaseEnv(args), args)


def AtariFlexibleFovealEnv(args: AtariEnvArgs):
    from .fov_env import FlexibleFovealEnv
    return FlexibleFovealEnv(AtariBaseEnv(args), args)


def AtariFixedFovealPeripheralEnv(args: AtariEnvArgs):
    from .fov_env import FixedFovealPeripheralEnv
    return FixedFovealPeripheralEnv(AtariBaseEnv(args), args)

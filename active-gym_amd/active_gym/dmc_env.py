"""DMC pixel variant of the same observation path (reference dmc_env.py:58-273): dm_control physics renders an
obs-sized RGB image per step on the host; ``cv2.cvtColor(obs, COLOR_BGR2GRAY)`` (applied by the reference to an RGB
image, so channel 0 carries the blue weight), ``/255`` and the frame stack happen on the GPU (``agx_ingest_rgb``),
and the fovea kernels K2-K4 are the Atari ones unchanged.

dm_control is not a dependency: ``args.frame_source`` may be a factory ``(args, env_index) -> environment`` with the
dm_control surface used by the reference (``reset() / step(a) -> time_step(.reward .discount .observation .last())``,
``physics.render(height, width, camera_id)``, ``physics.get_state()``, ``action_spec()``, ``observation_spec()``);
by default ``dm_control.suite.load`` is used exactly as the reference does (dmc_env.py:102-108) and its absence is
an ImportError.  Only ``from_pixels=True, grey=True`` (the reference's defaults, dmc_env.py:68-69) is built."""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native as nat
from .atari_env import _SingleEnv
from .spaces import Box
from .vector import AtariVecEnv


class DMCEnvArgs:
    """Same attribute bag as the reference (dmc_env.py:58-77)."""

    def __init__(self, domain_name: str, task_name: str, seed: int, obs_size: Tuple[int, int], **kwargs):
        self.env_backend = "dmc"
        self.seed = seed
        self.domain_name = domain_name
        self.task_name = task_name
        self.obs_size = obs_size
        self.task_kwargs = {}
        self.visualize_reward = False
        self.from_pixels = True
        self.grey = True
        self.camera_id = 0
        self.action_repeat = 4
        self.frame_stack = 3
        self.mask_out = False
        self.environment_kwargs = {}
        self.clip_reward = False
        self.record = False
        self.device = None
        for k, v in kwargs.items():
            self.__setattr__(k, v)


def _flatten_obs(obs) -> np.ndarray:                                  # dmc_env.py:50-56
    pieces = []
    for v in obs.values():
        pieces.append(np.array([v]) if np.isscalar(v) else np.asarray(v).ravel())
    return np.concatenate(pieces, axis=0)


def _bounds(spec, dtype):
    """`_spec_to_box` (dmc_env.py:27-47) without dm_env.specs: a spec with minimum/maximum is bounded."""
    mins, maxs = [], []
    for s in spec:
        dim = int(np.prod(s.shape))
        if hasattr(s, "minimum") and hasattr(s, "maximum"):
            z = np.zeros(dim, dtype=np.float32)
            mins.append(s.minimum + z)
            maxs.append(s.maximum + z)
        else:
            b = np.inf * np.ones(dim, dtype=np.float32)
            mins.append(-b)
            maxs.append(b)
    return np.concatenate(mins).astype(dtype), np.concatenate(maxs).astype(dtype)


def _make_dmc(args, index):
    src = getattr(args, "frame_source", None)
    if callable(src):
        return src(args, index)
    try:
        from dm_control import suite  # type: ignore
    except ImportError as e:
        raise ImportError("DMC envs need dm_control (or args.frame_source = a factory of dm_control-like environments)") from e
    kw = dict(args.task_kwargs)
    kw["random"] = args.seed + index                                 # dmc_env.py:96 (per-env seed for a batch)
    return suite.load(domain_name=args.domain_name, task_name=args.task_name, task_kwargs=kw,
                      visualize_reward=args.visualize_reward, environment_kwargs=args.environment_kwargs)


class DMCHostRunner:
    """Host half of N DMC envs: the reference's ``DMCEnv._step/_reset`` minus the image arithmetic
    (dmc_env.py:199-234).  One obs-sized RGB render per env per step goes into ``frames`` u8 [N, H, W, 3]."""

    def __init__(self, args, num_envs: int, frames: np.ndarray, workers: Optional[int] = None, env_offset: int = 0):
        self.args = args
        self.num_envs = int(num_envs)
        self.action_repeat = int(args.action_repeat)
        self.clip_reward = bool(args.clip_reward)
        self.camera_id = int(args.camera_id)
        self.obs_size = tuple(int(v) for v in args.obs_size)
        self.envs = [_make_dmc(args, env_offset + i) for i in range(self.num_envs)]
        lo, hi = _bounds([self.envs[0].action_spec()], np.float32)
        self.true_low, self.true_high = lo, hi                         # dmc_env.py:111
        self.action_dim = lo.shape[0]
        self.frames = frames
        self.current_state = [None] * self.num_envs
        self.internal_state = [None] * self.num_envs
        self.discount = np.empty(self.num_envs, dtype=object)
        n_workers = workers if workers is not None else min(self.num_envs, os.cpu_count() or 1, 64)
        self._pool = ThreadPoolExecutor(max_workers=n_workers) if n_workers > 1 and self.num_envs > 1 else None
        self.training = True

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    def _map(self, fn, items):
        if self._pool is None:
            return [fn(i) for i in items]
        return list(self._pool.map(fn, items))

    def _render(self, i, out):
        h, w = self.obs_size
        out[...] = self.envs[i].physics.render(height=h, width=w, camera_id=self.camera_id)   # dmc_env.py:178-180

    def _after(self, i, ts):
        self.current_state[i] = _flatten_obs(ts.observation)
        self.internal_state[i] = self.envs[i].physics.get_state().copy()                     # dmc_env.py:190
        self.discount[i] = ts.discount

    def convert_action(self, action) -> np.ndarray:                                            # dmc_env.py:166-173
        action = np.asarray(action).astype(np.float64)
        true_delta = self.true_high - self.true_low
        norm_delta = np.float32(1.0) - np.float32(-1.0)
        action = (action - np.float32(-1.0)) / norm_delta
        action = action * true_delta + self.true_low
        return action.astype(np.float32)

    def step(self, actions):
        actions = np.asarray(actions)
        # `assert self._norm_action_space.contains(action)` (dmc_env.py:216): float32 Box(-1, 1, action_dim)
        assert actions.shape == (self.num_envs, self.action_dim) and np.all(actions >= -1.0) and np.all(actions <= 1.0), \
            "motor actions must lie in the normalised action space [-1, 1]^%d" % self.action_dim
        true = self.convert_action(actions)
        assert np.all(true >= self.true_low) and np.all(true <= self.true_high)                # dmc_env.py:218
        raw = np.zeros(self.num_envs, np.float64)
        done = np.zeros(self.num_envs, bool)

        def one(i):
            reward = 0
            for _ in range(self.action_repeat):                                               # dmc_env.py:222-227
                ts = self.envs[i].step(true[i])
                reward += ts.reward or 0
                d = ts.last()
                if d:
                    break
            self._render(i, self.frames[i])
            self._after(i, ts)
            raw[i], done[i] = reward, d

        self._map(one, range(self.num_envs))
        reward = np.sign(raw) if self.clip_reward else raw.copy()                            # dmc_env.py:232
        return reward, done, np.full(self.num_envs, 1, np.uint8), raw

    def reset(self, idx: Optional[Sequence[int]] = None, out: Optional[np.ndarray] = None, packed: bool = False) -> np.ndarray:
        idx = list(range(self.num_envs)) if idx is None else [int(i) for i in idx]
        buf = self.frames if out is None else out
        cmd = np.full(self.num_envs, nat.CMD_SKIP, np.uint8)
        row_of = {i: j for j, i in enumerate(idx)} if packed else None     # packed: the j-th reset env's render in row j

        def one(i):
            ts = self.envs[i].reset()                                                         # dmc_env.py:204
            self._render(i, buf[i if row_of is None else row_of[i]])
            self._after(i, ts)
            cmd[i] = nat.CMD_CLEAR | 1                                                        # _reset_buffer + one append

        self._map(one, idx)
        return cmd

    def render(self, i=0, obs_size=None, camera_id=0):                                        # dmc_env.py:243-251
        h, w = obs_size if obs_size is not None else self.obs_size
        return self.envs[i].physics.render(height=h, width=w, camera_id=camera_id or self.camera_id)


class DMCVecEnv(AtariVecEnv):
    """N DMC pixel envs of one kind on one GPU; same conventions as :class:`AtariVecEnv`, continuous motor
    actions ``(N, action_dim)`` in [-1, 1]; ``args.gray_mode`` = "cv15" (OpenCV 4.x, default) | "cv14"."""

    def _check_obs_size(self):
        if not (getattr(self.args, "from_pixels", True) and getattr(self.args, "grey", True)):
            raise NotImplementedError("only the reference's default from_pixels=True, grey=True DMC path is built")
        if (self.obs_size[0] * self.obs_size[1]) % 4:
            raise ValueError("obs_size must have a pixel count divisible by 4")

    def _setup_source(self, args, noop_fn, env_offset):
        h, w = self.obs_size
        shape = (self.num_envs, h, w, 3)
        self._h_frames = torch.empty(shape, dtype=torch.uint8, pin_memory=True)
        self._d_frames = torch.empty(shape, dtype=torch.uint8, device=self.device)
        self._h_rframes = torch.empty(shape, dtype=torch.uint8, pin_memory=True)
        self._h_rcmd = torch.empty((self.num_envs,), dtype=torch.uint8, pin_memory=True)
        self._h_cmd = torch.empty((self.num_envs,), dtype=torch.uint8, pin_memory=True)
        self._d_cmd = torch.empty((self.num_envs,), dtype=torch.uint8, device=self.device)
        self._ev_copy = torch.cuda.Event()
        self._alloc_reset_buffers()
        self.runner = DMCHostRunner(args, self.num_envs, self._h_frames.numpy(), workers=getattr(args, "num_workers", None),
                                    env_offset=env_offset)
        mode = getattr(args, "gray_mode", "cv15")
        if mode not in ("cv15", "cv14"):
            raise ValueError("gray_mode must be 'cv15' or 'cv14'")
        self._gray_mode = nat.GRAY_CV15 if mode == "cv15" else nat.GRAY_CV14

    def _motor_space(self):
        return Box(low=-1.0, high=1.0, shape=(self.runner.action_dim,), dtype=np.float32)      # dmc_env.py:112-117

    def _ingest(self, cmd=None):
        self.pipe.ingest_rgb(self._d_frames, self._d_cmd if cmd is None else cmd, self._gray_mode)

    def _h_reset_rows(self, buf=None):
        return self._h_rframes if buf is None else buf

    def _d_reset_target(self):
        return self._d_frames

    def _extra_info(self, info):                                                               # dmc_env.py:189-192
        info["internal_state"] = np.stack([np.asarray(s) for s in self.runner.internal_state])
        info["discount"] = self.runner.discount.copy()
        return info

    def render(self, index=0, obs_size=None, camera_id=0):
        return self.runner.render(index, obs_size, camera_id)


class DMCEnv(_SingleEnv):
    """``DMCEnv`` (reference dmc_env.py:79-254), a view over an N=1 :class:`DMCVecEnv`."""

    def __init__(self, args, _kind="base"):
        self.args = args
        self.seed_num = args.seed
        self._core = DMCVecEnv(args, 1, kind=_kind, autoreset=False)
        self.from_pixels, self.grey = True, True
        self.obs_size = self._core.obs_size
        self.camera_id = int(args.camera_id)
        self.action_repeat = self._core.action_repeat
        self.frame_stack = self._core.frame_stack
        self.clip_reward = bool(args.clip_reward)
        r = self._core.runner
        self._true_action_space = Box(r.true_low, r.true_high, dtype=np.float32)
        self._norm_action_space = self._core.single_motor_space
        self._observation_space = Box(low=-1., high=1., shape=(self.frame_stack,) + self.obs_size, dtype=np.float32)
        lo, hi = _bounds(r.envs[0].observation_spec().values(), np.float32)
        self._state_space = Box(lo, hi, dtype=np.float32)

    def _rekind(self, kind):
        return self._core.rekind(kind)

    observation_space = property(lambda self: self._observation_space)
    state_space = property(lambda self: self._state_space)
    action_space = property(lambda self: self._norm_action_space)
    reward_range = property(lambda self: (0, self.action_repeat))
    current_state = property(lambda self: self._core.runner.current_state[0])
    dmc_env = property(lambda self: self._core.runner.envs[0])

    _KEYS = ("internal_state", "discount", "raw_reward")

    def reset(self, seed=None, options=None):
        obs, infos = self._core.reset()
        return obs[0], self._scalar_info(infos, self._KEYS)

    def step(self, action):
        a = np.asarray(action)[None]
        obs, r, d, t, infos = self._core.step(a if self._core.kind == "base" else action)
        return obs[0], r[0].item(), bool(d[0]), False, self._scalar_info(infos, self._KEYS)

    def render(self, mode="rgb_array", obs_size=None, camera_id=0):
        assert mode == "rgb_array", "only support rgb_array mode, given %s" % mode
        return self._core.render(0, obs_size, camera_id)

    def close(self):
        self._core.close()


def DMCBaseEnv(args: DMCEnvArgs):
    from .fov_env import RecordWrapper
    return RecordWrapper(DMCEnv(args), args)


def DMCFixedFovealEnv(args: DMCEnvArgs):
    from .fov_env import FixedFovealEnv
    return FixedFovealEnv(DMCBaseEnv(args), args)


def DMCFlexibleFovealEnv(args: DMCEnvArgs):
    from .fov_env import FlexibleFovealEnv
    return FlexibleFovealEnv(DMCBaseEnv(args), args)


def DMCFixedFovealPeripheralEnv(args: DMCEnvArgs):
    from .fov_env import FixedFovealPeripheralEnv
    return FixedFovealPeripheralEnv(DMCBaseEnv(args), args)

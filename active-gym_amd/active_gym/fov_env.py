"""Single-env drop-in surface of the reference's ``active_gym/fov_env.py``:
``RecordWrapper``, ``FixedFovealEnv``, ``FlexibleFovealEnv`` (+ action type
enum) and ``FixedFovealPeripheralEnv`` (reference fov_env.py:15-105, 107-234,
236-355, 358-388).

The wrappers keep the reference's constructor shape ``Wrapper(env, args)`` and
its ``reset()/step(dict)`` results, but they do not crop NumPy arrays: they
switch the one-env device core underneath to their kernel kind, so the crop /
mask / resize / peripheral arithmetic runs in libagx.so."""
from __future__ import annotations

import copy

from enum import IntEnum

import numpy as np
import torch

from .atari_env import AtariEnv, _SingleEnv
from .spaces import Box, Dict, Discrete


class RecordWrapper(_SingleEnv):
    """Cumulative (unclipped) reward and episode length into ``info`` (reference fov_env.py:29-67).
    Trajectory recording (``args.record``) is provided by :mod:`active_gym.record`."""

    def __init__(self, env, args):
        self.env = env
        self.args = args
        self.record = bool(args.record)
        self.record_buffer = None
        self.prev_record_buffer = None
        from .record import Recorder
        self._rec = Recorder(self) if self.record else None

    def __getattr__(self, name):
        if name.startswith("__") or name in ("env",):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def cumulative_reward(self):
        return self.unwrapped._core.cumulative_reward[0].item()

    @property
    def ep_len(self):
        return int(self.unwrapped._core.ep_len[0])

    def _core(self):
        return self.unwrapped._core

    def _base_keys(self):
        """info keys the base env contributes besides raw_reward (DMC: internal_state, discount)."""
        return tuple(k for k in getattr(self.unwrapped, "_KEYS", ()) if k != "raw_reward")

    def reset(self, seed=None, options=None):
        obs, infos = self._core().reset()
        info = self._scalar_info(infos, ("raw_reward", "reward", "ep_len") + self._base_keys())
        if self._rec:
            self._rec.on_reset(obs[0], info)
        return obs[0], info

    def step(self, action):
        core = self._core()
        obs, r, d, t, infos = core.step(np.asarray([action]) if core.kind == "base" else action)
        info = self._scalar_info(infos, ("raw_reward", "reward", "ep_len") + self._base_keys())
        ret = r[0].item()
        if self._rec:
            self._rec.on_step(obs[0], action, info["reward"], bool(d[0]), False, info, ret)
        return obs[0], ret, bool(d[0]), False, info

    def save_record_to_file(self, file_path: str):
        if self._rec:
            self._rec.save(file_path)

    def render(self, **kwargs):
        return self.env.render(**kwargs)


class FixedFovealEnv(_SingleEnv):
    """``FixedFovealEnv`` (reference fov_env.py:107-234)."""
    _KIND = "fixed"

    def __init__(self, env, args):
        self.env = env
        self.args = args
        base = self.unwrapped
        if not hasattr(base, "_rekind"):
            raise TypeError("the fovea wrappers drive the libagx device core and need an active_gym AtariEnv or DMCEnv "
                            "underneath (other simulators are not wired to the HIP pipeline)")
        self.fov_size = tuple(args.fov_size)
        self.fov_init_loc = tuple(args.fov_init_loc)
        assert (np.array(self.fov_size) < np.array(base.obs_size)).all()         # fov_env.py:112
        self.sensory_action_mode = args.sensory_action_mode
        if self._KIND == "peripheral":
            self.peripheral_res = tuple(args.peripheral_res)
        core = base._rekind(self._KIND)
        self.sensory_action_space = core.sensory_action_space
        self.mask_out = core.mask_out
        self.resize_to_full = core.resize_to_full
        self.action_space = Dict({"motor_action": base.action_space,
                                  "sensory_action": Box(low=self.sensory_action_space[0],
                                                        high=self.sensory_action_space[1], dtype=int)})
        # declared exactly as the reference declares it (fov_env.py:132-142): the flexible env inherits the
        # (fs,) + fov_size declaration for its crop mode although its crops are ragged (fov_env.py:283-286)
        crop = self._KIND != "peripheral" and not (self.mask_out or self.resize_to_full)
        self.observation_space = Box(low=-1., high=1., dtype=np.float32,
                                     shape=(core.frame_stack,) + (tuple(self.fov_size) if crop else tuple(core.obs_size)))
        self.fov_loc = np.rint(np.array(self.fov_init_loc, copy=True)).astype(np.int32)   # fov_env.py:149-150

    def __getattr__(self, name):
        if name.startswith("__") or name in ("env",):
            raise AttributeError(name)
        return getattr(self.env, name)

    _INFO_KEYS = ("raw_reward", "reward", "ep_len", "fov_loc")

    def _core(self):
        return self.unwrapped._core

    def _base_keys(self):
        return tuple(k for k in getattr(self.unwrapped, "_KEYS", ()) if k != "raw_reward")

    def _obs(self, obs, info):
        return obs[0]

    def _rec(self):
        e = self.env
        while e is not None:
            if isinstance(e, RecordWrapper):
                return e._rec
            e = getattr(e, "env", None)
        return None

    def reset(self):                                               # takes no arguments, fov_env.py:156
        obs, infos = self._core().reset()
        info = self._scalar_info(infos, self._INFO_KEYS + self._base_keys())
        self._sync(info)
        o = self._obs(obs, info)
        rec = self._rec()
        if rec:
            rec.on_reset(self._full_state(), info, fovea=self)
        return o, info

    def _full_state(self):
        """The wrapped env's full-frame state (what the reference's RecordWrapper sits on and records,
        fov_env.py:59,74): float64 [fs, H, W] from the device ring.  Recording only."""
        return self._core().pipe.observe_full()[0].cpu().numpy().astype(np.float64)

    def _sync(self, info):
        self.fov_loc = info["fov_loc"]

    def _action(self, action):
        m = action["motor_action"]
        if isinstance(m, torch.Tensor):
            m = m.detach().cpu().numpy()
        discrete = hasattr(self._core().single_motor_space, "n")          # Atari: one index; DMC: a Box vector
        a = {"motor_action": np.asarray(m).reshape(1) if discrete else np.asarray(m).reshape(1, -1),
             "sensory_action": self._one(action["sensory_action"], 2)}
        return a

    @staticmethod
    def _one(x, cols):
        # a scalar (what the reference's scalar sensory_action Box samples) is broadcast to every column, as
        # np.clip(loc, 0, obs - fov) does in the reference (fov_env.py:166-167)
        if isinstance(x, torch.Tensor):
            x = x.detach()
            if cols and x.numel() == 1:
                x = x.reshape(1, 1).expand(1, cols)
            return x.reshape(1, cols) if cols else x.reshape(1)
        x = np.asarray(x)
        if cols and x.size == 1:
            x = np.broadcast_to(x.reshape(1, 1), (1, cols))
        return x.reshape(1, cols) if cols else x.reshape(-1)[:1]

    def step(self, action):
        obs, r, d, t, infos = self._core().step(self._action(action))
        info = self._scalar_info(infos, self._INFO_KEYS + self._base_keys())
        self._sync(info)
        o = self._obs(obs, info)
        ret = r[0].item()
        rec = self._rec()
        if rec:
            rec.on_step(self._full_state(), action["motor_action"], info["reward"], bool(d[0]), False, info, ret, fovea=self)
        return o, ret, bool(d[0]), False, info


class FlexibleFovealEnvActionType(IntEnum):
    FOV_LOC = 0
    FOV_RES = 1


class FlexibleFovealEnv(FixedFovealEnv):
    """``FlexibleFovealEnv`` (reference fov_env.py:240-355)."""
    _KIND = "flexible"
    _INFO_KEYS = ("raw_reward", "reward", "ep_len", "fov_loc", "fov_res")

    def __init__(self, env, args):
        if not (getattr(args, "mask_out", False) or getattr(args, "resize_to_full", False)):
            # single env, raw crops: return the ragged view itself.  The flag goes on a COPY: the caller's args object may
            # build other envs later (an AtariVecEnv of kind "flexible" must keep returning the padded batch)
            args = copy.copy(args)
            args.ragged_obs = "packed"
        super().__init__(env, args)
        self.action_space["sensory_action_type"] = Discrete(len(FlexibleFovealEnvActionType))
        self.fov_init_res = tuple(args.fov_size)
        self.fov_res = np.rint(np.array(self.fov_init_res, copy=True)).astype(np.int32)

    def _sync(self, info):
        self.fov_loc = info["fov_loc"]
        self.fov_res = info["fov_res"]

    def _obs(self, obs, info):
        # raw crops are ragged (fov_env.py:283-298): the core packs them (args.ragged_obs = "packed", set in __init__), so
        # obs[0] already is the [fs, res_h, res_w] view - no padded buffer to slice
        o = obs[0]
        if not (self.mask_out or self.resize_to_full) and not getattr(self._core(), "_ragged_packed", False):
            rh, rw = (int(v) for v in info["fov_res"])          # a core built from another args object: padded batch
            o = o[..., :rh, :rw]
        return o

    def _action(self, action):
        a = super()._action(action)
        t = action["sensory_action_type"]
        if isinstance(t, torch.Tensor):
            t = t.detach().cpu().numpy()
        t = int(np.asarray(t).reshape(-1)[0])
        FlexibleFovealEnvActionType(t)                           # ValueError on unknown types, like the enum call
        a["sensory_action_type"] = np.asarray([t], dtype=np.int32)
        return a


class FixedFovealPeripheralEnv(FixedFovealEnv):
    """``FixedFovealPeripheralEnv`` (reference fov_env.py:358-388)."""
    _KIND = "peripheral"

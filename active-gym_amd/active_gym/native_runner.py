"""NativeHostRunner — ctypes view of libagx_runner.so (include/agx_runner.h): the C++ thread-per-core host
runner.  Same interface and the same per-env control flow as :class:`active_gym.runner.AtariHostRunner`
(reference atari_env.py:84-148); selected with ``args.frame_source = "native"`` (built-in scripted emulator)
or ``"native:ale"`` (real ALE through atari_py's libale_c.so)."""
from __future__ import annotations

import ctypes as C
import os
import random
from typing import Callable, Optional, Sequence

import numpy as np

from . import _native as nat
from .frame_source import RAW_H, RAW_W, resolve_frame_format

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AGXR_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libagx_runner.so")     # AGXR_LIB: a sanitizer build (tools/README.md)


class AgxrConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("num_envs", C.c_int32), ("env_offset", C.c_int32),
                ("action_repeat", C.c_int32), ("clip_reward", C.c_int32), ("num_threads", C.c_int32),
                ("seed", C.c_int64), ("max_episode_frames", C.c_int32), ("scripted_actions", C.c_int32),
                ("scripted_lives", C.c_int32), ("scripted_p_life", C.c_int32), ("scripted_p_over", C.c_int32),
                ("backend", C.c_char_p), ("ale_lib", C.c_char_p), ("rom_path", C.c_char_p),
                ("gray_frames", C.c_int32), ("n_src_rows", C.c_int32), ("src_rows", C.POINTER(C.c_int32)),
                ("cpu_list", C.POINTER(C.c_int32)), ("n_cpus", C.c_int32), ("reserved", C.c_int32)]


_P = C.c_void_p
SIGNATURES = {
    "agxr_create": (C.c_int, [C.POINTER(AgxrConfig), C.POINTER(_P)]),
    "agxr_destroy": (C.c_int, [_P]),
    "agxr_last_error": (C.c_char_p, [_P]),
    "agxr_num_actions": (C.c_int, [_P]),
    "agxr_set_training": (None, [_P, C.c_int]),
    "agxr_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "agxr_step_begin": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32]),
    "agxr_step_wait": (C.c_int, [_P, C.c_int32]),
    "agxr_reset": (C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_int64, _P]),
    "agxr_reset_packed": (C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_int64, _P]),
    "agxr_default_threads": (C.c_int, []),
    "agxr_host_cpus": (None, [C.POINTER(C.c_int32 * 3)]),
    "agxr_num_threads": (C.c_int, [_P]),
    "agxr_worker_cpu": (C.c_int, [_P, C.c_int32]),
    "agxr_get_state": (C.c_int, [_P, _P, _P]),
    "agxr_render": (C.c_int, [_P, C.c_int32, _P]),
}
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: run `python active-gym_amd/build.py`")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def _find_libale_c():
    try:
        import atari_py  # type: ignore
    except ImportError as e:
        raise ImportError("frame_source='native:ale' needs atari_py (its ale_interface/libale_c.so and ROMs)") from e
    base = os.path.dirname(atari_py.__file__)
    for cand in ("ale_interface/libale_c.so", "ale_interface/build/libale_c.so"):
        p = os.path.join(base, cand)
        if os.path.exists(p):
            return p, atari_py
    raise ImportError("libale_c.so not found inside atari_py")


class NativeHostRunner:
    def __init__(self, args, num_envs: int, frames: Optional[np.ndarray] = None, workers: Optional[int] = None,
                 noop_fn: Optional[Callable[[], int]] = None, env_offset: int = 0, backend: str = "scripted",
                 noop_per_env: bool = False, src_rows: Optional[Sequence[int]] = None, cpus: Optional[Sequence[int]] = None,
                 alloc_frames: bool = True):
        """``src_rows``: compact staging - only these screen rows are staged (``ObsPipeline.source_rows()``), every screen
        in ``frames`` has ``len(src_rows)`` rows.  ``cpus``: worker w is pinned to ``cpus[w % len(cpus)]``
        (``active_gym.hostplan``).  ``workers`` None / 0: usable CPUs // LOCAL_WORLD_SIZE (agxr_default_threads)."""
        self._lib = lib()
        self.args = args
        self.num_envs = int(num_envs)
        self.noop_fn = noop_fn or (lambda: random.randrange(30))       # reference atari_env.py:96
        from .runner import per_env_noop_seed
        self._noop_rngs = ([random.Random(per_env_noop_seed(args.seed, int(env_offset) + i)) for i in range(self.num_envs)]
                           if noop_per_env and noop_fn is None else None)      # see AtariHostRunner
        cfg = AgxrConfig()
        cfg.struct_size = C.sizeof(AgxrConfig)
        cfg.num_envs, cfg.env_offset = self.num_envs, int(env_offset)
        cfg.action_repeat, cfg.clip_reward = int(args.action_repeat), int(bool(args.clip_reward))
        cfg.num_threads = int(workers or 0)
        cfg.seed = int(args.seed)
        cfg.max_episode_frames = int(args.max_episode_length)
        cfg.scripted_actions = int(getattr(args, "scripted_actions", 4))
        cfg.scripted_lives = int(getattr(args, "scripted_lives", 3))
        cfg.scripted_p_life = int(getattr(args, "scripted_p_life", 4))
        cfg.scripted_p_over = int(getattr(args, "scripted_p_over", 1))
        self.gray = resolve_frame_format(args, real=(backend == "ale_c")) == "gray"
        cfg.gray_frames = int(self.gray)
        self.src_rows = None if src_rows is None else np.ascontiguousarray(src_rows, dtype=np.int32)
        if self.src_rows is not None:
            cfg.n_src_rows = len(self.src_rows)
            cfg.src_rows = self.src_rows.ctypes.data_as(C.POINTER(C.c_int32))
        self._cpus = None if not cpus else np.ascontiguousarray(list(cpus), dtype=np.int32)
        if self._cpus is not None:
            cfg.n_cpus = len(self._cpus)
            cfg.cpu_list = self._cpus.ctypes.data_as(C.POINTER(C.c_int32))
        cfg.backend = backend.encode()
        if backend == "ale_c":
            so, atari_py = _find_libale_c()
            cfg.ale_lib = so.encode()
            cfg.rom_path = atari_py.get_game_path(args.game).encode()
        self._cfg = cfg
        self._h = _P()
        rc = self._lib.agxr_create(C.byref(cfg), C.byref(self._h))
        if rc:
            raise RuntimeError("agxr_create: " + (self._lib.agxr_last_error(None) or b"").decode())
        self.num_actions = self._lib.agxr_num_actions(self._h)
        self.actions = [list(range(self.num_actions))] * self.num_envs
        self.num_workers = self._lib.agxr_num_threads(self._h)
        self.worker_cpus = [self._lib.agxr_worker_cpu(self._h, w) for w in range(self.num_workers)]
        self.rows = RAW_H if self.src_rows is None else len(self.src_rows)
        shape = (self.num_envs, 2, self.rows, RAW_W) + (() if self.gray else (3,))
        if frames is None and alloc_frames:
            frames = np.zeros(shape, np.uint8)
        if frames is not None:              # (alloc_frames=False: the native step loop owns the staging, step() / reset() here are unused)
            assert frames.shape == shape and frames.dtype == np.uint8 and frames.flags.c_contiguous
        self.frames = frames
        self.frames_shape = shape
        self.training = True
        self._motor = np.zeros(self.num_envs, np.int32)
        self._cmd = np.zeros(self.num_envs, np.uint8)
        self._rew = np.zeros(self.num_envs, np.float64)
        self._raw = np.zeros(self.num_envs, np.float64)
        self._done = np.zeros(self.num_envs, np.uint8)

    def _check(self, rc):
        if rc:
            raise RuntimeError("libagx_runner: " + (self._lib.agxr_last_error(self._h) or b"").decode())

    def close(self):
        for dep in list(getattr(self, "_dependents", ())):      # a native step loop holds this runner's raw handle as its host source
            dep.close()
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.agxr_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def set_frames(self, frames: np.ndarray):
        """Point the runner at another staging buffer of the same shape (double-buffered pinned staging)."""
        assert tuple(frames.shape) == tuple(self.frames_shape) and frames.dtype == np.uint8 and frames.flags.c_contiguous
        self.frames = frames

    def train(self):
        self.training = True
        self._lib.agxr_set_training(self._h, 1)

    def eval(self):
        self.training = False
        self._lib.agxr_set_training(self._h, 0)

    @property
    def lives(self):
        out = np.zeros(self.num_envs, np.int32)
        self._lib.agxr_get_state(self._h, out.ctypes.data, None)
        return out.astype(np.int64)

    @property
    def life_termination(self):
        out = np.zeros(self.num_envs, np.uint8)
        self._lib.agxr_get_state(self._h, None, out.ctypes.data)
        return out.astype(bool)

    def _need_frames(self):
        if self.frames is None:
            raise RuntimeError("this runner was created without a staging buffer (alloc_frames=False: the native step loop owns the "
                               "staging and drives the emulators); pass out= / set_frames() to use it directly")

    def step(self, motor_actions):
        self._need_frames()
        self._motor[:] = np.asarray(motor_actions).reshape(self.num_envs)
        self._check(self._lib.agxr_step(self._h, self._motor.ctypes.data, self.frames.ctypes.data, self._cmd.ctypes.data,
                                        self._rew.ctypes.data, self._raw.ctypes.data, self._done.ctypes.data))
        return self._rew.copy(), self._done.astype(bool), self._cmd.copy(), self._raw.copy()

    def step_begin(self, motor_actions, chunk_envs: int) -> int:
        """Start a step; returns the number of chunks.  ``step_wait(c)`` blocks until chunk c's screens are in
        ``frames`` (so their H2D copy can start while later chunks still emulate); ``step_finish()`` returns what
        :meth:`step` returns."""
        self._need_frames()
        self._motor[:] = np.asarray(motor_actions).reshape(self.num_envs)
        chunk_envs = max(1, min(int(chunk_envs), self.num_envs))
        self._check(self._lib.agxr_step_begin(self._h, self._motor.ctypes.data, self.frames.ctypes.data, self._cmd.ctypes.data,
                                              self._rew.ctypes.data, self._raw.ctypes.data, self._done.ctypes.data, chunk_envs))
        return -(-self.num_envs // chunk_envs)

    def step_wait(self, chunk: int):
        self._check(self._lib.agxr_step_wait(self._h, int(chunk)))

    def step_finish(self):
        self._check(self._lib.agxr_step_wait(self._h, -1))
        return self._rew.copy(), self._done.astype(bool), self._cmd.copy(), self._raw.copy()

    def reset(self, idx: Optional[Sequence[int]] = None, out: Optional[np.ndarray] = None, packed: bool = False) -> np.ndarray:
        """Reset the envs in `idx` (all by default); env i's reset screen goes to ``out[i, 0]``, or - ``packed`` - the j-th
        reset env's to ``out[j, 0]`` (the vector env uploads the reset screens of one step as ONE contiguous copy)."""
        idx = np.arange(self.num_envs, dtype=np.int32) if idx is None else np.asarray(list(idx), dtype=np.int32)
        noops = self.draw_noops(idx)
        if out is None:
            self._need_frames()
        buf = self.frames if out is None else out
        assert buf.dtype == np.uint8 and buf.flags.c_contiguous and buf.shape[0] == self.num_envs
        stride = buf.strides[0]
        fn = self._lib.agxr_reset_packed if packed else self._lib.agxr_reset
        self._check(fn(self._h, idx.ctypes.data, len(idx), noops.ctypes.data, buf.ctypes.data, stride, self._cmd.ctypes.data))
        return self._cmd.copy()

    def draw_noops(self, idx) -> np.ndarray:
        """The ``random.randrange(30)`` draws of the envs in `idx` (reference atari_env.py:96), in that order, only for full
        resets (a life-loss reset plays one no-op and draws nothing) - like the Python runner."""
        lt = self.life_termination
        return np.array([0 if lt[i] else (self._noop_rngs[i].randrange(30) if self._noop_rngs is not None else int(self.noop_fn()))
                         for i in idx], dtype=np.int32)

    def last_error(self) -> str:
        return (self._lib.agxr_last_error(self._h) or b"").decode()

    def render(self, i=0, size=(256, 256)):
        out = np.empty((RAW_H, RAW_W, 3), np.uint8)
        self._check(self._lib.agxr_render(self._h, int(i), out.ctypes.data))
        return out

"""NativeStepLoop — ctypes view of the native step loop of libagx.so (include/agx_loop.h): one C call per vector step.

It owns what ``AtariVecEnv.step`` otherwise does in Python around the kernels (reference: gymnasium's SyncVectorEnv loop,
atari_env.py:241, and the emulator-facing control flow of ``AtariEnv._step/_reset``, atari_env.py:84-148): emulators (the
entry points of libagx_runner.so, handed over as C callbacks), pinned staging, the copy stream, ingest + fovea launches, and
the autoreset of the envs that ended an episode - terminal observations gathered, emulators reset, packed reset screens
uploaded, CLEAR ingest, masked re-observation.  Selected by ``AtariVecEnv`` for ``frame_source="native*"`` with device outputs
(``args.native_loop = False`` keeps the Python loop)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _native as nat

_P = C.c_void_p
STEP_FN = C.CFUNCTYPE(C.c_int, _P, _P, _P, _P, _P, _P, _P)
RESET_FN = C.CFUNCTYPE(C.c_int, _P, _P, C.c_int32, _P, _P, C.c_int64, _P)
NOOPS_FN = C.CFUNCTYPE(C.c_int, _P, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32))


class AgxHostSource(C.Structure):
    _fields_ = [("self", _P), ("step", _P), ("reset_packed", _P), ("draw_noops", NOOPS_FN), ("noops_user", _P)]


class AgxLoopConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("gray", C.c_int32), ("compact", C.c_int32), ("autoreset", C.c_int32)]


class AgxLoopResult(C.Structure):
    _fields_ = [("reward", C.POINTER(C.c_double)), ("raw", C.POINTER(C.c_double)), ("done", C.POINTER(C.c_uint8)),
                ("n_done", C.c_int32), ("done_idx", C.POINTER(C.c_int32)), ("d_final_obs", _P), ("d_final_loc", _P),
                ("d_final_res", _P), ("h2d_bytes", C.c_int64)]


SIGNATURES = {
    "agx_loop_create": (C.c_int, [_P, C.POINTER(AgxHostSource), C.POINTER(AgxLoopConfig), C.POINTER(_P)]),
    "agx_loop_destroy": (C.c_int, [_P]),
    "agx_loop_last_error": (C.c_char_p, [_P]),
    "agx_loop_reset": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "agx_loop_step": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P, _P, C.POINTER(AgxLoopResult), _P]),
    "agx_loop_reset_envs": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P, _P]),
}
_bound = False


def _lib():
    global _bound
    lib = nat.lib()
    if not _bound:
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)           # AttributeError if the .so does not export it
            fn.restype, fn.argtypes = res, args
        _bound = True
    return lib


class _DeviceView:
    """A device buffer owned by the loop as a torch tensor (no copy): __cuda_array_interface__ over the raw pointer."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _view(ptr, shape, typestr, device):
    return torch.as_tensor(_DeviceView(ptr, shape, typestr), device=device)


class NativeStepLoop:
    def __init__(self, pipe, runner, gray: bool, compact: bool, autoreset: bool):
        """pipe: ObsPipeline; runner: NativeHostRunner (its libagx_runner.so handle and entry points become the host source)."""
        self._lib = _lib()
        self.pipe, self.runner = pipe, runner
        self.device = pipe.device
        rl = runner._lib
        self._noops_cb = NOOPS_FN(self._draw_noops)              # kept alive for the life of the loop
        src = AgxHostSource()
        src.self = runner._h
        src.step = C.cast(rl.agxr_step, _P)
        src.reset_packed = C.cast(rl.agxr_reset_packed, _P)
        src.draw_noops = self._noops_cb
        cfg = AgxLoopConfig(C.sizeof(AgxLoopConfig), int(gray), int(compact), int(autoreset))
        self._h = _P()
        rc = self._lib.agx_loop_create(pipe._ctx, C.byref(src), C.byref(cfg), C.byref(self._h))
        if rc:
            raise nat.AgxError(rc, (self._lib.agx_loop_last_error(None) or b"").decode())
        # the loop holds the raw handles of both: whichever of the three is closed (or finalized) first closes the loop first
        for owner in (pipe, runner):
            if not hasattr(owner, "_dependents"):
                owner._dependents = []
            owner._dependents.append(self)
        self.n = pipe.num_envs
        self._motor = np.zeros(self.n, np.int32)
        self._motor_ptr = self._motor.ctypes.data
        self._views = {}
        self._res = AgxLoopResult()
        self._cb_error = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._views = {}                                       # views of buffers that are about to be freed
            self._lib.agx_loop_destroy(self._h)
            self._h = _P()
            for owner in (self.pipe, self.runner):
                if self in getattr(owner, "_dependents", ()):
                    owner._dependents.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _check(self, rc):
        if self._cb_error is not None:
            e, self._cb_error = self._cb_error, None
            raise e
        if rc:
            msg = (self._lib.agx_loop_last_error(self._h) or b"").decode()
            if msg.startswith("host source"):          # the callback's own reason (e.g. a motor action outside the action set)
                msg += ": " + self.runner.last_error()
            raise nat.AgxError(rc, msg)

    def _draw_noops(self, _user, idx, k, out):
        """C callback: the no-op counts of the k envs about to be reset, drawn where the reference draws them (Python's
        ``random``, in env order, only for full resets: NativeHostRunner.draw_noops)."""
        try:
            ids = np.ctypeslib.as_array(idx, shape=(k,))
            vals = self.runner.draw_noops(ids)
            for j in range(k):
                out[j] = int(vals[j])
            return 0
        except Exception as e:  # noqa: BLE001 - an exception must not cross the C frame
            self._cb_error = e
            return 1

    def _stream(self):
        return _P(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _ptr(t: Optional[torch.Tensor]):
        return None if t is None else _P(t.data_ptr())

    def reset(self, obs: torch.Tensor, loc: Optional[torch.Tensor], res: Optional[torch.Tensor]):
        noops = np.ascontiguousarray(self.runner.draw_noops(np.arange(self.n, dtype=np.int32)), dtype=np.int32)
        self._check(self._lib.agx_loop_reset(self._h, noops.ctypes.data, self._ptr(obs), self._ptr(loc), self._ptr(res), self._stream()))

    def reset_envs(self, idx, obs, loc, res):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        noops = np.ascontiguousarray(self.runner.draw_noops(idx), dtype=np.int32)
        self._check(self._lib.agx_loop_reset_envs(self._h, idx.ctypes.data, len(idx), noops.ctypes.data, self._ptr(obs), self._ptr(loc),
                                                  self._ptr(res), self._stream()))

    def step(self, motor, action: Optional[torch.Tensor], action_dt: int, action_type: Optional[torch.Tensor], obs: torch.Tensor,
             loc: Optional[torch.Tensor], res: Optional[torch.Tensor]):
        """Returns (reward f64[N], raw f64[N], done bool[N], done_idx i32[k], final_obs [k, ...] | None, final_loc | None,
        final_res | None) - host arrays are copies, device tensors are views of loop-owned buffers valid until the next step."""
        self._motor[:] = motor
        r = self._res
        self._check(self._lib.agx_loop_step(self._h, self._motor_ptr, self._ptr(action), int(action_dt), self._ptr(action_type),
                                            self._ptr(obs), self._ptr(loc), self._ptr(res), C.byref(r), self._stream()))
        n, k = self.n, int(r.n_done)
        reward = self._host("reward", r.reward, np.float64).copy()
        raw = self._host("raw", r.raw, np.float64).copy()
        done = self._host("done", r.done, np.uint8).astype(bool)
        idx = self._host("done_idx", r.done_idx, np.int32)[:k].copy() if k else np.zeros(0, np.int32)
        fo = fl = fr = None
        if k and r.d_final_obs:
            fo = self._device("final_obs", r.d_final_obs, (n,) + tuple(obs.shape[1:]), "<f4")[:k]
            if r.d_final_loc:
                fl = self._device("final_loc", r.d_final_loc, (n, 2), "<i4")[:k]
            if r.d_final_res:
                fr = self._device("final_res", r.d_final_res, (n, 2), "<i4")[:k]
        self.h2d_bytes = int(r.h2d_bytes)
        return reward, raw, done, idx, fo, fl, fr

    # The loop's result arrays and side buffers are allocated once in agx_loop_create and never move: ONE NumPy / torch view per
    # buffer, made at first use and sliced per step.  (A view per step would leak: a tensor made from __cuda_array_interface__
    # keeps its source object alive for good - 180 B per view, found by tools/soak.py - and np.ctypeslib.as_array builds a ctypes
    # type per distinct length.)
    def _host(self, name, ptr, dtype):
        addr = C.cast(ptr, C.c_void_p).value
        hit = self._views.get(name)
        if hit is None or hit[0] != addr:
            buf = (C.c_char * (self.n * np.dtype(dtype).itemsize)).from_address(addr)
            hit = self._views[name] = (addr, np.frombuffer(buf, dtype=dtype, count=self.n))
        return hit[1]

    def _device(self, name, ptr, shape, typestr):
        addr = int(ptr)
        hit = self._views.get(name)
        if hit is None or hit[0] != addr or tuple(hit[1].shape) != tuple(shape):
            hit = self._views[name] = (addr, _view(addr, shape, typestr, self.device))
        return hit[1]

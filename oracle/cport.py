"""ctypes view of oracle/cport.c (TEST INFRASTRUCTURE ONLY: checker and the timed CPU baseline)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def available():
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_SO)
        _lib.agxo_get_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.agxo_step_fixed.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 3
        _lib.agxo_ingest.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        _lib.agxo_fovea_fixed.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 3
    return _lib


def get_state(rgb, obs=(84, 84)):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    out = np.empty(obs, np.uint8)
    lib().agxo_get_state(rgb.ctypes.data, obs[0], obs[1], out.ctypes.data)
    return out


class EnvBatch:
    """n independent envs of the headline config stepped one after the other, like SyncVectorEnv does."""

    def __init__(self, n, frame_stack=4, obs=(84, 84), fov=(30, 30)):
        self.n, self.fs, self.obs, self.fov = n, frame_stack, obs, fov
        self.ring = np.zeros((n, frame_stack) + obs, np.uint8)
        self.out = np.empty((n, frame_stack) + obs, np.float64)
        self.loc = np.zeros((n, 2), np.int32)

    def step_fixed(self, frames, actions, nvalid=None):
        L = lib()
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        actions = np.ascontiguousarray(actions, dtype=np.float64)
        for i in range(self.n):
            L.agxo_step_fixed(frames[i].ctypes.data, 2 if nvalid is None else int(nvalid[i]), self.ring[i].ctypes.data,
                              self.fs, self.obs[0], self.obs[1], self.fov[0], self.fov[1],
                              actions[i].ctypes.data, self.out[i].ctypes.data, self.loc[i].ctypes.data)
        return self.out, self.loc

    def ingest(self, frames, cmd):
        """The ring part of a step with agx_ingest's command bytes: nvalid | CLEAR 0x04 (zero the stack first) | SKIP 0x08."""
        L = lib()
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        for i in range(self.n):
            c = int(cmd[i])
            if c & 0x08:
                continue
            L.agxo_ingest(frames[i].ctypes.data, min(c & 3, 2), 1 if c & 0x04 else 0, self.ring[i].ctypes.data, self.fs,
                          self.obs[0], self.obs[1])
        return self.ring

    def fovea_fixed(self, actions):
        L = lib()
        actions = np.ascontiguousarray(actions, dtype=np.float64)
        for i in range(self.n):
            L.agxo_fovea_fixed(self.ring[i].ctypes.data, self.fs, self.obs[0], self.obs[1], self.fov[0], self.fov[1],
                               actions[i].ctypes.data, self.out[i].ctypes.data, self.loc[i].ctypes.data)
        return self.out, self.loc

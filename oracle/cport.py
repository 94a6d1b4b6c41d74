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

"""ctypes binding of libagx.so (include/agx.h).  No fallback: if the HIP
extension is missing this module raises, and so does every env built on it."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libagx.so")

ABI_VERSION = 2

OK, E_INVALID, E_HIP, E_NOMEM, E_STATE = 0, -1, -2, -3, -4
KIND_BASE, KIND_FIXED, KIND_FLEXIBLE, KIND_PERIPHERAL = 0, 1, 2, 3
OUT_RAW, OUT_RESIZE, OUT_MASK = 0, 1, 2
MODE_ABSOLUTE, MODE_RELATIVE = 0, 1
DT_F32, DT_F64, DT_I32, DT_I64 = 0, 1, 2, 3
FOV_LOC, FOV_RES = 0, 1
CMD_CLEAR, CMD_SKIP = 0x04, 0x08
K_INGEST, K_FOVEA, K_FULL, K_INGEST_RGB, K_INGEST_GRAY_RAW = 1, 2, 3, 4, 5
GRAY_CV15, GRAY_CV14 = 0, 1
SCREENS_GRAY, SCREENS_COMPACT = 1, 2          # agx_step_flexible_packed: screen layout bits

RAW_H, RAW_W = 210, 160


class AgxConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32), ("num_envs", C.c_int32), ("kind", C.c_int32),
        ("raw_h", C.c_int32), ("raw_w", C.c_int32), ("obs_h", C.c_int32), ("obs_w", C.c_int32),
        ("frame_stack", C.c_int32), ("fov_h", C.c_int32), ("fov_w", C.c_int32),
        ("per_h", C.c_int32), ("per_w", C.c_int32), ("out_mode", C.c_int32), ("action_mode", C.c_int32),
        ("antialias", C.c_int32), ("sas_lo", C.c_double), ("sas_hi", C.c_double), ("init_loc", C.c_double * 2),
    ]


# name -> (restype, argtypes); mirrors include/agx.h one to one
_P = C.c_void_p
SIGNATURES = {
    "agx_abi_version": (C.c_int, []),
    "agx_build_info": (C.c_char_p, []),
    "agx_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "agx_create": (C.c_int, [C.POINTER(AgxConfig), C.POINTER(_P)]),
    "agx_destroy": (C.c_int, [_P]),
    "agx_last_error": (C.c_char_p, [_P]),
    "agx_obs_shape": (C.c_int, [_P, C.POINTER(C.c_int32 * 4)]),
    "agx_algorithmic_bytes": (C.c_int64, [_P, C.c_int]),
    "agx_profile_next": (C.c_int, [_P, C.c_int, _P, _P]),
    "agx_ingest": (C.c_int, [_P, _P, _P, _P]),
    "agx_ingest_gray": (C.c_int, [_P, _P, _P, _P]),
    "agx_ingest_gray_raw": (C.c_int, [_P, _P, _P, _P]),
    "agx_ingest_rgb": (C.c_int, [_P, _P, _P, C.c_int, _P]),
    "agx_source_rows": (C.c_int, [_P, _P, C.POINTER(C.c_int32)]),
    "agx_ingest_compact": (C.c_int, [_P, _P, _P, _P]),
    "agx_ingest_gray_raw_compact": (C.c_int, [_P, _P, _P, _P]),
    "agx_observe_full": (C.c_int, [_P, _P, _P]),
    "agx_get_stack_u8": (C.c_int, [_P, _P, _P]),
    "agx_set_stack_u8": (C.c_int, [_P, _P, _P]),
    "agx_fovea_reset": (C.c_int, [_P, _P, _P]),
    "agx_get_fov_state": (C.c_int, [_P, _P, _P, _P]),
    "agx_set_fov_state": (C.c_int, [_P, _P, _P, _P]),
    "agx_fovea_fixed": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P]),
    "agx_step_fixed": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, _P, _P]),
    "agx_fovea_peripheral": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P]),
    "agx_fovea_flexible": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, _P, _P]),
    "agx_fovea_flexible_packed": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int64, _P, _P, _P, _P]),
    "agx_step_flexible_packed": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, _P, _P, C.c_int64, _P, _P, _P, _P]),
}

_lib = None


class AgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libagx error {code}: {msg}")
        self.code = code


def lib():
    """Load libagx.so once.  Raises ImportError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("AGX_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP extension is not built (run `python active-gym_amd/build.py`). "
            "active_gym has no CPU fallback for the observation path.")
    handle = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)       # AttributeError if the .so does not export it
        except AttributeError:
            if os.environ.get("AGX_LIB") and name == "agx_step_flexible_packed":
                continue                     # an older diagnostic build selected by hand (tools/canary_probe.py's r3bug library)
            raise
        fn.restype = res
        fn.argtypes = args
    v = handle.agx_abi_version()
    if v != ABI_VERSION:
        raise ImportError(f"{path}: ABI version {v}, binding expects {ABI_VERSION}")
    _lib = handle
    return _lib


def build_info() -> str:
    """"libagx abi <v> src <hash>": which sources the loaded library was built from."""
    return lib().agx_build_info().decode()


def last_error(ctx=None):
    s = lib().agx_last_error(ctx)
    return s.decode("utf-8", "replace") if s else ""


def check(rc, ctx=None):
    if rc != OK:
        raise AgxError(rc, last_error(ctx))

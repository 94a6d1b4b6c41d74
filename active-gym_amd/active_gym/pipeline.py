"""ObsPipeline — the batched, device-resident observation pipeline of N
environments on one MI355X.  PyTorch tensors are only the device-buffer / FFI
surface: every operation is one launch of a hand-written HIP kernel in
libagx.so through the C ABI (include/agx.h).

It replaces, for N envs at once, what the reference does per env on the CPU:
``AtariEnv._get_state/_step/_reset`` image work (reference atari_env.py:73-148)
and ``*FovealEnv._fov_step/_get_fov_state`` (reference fov_env.py:166-203,270-330,
375-388).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native as nat

_KINDS = {"base": nat.KIND_BASE, "fixed": nat.KIND_FIXED, "flexible": nat.KIND_FLEXIBLE,
          "peripheral": nat.KIND_PERIPHERAL}
_DT = {torch.float32: nat.DT_F32, torch.float64: nat.DT_F64, torch.int32: nat.DT_I32, torch.int64: nat.DT_I64}


def resolve_out_mode(mask_out: bool, resize_to_full: bool) -> int:
    """Priority of reference fov_env.py:176-185: mask_out > resize_to_full > raw."""
    if mask_out:
        return nat.OUT_MASK
    if resize_to_full:
        return nat.OUT_RESIZE
    return nat.OUT_RAW


class ObsPipeline:
    def __init__(self, num_envs: int, kind: str = "fixed", obs_size: Tuple[int, int] = (84, 84),
                 frame_stack: int = 4, fov_size: Optional[Tuple[int, int]] = None,
                 fov_init_loc: Sequence[float] = (0, 0), sensory_action_mode: str = "absolute",
                 sensory_action_space: Optional[Sequence[float]] = None, resize_to_full: bool = False,
                 mask_out: bool = False, peripheral_res: Optional[Tuple[int, int]] = None,
                 antialias: bool = True, device: Optional[torch.device] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("ObsPipeline needs a ROCm GPU: the observation path has no CPU implementation")
        self._lib = nat.lib()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise ValueError(f"device must be a cuda (ROCm) device, got {self.device}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if kind not in _KINDS:
            raise ValueError(f"kind must be one of {sorted(_KINDS)}")
        if sensory_action_mode not in ("absolute", "relative"):
            raise ValueError("sensory_action_mode must be 'absolute' or 'relative'")
        self.kind = kind
        self.num_envs = int(num_envs)
        self.obs_size = (int(obs_size[0]), int(obs_size[1]))
        self.frame_stack = int(frame_stack)
        cfg = nat.AgxConfig()
        cfg.struct_size = C.sizeof(nat.AgxConfig)
        cfg.device = self.device.index
        cfg.num_envs = self.num_envs
        cfg.kind = _KINDS[kind]
        cfg.raw_h, cfg.raw_w = nat.RAW_H, nat.RAW_W
        cfg.obs_h, cfg.obs_w = self.obs_size
        cfg.frame_stack = self.frame_stack
        if kind != "base":
            if fov_size is None:
                raise ValueError("fov_size is required")
            fov = np.asarray(fov_size)
            # reference fov_env.py:112
            if not (fov < np.asarray(self.obs_size)).all():
                raise ValueError(f"fov_size {tuple(fov_size)} must be smaller than obs_size {self.obs_size}")
            cfg.fov_h, cfg.fov_w = int(fov_size[0]), int(fov_size[1])
            cfg.init_loc[0], cfg.init_loc[1] = float(fov_init_loc[0]), float(fov_init_loc[1])
            cfg.action_mode = nat.MODE_RELATIVE if sensory_action_mode == "relative" else nat.MODE_ABSOLUTE
            if sensory_action_mode == "relative":
                if sensory_action_space is None:
                    raise ValueError("relative mode needs sensory_action_space=(lo, hi)")
                cfg.sas_lo, cfg.sas_hi = float(sensory_action_space[0]), float(sensory_action_space[1])
            if kind == "peripheral":
                if peripheral_res is None:
                    raise ValueError("peripheral_res is required")
                cfg.per_h, cfg.per_w = int(peripheral_res[0]), int(peripheral_res[1])
                cfg.out_mode = nat.OUT_RESIZE          # fov_env.py:361-364: mask_out forced off, full-size obs
            else:
                cfg.out_mode = resolve_out_mode(mask_out, resize_to_full)
            cfg.antialias = 1 if antialias else 0
        self.fov_size = (cfg.fov_h, cfg.fov_w)
        self.out_mode = cfg.out_mode
        self._cfg = cfg
        self._ctx = C.c_void_p()
        nat.check(self._lib.agx_create(C.byref(cfg), C.byref(self._ctx)))
        dims = (C.c_int32 * 4)()
        nat.check(self._lib.agx_obs_shape(self._ctx, C.byref(dims)), self._ctx)
        self.obs_shape = tuple(int(d) for d in dims)
        self.full_shape = (self.num_envs, self.frame_stack) + self.obs_size

    # ------------------------------------------------------------------ plumbing
    def close(self):
        # a native step loop built on this context holds the raw handle: it goes first (also when the garbage collector runs the
        # finalizers of an abandoned env in an order of its own)
        for dep in list(getattr(self, "_dependents", ())):
            dep.close()
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.agx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chk(self, t: torch.Tensor, shape, dtype, name):
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{name} must be a torch.Tensor")
        if t.device != self.device:
            raise ValueError(f"{name} is on {t.device}, pipeline is on {self.device}")
        if dtype is not None and t.dtype != dtype:
            raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
        if not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous")
        return C.c_void_p(t.data_ptr())

    def algorithmic_bytes(self, kernel: str) -> int:
        k = {"ingest": nat.K_INGEST, "fovea": nat.K_FOVEA, "full": nat.K_FULL, "ingest_rgb": nat.K_INGEST_RGB,
             "ingest_gray_raw": nat.K_INGEST_GRAY_RAW}[kernel]
        v = self._lib.agx_algorithmic_bytes(self._ctx, k)
        if v < 0:
            raise nat.AgxError(int(v), "algorithmic_bytes")
        return int(v)

    def profile_next(self, kernel: str, start: torch.cuda.Event, stop: torch.cuda.Event):
        """Arm `start` / `stop` (torch.cuda.Event(enable_timing=True), already recorded once so that their handles
        exist) for the next launch of `kernel` ("ingest" | "fovea"): they receive the kernel's own begin / end."""
        k = {"ingest": nat.K_INGEST, "fovea": nat.K_FOVEA}[kernel]
        nat.check(self._lib.agx_profile_next(self._ctx, k, C.c_void_p(start.cuda_event), C.c_void_p(stop.cuda_event)), self._ctx)

    # ------------------------------------------------------------------ K1
    def ingest(self, frames: torch.Tensor, cmd: torch.Tensor):
        """frames u8[N,2,210,160,3] RGB, cmd u8[N] (nvalid | CMD_CLEAR | CMD_SKIP)."""
        pf = self._chk(frames, (self.num_envs, 2, nat.RAW_H, nat.RAW_W, 3), torch.uint8, "frames")
        pc = self._chk(cmd, (self.num_envs,), torch.uint8, "cmd")
        nat.check(self._lib.agx_ingest(self._ctx, pf, pc, self._stream()), self._ctx)

    def ingest_gray_raw(self, gray: torch.Tensor, cmd: torch.Tensor):
        """gray u8[N,2,210,160]: ALE's own grayscale screens (what the reference reads, atari_env.py:74)."""
        pg = self._chk(gray, (self.num_envs, 2, nat.RAW_H, nat.RAW_W), torch.uint8, "gray")
        pc = self._chk(cmd, (self.num_envs,), torch.uint8, "cmd")
        nat.check(self._lib.agx_ingest_gray_raw(self._ctx, pg, pc, self._stream()), self._ctx)

    def source_rows(self) -> np.ndarray:
        """The screen rows K1 reads (agx_source_rows): the y0 / y1 of cv2.resize's INTER_LINEAR table from 210 to obs rows,
        ascending - 168 of 210 for 84 rows.  A host runner that stages only these rows feeds :meth:`ingest_compact`."""
        rows = (C.c_int32 * nat.RAW_H)()
        n = C.c_int32(0)
        nat.check(self._lib.agx_source_rows(self._ctx, rows, C.byref(n)), self._ctx)
        return np.array(rows[:n.value], dtype=np.int32)

    def ingest_compact(self, rows: torch.Tensor, cmd: torch.Tensor):
        """:meth:`ingest` from compact screens u8[N,2,n_rows,160,3]: row k = screen row ``source_rows()[k]``; same results."""
        n = len(self.source_rows()) if not hasattr(self, "_n_src") else self._n_src
        self._n_src = n
        pf = self._chk(rows, (self.num_envs, 2, n, nat.RAW_W, 3), torch.uint8, "rows")
        pc = self._chk(cmd, (self.num_envs,), torch.uint8, "cmd")
        nat.check(self._lib.agx_ingest_compact(self._ctx, pf, pc, self._stream()), self._ctx)

    def ingest_gray_raw_compact(self, rows: torch.Tensor, cmd: torch.Tensor):
        """:meth:`ingest_gray_raw` from compact grayscale screens u8[N,2,n_rows,160]."""
        n = len(self.source_rows()) if not hasattr(self, "_n_src") else self._n_src
        self._n_src = n
        pg = self._chk(rows, (self.num_envs, 2, n, nat.RAW_W), torch.uint8, "rows")
        pc = self._chk(cmd, (self.num_envs,), torch.uint8, "cmd")
        nat.check(self._lib.agx_ingest_gray_raw_compact(self._ctx, pg, pc, self._stream()), self._ctx)

    def ingest_gray(self, small: torch.Tensor, cmd: torch.Tensor):
        """small u8[N,2,obs_h,obs_w] already obs-sized gray frames."""
        ps = self._chk(small, (self.num_envs, 2) + self.obs_size, torch.uint8, "small")
        pc = self._chk(cmd, (self.num_envs,), torch.uint8, "cmd")
        nat.check(self._lib.agx_ingest_gray(self._ctx, ps, pc, self._stream()), self._ctx)

    # ------------------------------------------------------------------ fused step (fixed kind)
    def step_fixed(self, frames: torch.Tensor, cmd: torch.Tensor, action: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None, loc_out: Optional[torch.Tensor] = None,
                   mid_event: Optional[torch.cuda.Event] = None):
        """ingest(frames, cmd) + fovea(action) in one ABI call (same results); returns (obs, fov_loc).
        `mid_event` (a torch.cuda.Event that has been recorded once, so that its handle exists) is recorded
        between the two launches."""
        if self.kind != "fixed":
            raise RuntimeError("step_fixed needs a pipeline of kind 'fixed'")
        N = self.num_envs
        pf = self._chk(frames, (N, 2, nat.RAW_H, nat.RAW_W, 3), torch.uint8, "frames")
        pc = self._chk(cmd, (N,), torch.uint8, "cmd")
        pa, dt = None, 0
        if action is not None:
            if action.dtype not in _DT:
                raise TypeError(f"sensory action dtype {action.dtype} not supported (f32/f64/i32/i64)")
            pa = self._chk(action, (N, 2), None, "action")
            dt = _DT[action.dtype]
        if out is None:
            out = torch.empty(self.obs_shape, dtype=torch.float32, device=self.device)
        po = self._chk(out, self.obs_shape, torch.float32, "out")
        if loc_out is None:
            loc_out = torch.empty((N, 2), dtype=torch.int32, device=self.device)
        pl = self._chk(loc_out, (N, 2), torch.int32, "loc_out")
        pe = C.c_void_p(mid_event.cuda_event) if mid_event is not None else None
        nat.check(self._lib.agx_step_fixed(self._ctx, pf, pc, pa, dt, po, pl, pe, self._stream()), self._ctx)
        return out, loc_out

    # ------------------------------------------------------------------ K0
    def ingest_rgb(self, frames: torch.Tensor, cmd: torch.Tensor, gray_mode: int = nat.GRAY_CV15):
        """frames u8[N,obs_h,obs_w,3] obs-sized RGB renders (DMC pixel path, reference dmc_env.py:175-186):
        cv2 BGR2GRAY fixed point -> one append to the stack."""
        pf = self._chk(frames, (self.num_envs,) + self.obs_size + (3,), torch.uint8, "frames")
        pc = self._chk(cmd, (self.num_envs,), torch.uint8, "cmd")
        nat.check(self._lib.agx_ingest_rgb(self._ctx, pf, pc, int(gray_mode), self._stream()), self._ctx)

    def observe_full(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty(self.full_shape, dtype=torch.float32, device=self.device)
        po = self._chk(out, self.full_shape, torch.float32, "out")
        nat.check(self._lib.agx_observe_full(self._ctx, po, self._stream()), self._ctx)
        return out

    def stack_u8(self) -> torch.Tensor:
        out = torch.empty(self.full_shape, dtype=torch.uint8, device=self.device)
        nat.check(self._lib.agx_get_stack_u8(self._ctx, C.c_void_p(out.data_ptr()), self._stream()), self._ctx)
        return out

    def set_stack_u8(self, stack: torch.Tensor):
        ps = self._chk(stack, self.full_shape, torch.uint8, "stack")
        nat.check(self._lib.agx_set_stack_u8(self._ctx, ps, self._stream()), self._ctx)

    # ------------------------------------------------------------------ fovea
    def fovea_packed(self, action: Optional[torch.Tensor] = None, action_type: Optional[torch.Tensor] = None,
                     packed: Optional[torch.Tensor] = None, offsets: Optional[torch.Tensor] = None,
                     loc_out: Optional[torch.Tensor] = None, res_out: Optional[torch.Tensor] = None):
        """Flexible env, raw-crop mode, ragged crops packed (reference fov_env.py:283-298 returns [fs, rh, rw] per env):
        returns (packed f32 [capacity], offsets i64 [N+1], fov_loc, fov_res); env n's crops are
        ``packed[offsets[n]:offsets[n+1]].view(fs, rh, rw)``."""
        if self.kind != "flexible" or self.out_mode != nat.OUT_RAW:
            raise RuntimeError("fovea_packed needs kind='flexible' in raw-crop mode (no mask_out, no resize_to_full)")
        N = self.num_envs
        pa, dt = None, 0
        if action is not None:
            if action.dtype not in _DT:
                raise TypeError(f"sensory action dtype {action.dtype} not supported (f32/f64/i32/i64)")
            pa = self._chk(action, (N, 2), None, "action")
            dt = _DT[action.dtype]
        pt = self._chk(action_type, (N,), torch.int32, "action_type") if action_type is not None else None
        if packed is None:
            packed = torch.empty((N * self.frame_stack * self.obs_size[0] * self.obs_size[1],), dtype=torch.float32, device=self.device)
        if packed.dtype != torch.float32 or packed.device != self.device or packed.dim() != 1 or not packed.is_contiguous():
            raise ValueError("packed must be a contiguous 1-D float32 tensor on the pipeline's device")
        if offsets is None:
            offsets = torch.empty((N + 1,), dtype=torch.int64, device=self.device)
        pof = self._chk(offsets, (N + 1,), torch.int64, "offsets")
        if loc_out is None:
            loc_out = torch.empty((N, 2), dtype=torch.int32, device=self.device)
        if res_out is None:
            res_out = torch.empty((N, 2), dtype=torch.int32, device=self.device)
        pl = self._chk(loc_out, (N, 2), torch.int32, "loc_out")
        pr = self._chk(res_out, (N, 2), torch.int32, "res_out")
        nat.check(self._lib.agx_fovea_flexible_packed(self._ctx, pa, dt, pt, C.c_void_p(packed.data_ptr()), packed.numel(), pof, pl,
                                                      pr, self._stream()), self._ctx)
        return packed, offsets, loc_out, res_out

    def step_flexible_packed(self, screens: torch.Tensor, cmd: torch.Tensor, action: Optional[torch.Tensor] = None,
                             action_type: Optional[torch.Tensor] = None, packed: Optional[torch.Tensor] = None,
                             offsets: Optional[torch.Tensor] = None, loc_out: Optional[torch.Tensor] = None,
                             res_out: Optional[torch.Tensor] = None):
        """One whole step of a flexible raw-crop pipeline with packed ragged observations - ``ingest*(screens, cmd)`` followed by
        :meth:`fovea_packed` - in one ABI call and two launches (agx_step_flexible_packed: the state update + scan ride in the
        ingest launch).  `screens`: u8 [N,2,210,160,3] | [N,2,210,160] (ALE grayscale) | [N,2,R,160,3] | [N,2,R,160] with R =
        len(source_rows()) (compact).  Same results as the two calls; returns what :meth:`fovea_packed` returns."""
        if self.kind != "flexible" or self.out_mode != nat.OUT_RAW:
            raise RuntimeError("step_flexible_packed needs kind='flexible' in raw-crop mode (no mask_out, no resize_to_full)")
        N = self.num_envs
        if not hasattr(self, "_n_src"):
            self._n_src = len(self.source_rows())
        if not isinstance(screens, torch.Tensor) or screens.dim() not in (4, 5):
            raise ValueError("screens must be a u8 tensor [N,2,rows,160,3] (RGB) or [N,2,rows,160] (gray)")
        gray = screens.dim() == 4
        rows = int(screens.shape[2])
        if rows not in (nat.RAW_H, self._n_src):
            raise ValueError(f"screens have {rows} rows; whole screens have {nat.RAW_H}, compact ones {self._n_src}")
        compact = rows != nat.RAW_H
        ps = self._chk(screens, (N, 2, rows, nat.RAW_W) + (() if gray else (3,)), torch.uint8, "screens")
        pc = self._chk(cmd, (N,), torch.uint8, "cmd")
        pa, dt = None, 0
        if action is not None:
            if action.dtype not in _DT:
                raise TypeError(f"sensory action dtype {action.dtype} not supported (f32/f64/i32/i64)")
            pa = self._chk(action, (N, 2), None, "action")
            dt = _DT[action.dtype]
        pt = self._chk(action_type, (N,), torch.int32, "action_type") if action_type is not None else None
        if packed is None:
            packed = torch.empty((N * self.frame_stack * self.obs_size[0] * self.obs_size[1],), dtype=torch.float32, device=self.device)
        if packed.dtype != torch.float32 or packed.device != self.device or packed.dim() != 1 or not packed.is_contiguous():
            raise ValueError("packed must be a contiguous 1-D float32 tensor on the pipeline's device")
        if offsets is None:
            offsets = torch.empty((N + 1,), dtype=torch.int64, device=self.device)
        pof = self._chk(offsets, (N + 1,), torch.int64, "offsets")
        if loc_out is None:
            loc_out = torch.empty((N, 2), dtype=torch.int32, device=self.device)
        if res_out is None:
            res_out = torch.empty((N, 2), dtype=torch.int32, device=self.device)
        pl = self._chk(loc_out, (N, 2), torch.int32, "loc_out")
        pr = self._chk(res_out, (N, 2), torch.int32, "res_out")
        layout = (nat.SCREENS_GRAY if gray else 0) | (nat.SCREENS_COMPACT if compact else 0)
        nat.check(self._lib.agx_step_flexible_packed(self._ctx, ps, layout, pc, pa, dt, pt, C.c_void_p(packed.data_ptr()), packed.numel(),
                                                     pof, pl, pr, self._stream()), self._ctx)
        return packed, offsets, loc_out, res_out

    def fovea_reset(self, mask: Optional[torch.Tensor] = None):
        pm = self._chk(mask, (self.num_envs,), torch.uint8, "mask") if mask is not None else None
        nat.check(self._lib.agx_fovea_reset(self._ctx, pm, self._stream()), self._ctx)

    def fov_state(self):
        loc = torch.empty((self.num_envs, 2), dtype=torch.int32, device=self.device)
        res = torch.empty((self.num_envs, 2), dtype=torch.int32, device=self.device)
        nat.check(self._lib.agx_get_fov_state(self._ctx, C.c_void_p(loc.data_ptr()), C.c_void_p(res.data_ptr()),
                                              self._stream()), self._ctx)
        return loc, res

    def set_fov_state(self, loc: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None):
        pl = self._chk(loc, (self.num_envs, 2), torch.int32, "loc") if loc is not None else None
        pr = self._chk(res, (self.num_envs, 2), torch.int32, "res") if res is not None else None
        nat.check(self._lib.agx_set_fov_state(self._ctx, pl, pr, self._stream()), self._ctx)

    def fovea(self, action: Optional[torch.Tensor] = None, action_type: Optional[torch.Tensor] = None,
              mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
              loc_out: Optional[torch.Tensor] = None, res_out: Optional[torch.Tensor] = None):
        """One ``_fov_step`` for all envs.  action [N,2] (f32/f64/i32/i64) or None
        (observe at the current fov_loc).  Returns (obs, fov_loc[, fov_res])."""
        if self.kind == "base":
            raise RuntimeError("base pipeline has no fovea; use observe_full()")
        N = self.num_envs
        pa, dt = None, 0
        if action is not None:
            if action.dtype not in _DT:
                raise TypeError(f"sensory action dtype {action.dtype} not supported (f32/f64/i32/i64)")
            pa = self._chk(action, (N, 2), None, "action")
            dt = _DT[action.dtype]
        pm = self._chk(mask, (N,), torch.uint8, "mask") if mask is not None else None
        if out is None:
            out = torch.empty(self.obs_shape, dtype=torch.float32, device=self.device)
        po = self._chk(out, self.obs_shape, torch.float32, "out")
        if loc_out is None:
            loc_out = torch.empty((N, 2), dtype=torch.int32, device=self.device)
        pl = self._chk(loc_out, (N, 2), torch.int32, "loc_out")
        st = self._stream()
        if self.kind == "fixed":
            nat.check(self._lib.agx_fovea_fixed(self._ctx, pa, dt, pm, po, pl, st), self._ctx)
            return out, loc_out
        if self.kind == "peripheral":
            nat.check(self._lib.agx_fovea_peripheral(self._ctx, pa, dt, pm, po, pl, st), self._ctx)
            return out, loc_out
        pt = self._chk(action_type, (N,), torch.int32, "action_type") if action_type is not None else None
        if res_out is None:
            res_out = torch.empty((N, 2), dtype=torch.int32, device=self.device)
        pr = self._chk(res_out, (N, 2), torch.int32, "res_out")
        nat.check(self._lib.agx_fovea_flexible(self._ctx, pa, dt, pt, pm, po, pl, pr, st), self._ctx)
        return out, loc_out, res_out

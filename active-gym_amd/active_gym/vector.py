"""Batched active-vision Atari envs on one MI355X.

The reference has no vectorization of its own; callers loop N Python envs
through ``gymnasium.vector.SyncVectorEnv`` (reference atari_env.py:241,276).
:class:`AtariVecEnv` replaces that loop: N emulators on host cores
(:class:`~active_gym.runner.AtariHostRunner`), raw screens staged through a
pinned buffer and copied asynchronously to HBM, and the whole observation
pipeline as HIP kernels (:class:`~active_gym.pipeline.ObsPipeline`).

It keeps the SyncVectorEnv conventions of gymnasium<1.0 (setup.py:15 of the
reference pins ``gymnasium>=0.28.1,<1.0.0``):
  * ``step({"motor_action": (N,), "sensory_action": (N,2)[, "sensory_action_type": (N,)|(N,1)]})``
    -> ``obs (N,fs,h,w) f32, reward (N,), terminated (N,), truncated (N,), infos``
  * infos is a dict of arrays with ``_key`` masks; done envs are reset inside
    the same call and their last observation / info are returned under
    ``final_observation`` / ``final_info``.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _native as nat
from .pipeline import ObsPipeline
from .runner import AtariHostRunner
from .spaces import Box, Dict, Discrete, batch_space

_KINDS = ("base", "fixed", "flexible", "peripheral")


def _resolve_antialias(args):
    """torchvision's Resize default: antialias=True from 0.17 on (the reference leaves the version
    unpinned, setup.py:17).  ``args.antialias`` overrides."""
    return bool(getattr(args, "antialias", True))


class _HostObsPool:
    """Fresh-array semantics for host (NumPy) observations at pinned-buffer cost.

    The reference's envs hand out a new array per call (atari_env.py:143 ``np.stack``), so a caller may keep any of them.  A
    device-to-PAGEABLE copy into a fresh 115 MB array costs 30+ ms per step at N = 1024 (page faults of the fresh mapping +
    the staged copy); a copy into PINNED memory 2 ms.  The pool hands out NumPy views of pinned buffers and takes a buffer back
    only when the array it handed out - and every view derived from it - has been garbage collected (``weakref.finalize`` on
    the array: derived views keep their base alive).  A caller that drops its observations as it goes (the usual loop) cycles
    through 2-3 buffers; one that keeps them all gets ``max_buffers`` pinned ones and ordinary pageable arrays after that.
    """

    def __init__(self, max_buffers: int):
        import collections
        self.max_buffers = int(max_buffers)
        self._free = collections.deque()         # append / pop are atomic: finalizers may run on any thread
        self._made = 0
        self._shape = None

    def take(self, shape, dtype):
        """A pinned tensor no array refers to, or None (budget spent: the caller falls back to a pageable array)."""
        shape = tuple(shape)
        if self._shape != (shape, dtype):       # another observation shape: start over (outstanding arrays keep their buffers)
            self._free.clear()
            self._made = 0
            self._shape = (shape, dtype)
        try:
            return self._free.pop()
        except IndexError:
            pass
        if self._made >= self.max_buffers:
            return None
        self._made += 1
        return torch.empty(shape, dtype=dtype, pin_memory=True)

    def hand_out(self, buf: torch.Tensor) -> np.ndarray:
        import weakref
        arr = buf.numpy()
        key = self._shape
        weakref.finalize(arr, self._give_back, buf, key).atexit = False      # nothing to recycle at interpreter shutdown
        return arr

    def _give_back(self, buf, key):
        if key == self._shape:
            self._free.append(buf)


class AtariVecEnv:
    """N envs of one kind.  ``args`` is an ``AtariEnvArgs``; extra optional attributes:
    ``frame_source`` ("ale" | "synthetic" | factory), ``device`` (None -> NumPy outputs on the host like
    the reference; a cuda device -> torch tensors that stay in HBM), ``antialias``, ``num_workers``."""

    _loop = None             # NativeStepLoop when the native step loop drives this env (subclasses with their own source: never)
    _want_loop = False

    def __init__(self, args, num_envs: int, kind: str = "fixed", env_offset: int = 0, noop_fn=None,
                 autoreset: bool = True, noop_per_env: bool = False):
        self._noop_per_env = bool(noop_per_env)
        self.autoreset = bool(autoreset)        # False: single-env semantics, the caller calls reset()
        if kind not in _KINDS:
            raise ValueError(f"kind must be one of {_KINDS}")
        if not torch.cuda.is_available():
            raise RuntimeError("active_gym envs need a ROCm GPU: the observation pipeline has no CPU implementation")
        self.args = args
        self.kind = kind
        self.num_envs = int(num_envs)
        self.obs_size = tuple(int(v) for v in args.obs_size)
        self._check_obs_size()
        self.frame_stack = int(args.frame_stack)
        self.action_repeat = int(args.action_repeat)
        dev = getattr(args, "device", None)
        self._numpy_out = dev is None
        self.device = torch.device(dev) if dev is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())

        self._build_pipeline()
        self._setup_source(args, noop_fn, env_offset)
        self._build_spaces()

    def rekind(self, kind: str):
        """Swap the wrapper kind before the first reset, keeping the host runner and its emulators (the single-env fovea
        wrappers build on the base env's core: one ALE / MuJoCo construction per env, not two)."""
        if kind not in _KINDS:
            raise ValueError(f"kind must be one of {_KINDS}")
        if self._was_reset:
            raise RuntimeError("rekind() after reset()")
        if kind != self.kind:
            if self._loop is not None:
                self._loop.close()
                self._loop = None
            self.pipe.close()
            self.kind = kind
            self._build_pipeline()
            self._make_loop()
            self._build_spaces()
        return self

    def _build_pipeline(self):
        args, kind = self.args, self.kind
        kw = dict(num_envs=self.num_envs, kind=kind, obs_size=self.obs_size, frame_stack=self.frame_stack,
                  device=self.device)
        if kind != "base":
            # these have no defaults in the reference and are read unconditionally (fov_env.py:110-120)
            self.fov_size = tuple(int(v) for v in args.fov_size)
            self.fov_init_loc = tuple(args.fov_init_loc)
            self.sensory_action_mode = args.sensory_action_mode
            if self.sensory_action_mode == "relative":
                self.sensory_action_space = np.array(args.sensory_action_space)          # fov_env.py:116
            elif self.sensory_action_mode == "absolute":
                self.sensory_action_space = np.array(self.obs_size) - np.array(self.fov_size)   # fov_env.py:118
            else:
                raise ValueError("sensory_action_mode must be 'absolute' or 'relative'")
            resize_to_full = bool(args.resize_to_full)
            mask_out = bool(args.mask_out)
            kw.update(fov_size=self.fov_size, fov_init_loc=self.fov_init_loc,
                      sensory_action_mode=self.sensory_action_mode,
                      sensory_action_space=tuple(np.asarray(args.sensory_action_space, dtype=float))
                      if self.sensory_action_mode == "relative" else None,
                      resize_to_full=resize_to_full, mask_out=mask_out, antialias=_resolve_antialias(args))
            if kind == "peripheral":
                self.peripheral_res = tuple(int(v) for v in args.peripheral_res)
                kw["peripheral_res"] = self.peripheral_res
                mask_out, resize_to_full = False, True                                  # fov_env.py:361-362
            self.mask_out, self.resize_to_full = mask_out, resize_to_full
        self.pipe = ObsPipeline(**kw)

    def _build_spaces(self):
        kind = self.kind
        # spaces (reference atari_env.py:69-70, fov_env.py:125-142,243)
        self.single_motor_space = self._motor_space()
        full = (self.frame_stack,) + self.obs_size
        if kind == "base":
            self.single_action_space = self.single_motor_space
            self.single_observation_space = Box(low=-1., high=1., shape=full, dtype=np.float32)
        else:
            sas = self.sensory_action_space
            spaces = {"motor_action": self.single_motor_space,
                      "sensory_action": Box(low=sas[0], high=sas[1], dtype=int)}      # scalar Box, as the reference
            if kind == "flexible":
                spaces["sensory_action_type"] = Discrete(2)
            self.single_action_space = Dict(spaces)
            crop = kind == "fixed" and not (self.mask_out or self.resize_to_full)
            shp = (self.frame_stack,) + (self.fov_size if crop else self.obs_size)
            self.single_observation_space = Box(low=-1., high=1., shape=shp, dtype=np.float32)
        # SyncVectorEnv conventions (gymnasium<1.0): batched spaces under action_space / observation_space
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)

        # RecordWrapper bookkeeping (fov_env.py:29-32,58-63)
        self.cumulative_reward = np.zeros(self.num_envs, np.float64)
        self.ep_len = np.zeros(self.num_envs, np.int64)
        # Device-tensor outputs are double-buffered: step() returns buffer t % 2, so an observation stays valid until the
        # step after next without a 115 MB clone per step (args.copy_obs=True restores a fresh tensor per call)
        shp = self.pipe.obs_shape if kind != "base" else self.pipe.full_shape
        self._copy_obs = bool(getattr(self.args, "copy_obs", False))
        # Host (NumPy) outputs: a fresh array per call by default, like the reference's envs.  args.copy_obs = False (gymnasium's
        # SyncVectorEnv(copy=False)) returns views of two PINNED host buffers used alternately instead (4 ms at N = 1024): an
        # observation then stays valid until the step after next, as with device outputs.
        self._pinned_host_obs = self._numpy_out and getattr(self.args, "copy_obs", True) is False
        self._h_obs = None
        self._h_obs_i = 0
        # ... and the default itself: fresh-array SEMANTICS from a pool of pinned buffers that are recycled once the caller has
        # dropped the array (_HostObsPool; args.host_obs_buffers = 0 restores the pageable copy per call)
        nbuf = getattr(self.args, "host_obs_buffers", 4)
        self._host_pool = _HostObsPool(int(nbuf)) if (self._numpy_out and not self._pinned_host_obs and int(nbuf or 0) > 0) else None
        self._obs_bufs = [torch.empty(shp, dtype=torch.float32, device=self.device)
                          for _ in range(1 if (self._numpy_out or self._copy_obs) else 2)]
        self._obs_i = 0
        self._obs = self._obs_bufs[0]
        # flexible env, raw-crop mode: args.ragged_obs = "packed" returns the ragged crops themselves - a list of N
        # arrays [fs, res_h, res_w] (views into one packed device buffer), what the reference's env returns per env
        # (fov_env.py:283-298) - instead of the zero-padded [N, fs, obs_h, obs_w] batch
        self._ragged_packed = (kind == "flexible" and not (self.mask_out or self.resize_to_full)
                               and getattr(self.args, "ragged_obs", "padded") == "packed")
        if self._ragged_packed:
            self._packed = torch.empty((self.num_envs * self.frame_stack * self.obs_size[0] * self.obs_size[1],),
                                       dtype=torch.float32, device=self.device)
            self._poff = torch.zeros((self.num_envs + 1,), dtype=torch.int64, device=self.device)
        self._loc = torch.zeros((self.num_envs, 2), dtype=torch.int32, device=self.device)
        self._res = torch.zeros((self.num_envs, 2), dtype=torch.int32, device=self.device)
        self._was_reset = False

    def _check_obs_size(self):
        if self.obs_size[0] != self.obs_size[1]:
            # reference: cv2.resize(dsize=obs_size) yields (obs_size[1], obs_size[0]) and the assignment into
            # frame_buffer raises (atari_env.py:74,121-128)
            raise ValueError(f"obs_size {self.obs_size} must be square for Atari (cv2.resize takes (width, height))")

    def _motor_space(self):
        return Discrete(self.runner.num_actions)

    def _ingest(self, cmd=None):
        cmd = self._d_cmd if cmd is None else cmd
        if self._compact:
            (self.pipe.ingest_gray_raw_compact if self._gray else self.pipe.ingest_compact)(self._d_frames, cmd)
        elif self._gray:
            self.pipe.ingest_gray_raw(self._d_frames, cmd)
        else:
            self.pipe.ingest(self._d_frames, cmd)

    def _extra_info(self, info):
        return info

    def _setup_source(self, args, noop_fn, env_offset):
        # host side: pinned staging for step frames and for reset frames, device twins
        from .frame_source import resolve_frame_format
        fmt = resolve_frame_format(args)         # real emulators default to ALE's own grayscale screens (atari_env.py:74)
        self._gray = fmt == "gray"
        px = () if self._gray else (3,)
        # Compact staging (default; args.compact_rows = False restores whole screens): the runner stages only the screen rows
        # the vertical resize reads (168 of 210 at 84 x 84 - SURVEY.md 8d's algorithmic bytes already leave the other 42 out),
        # so a step's H2D copy is 165 MB instead of 206 MB on the PCIe-bound RGB path
        self._compact = bool(getattr(args, "compact_rows", True))
        self._src_rows = self.pipe.source_rows() if self._compact else None
        rows = len(self._src_rows) if self._compact else nat.RAW_H
        shape = (self.num_envs, 2, rows, nat.RAW_W) + px
        # Host placement (hostplan.py): this rank's share of the CPUs of its GPU's NUMA node; the pinned staging below is
        # allocated while bound to them (first touch on that node), the native runner pins one worker per CPU
        from . import hostplan
        self.host_plan = hostplan.plan_for_process(self.device.index, workers=getattr(args, "num_workers", None))
        src = getattr(args, "frame_source", "ale")
        native = isinstance(src, str) and src.startswith("native")
        # The native step loop (libagx.so: agx_loop_*, include/agx_loop.h) owns staging, copy stream, launches and the autoreset:
        # step() is then ONE C call.  Used with the native runner and device outputs (args.native_loop = False keeps the Python
        # loop below); the chunked-H2D form and the ragged packed observations stay on the Python loop.
        self._loop = None
        self._want_loop = bool(native and not self._numpy_out and getattr(args, "native_loop", True)
                               and not int(getattr(args, "h2d_chunk_envs", 0) or 0)
                               and not (self.kind == "flexible" and not (bool(args.mask_out) or bool(args.resize_to_full))
                                        and getattr(args, "ragged_obs", "padded") == "packed"))
        if not self._want_loop:
            with hostplan.bound_to(self.host_plan["cpus"] and self.host_plan["domain"]):
                self._alloc_staging(shape, rows, px)
        if native:
            # C++ thread-per-core runner (libagx_runner.so): "native" = built-in scripted emulator,
            # "native:ale" = real ALE through atari_py's libale_c.so
            from .native_runner import NativeHostRunner
            self.runner = NativeHostRunner(args, self.num_envs, frames=None if self._want_loop else self._h_frames.numpy(),
                                           workers=self.host_plan["workers"], noop_fn=noop_fn,
                                           env_offset=env_offset, backend="ale_c" if src == "native:ale" else "scripted",
                                           noop_per_env=self._noop_per_env, src_rows=self._src_rows,
                                           cpus=self.host_plan["cpus"], alloc_frames=not self._want_loop)
            self._make_loop()
        else:
            self.runner = AtariHostRunner(args, self.num_envs, frames=self._h_frames.numpy(),
                                          workers=self.host_plan["workers"], noop_fn=noop_fn,
                                          env_offset=env_offset, noop_per_env=self._noop_per_env, src_rows=self._src_rows)

    def _make_loop(self):
        if getattr(self, "_want_loop", False):
            from . import hostplan
            from .native_loop import NativeStepLoop
            # its pinned staging is allocated inside: bound to this rank's CPUs (first touch on the GPU's NUMA node)
            with hostplan.bound_to(self.host_plan["cpus"] and self.host_plan["domain"]):
                self._loop = NativeStepLoop(self.pipe, self.runner, gray=self._gray, compact=self._compact, autoreset=self.autoreset)

    def _alloc_staging(self, shape, rows, px):
        # Two pinned staging sets (screens, command bytes, copy-done event), used alternately: with device outputs step()
        # returns without synchronising, so the emulators of step t+1 fill one set while the H2D copy of step t still
        # drains the other (host-side double buffering; on the device the copies are stream-ordered behind the kernels
        # that read the previous screens, so one device buffer is enough)
        nstage = 1 if self._numpy_out else 2
        self._stage = [{"frames": torch.empty(shape, dtype=torch.uint8, pin_memory=True),
                        "cmd": torch.empty((self.num_envs,), dtype=torch.uint8, pin_memory=True),
                        "ev": torch.cuda.Event()} for _ in range(nstage)]
        self._stage_i = 0
        self._h_frames = self._stage[0]["frames"]
        # Device side.  With device outputs (no synchronisation inside step()) the step screens are double-buffered on the
        # device too and travel on a COPY STREAM of their own: the H2D copy of step t+1 then runs under the kernels (and the
        # autoreset pass) of step t instead of queueing behind them on the one stream - on a PCIe-bound step that is the
        # difference between 87 % and ~95 % of the link.  Two events per buffer order the streams: `copied` (copy stream ->
        # the kernels wait for their screens) and `free` (launch stream -> the copy that overwrites a buffer waits for the
        # kernels that read it two steps earlier).  NumPy outputs synchronise every step anyway: one buffer, one stream.
        ndev = 1 if self._numpy_out else 2
        self._dsets = [{"frames": torch.empty(shape, dtype=torch.uint8, device=self.device),
                        "cmd": torch.empty((self.num_envs,), dtype=torch.uint8, device=self.device),
                        "free": torch.cuda.Event()} for _ in range(ndev)]
        self._dset_i = 0
        self._copy_stream = torch.cuda.Stream(device=self.device) if ndev > 1 else None
        self._d_frames = self._dsets[0]["frames"]
        # reset screens get their own pinned buffer: the autoreset inside step() must not overwrite step
        # screens whose asynchronous H2D copy may still be in flight
        self._h_rframes = torch.empty((self.num_envs, 1, rows, nat.RAW_W) + px, dtype=torch.uint8, pin_memory=True)
        self._h_rcmd = torch.empty((self.num_envs,), dtype=torch.uint8, pin_memory=True)
        self._alloc_reset_buffers()
        self._ev_copy = self._stage[0]["ev"]
        self._h_cmd = self._stage[0]["cmd"]
        self._d_cmd = self._dsets[0]["cmd"]

    def _alloc_reset_buffers(self):
        """Staging for resets of a SUBSET of the envs (the autoreset inside step(), reset_envs()).  The runner writes the K reset
        screens PACKED into the first K rows of a pinned buffer: one contiguous H2D copy into _d_rframes, one index_copy_ into
        slot 0 of the step screens.  Env indices, command bytes and the done mask travel together in one small pinned buffer
        (one copy; a pageable .to(device) would be a synchronous one).  TWO pinned sets, used alternately: the set a reset
        writes was last read by the copies of the reset before the previous one - waiting on the previous reset's event
        instead would wait for that whole step's H2D copy and kernels, i.e. serialise the host with the GPU."""
        n = self.num_envs
        self._rsets = [{"frames": self._h_rframes if k == 0 else torch.empty_like(self._h_rframes).pin_memory(),
                        "meta": torch.empty((10 * n,), dtype=torch.uint8, pin_memory=True),        # idx i64 [N] | cmd [N] | mask [N]
                        "ev": torch.cuda.Event()} for k in range(2)]
        self._rset_i = 0
        self._rfree = torch.cuda.Event()        # launch stream: the kernels of the last partial reset have read the device-side staging
        self._d_rframes = None                  # allocated at the first partial reset
        self._d_rmeta = torch.empty((10 * n,), dtype=torch.uint8, device=self.device)
        self._d_ridx = self._d_rmeta[:8 * n].view(torch.int64)
        self._d_rcmd = self._d_rmeta[8 * n:9 * n]
        self._d_rmask = self._d_rmeta[9 * n:]

    def _h_reset_rows(self, buf=None):
        """Pinned reset screens, one row per env (Atari: slot 0 of a [N, 1, ...] buffer)."""
        return (self._h_rframes if buf is None else buf)[:, 0]

    def _d_reset_target(self):
        """Where a reset screen lands on the device: slot 0 of the env's step screens."""
        return self._d_frames[:, 0]

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, "_loop", None) is not None:
            self._loop.close()
            self._loop = None
        self.runner.close()
        self.pipe.close()

    def train(self):
        self.runner.train()

    def eval(self):
        self.runner.eval()

    @property
    def fov_loc(self) -> np.ndarray:
        return self.pipe.fov_state()[0].cpu().numpy()

    @property
    def fov_res(self) -> np.ndarray:
        return self.pipe.fov_state()[1].cpu().numpy()

    def _upload(self, cmd: np.ndarray):
        """Asynchronous H2D of the step screens and the command bytes (pinned -> HBM): on the copy stream when there is one
        (device outputs), ordered against the launch stream by events; on the current stream otherwise."""
        self._h_cmd.numpy()[:] = cmd
        cur = torch.cuda.current_stream(self.device)
        cs = getattr(self, "_copy_stream", None)
        if cs is None:
            self._d_cmd.copy_(self._h_cmd, non_blocking=True)
            self._d_frames.copy_(self._h_frames, non_blocking=True)
            self._ev_copy.record(cur)
            return
        cs.wait_event(self._dsets[self._dset_i]["free"])        # the kernels that read this device buffer two steps ago
        with torch.cuda.stream(cs):
            self._d_cmd.copy_(self._h_cmd, non_blocking=True)
            self._d_frames.copy_(self._h_frames, non_blocking=True)
            self._ev_copy.record(cs)
        cur.wait_event(self._ev_copy)

    def _next_dset(self):
        ds = getattr(self, "_dsets", None)
        if ds is not None and len(ds) > 1:
            self._dset_i ^= 1
            self._d_frames, self._d_cmd = ds[self._dset_i]["frames"], ds[self._dset_i]["cmd"]

    def _release_dset(self):
        """Called when the last kernel that reads the current device screens has been enqueued."""
        ds = getattr(self, "_dsets", None)
        if ds is not None and len(ds) > 1:
            cur = torch.cuda.current_stream(self.device)
            ds[self._dset_i]["free"].record(cur)
            self._rfree.record(cur)

    def _reset_subset(self, idx):
        """runner.reset of the envs in `idx` + H2D of their screens, command bytes, indices and mask (see _alloc_reset_buffers).
        Returns (done mask, env indices) as device tensors; the command bytes of this pass are in self._d_rcmd."""
        idx = np.asarray(idx, dtype=np.int64)
        k, n = len(idx), self.num_envs
        self._rset_i ^= 1
        st = self._rsets[self._rset_i]
        st["ev"].synchronize()                  # the reset before the previous one has left this pinned set
        cmd = self.runner.reset(idx, out=st["frames"].numpy(), packed=True)
        meta = st["meta"].numpy()
        meta[:8 * n].view(np.int64)[:k] = idx
        meta[8 * n:9 * n] = cmd
        m = meta[9 * n:]
        m[:] = 0
        m[idx] = 1
        rows = self._h_reset_rows(st["frames"])
        if self._d_rframes is None:
            self._d_rframes = torch.empty(rows.shape, dtype=torch.uint8, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        cs = getattr(self, "_copy_stream", None)
        if cs is None:
            self._d_rmeta.copy_(st["meta"], non_blocking=True)
            self._d_rframes[:k].copy_(rows[:k], non_blocking=True)
            st["ev"].record(cur)
        else:
            # on the COPY stream, i.e. queued between this step's screens and the next step's: a small copy issued on the launch
            # stream would reach the DMA engine behind the next step's 200 MB copy and stall this step's reset kernels (and
            # everything ordered after them) for a whole copy time - measured: 5.1 ms per RGB step instead of 4.1
            cs.wait_event(self._rfree)                           # the previous reset's index_copy_ has read _d_rframes / _d_rmeta
            with torch.cuda.stream(cs):
                self._d_rmeta.copy_(st["meta"], non_blocking=True)
                self._d_rframes[:k].copy_(rows[:k], non_blocking=True)
                st["ev"].record(cs)
            cur.wait_event(st["ev"])
        self._d_reset_target().index_copy_(0, self._d_ridx[:k], self._d_rframes[:k])
        return self._d_rmask, self._d_ridx[:k]

    def _upload_reset_all(self, cmd: np.ndarray):
        """H2D of every env's reset screen (slot 0 only: one strided copy) and the command bytes."""
        self._h_rcmd.numpy()[:] = cmd
        self._d_cmd.copy_(self._h_rcmd, non_blocking=True)
        self._d_reset_target().copy_(self._h_reset_rows(), non_blocking=True)
        self._rsets[0]["ev"].record(torch.cuda.current_stream(self.device))

    def _as_device_action(self, a, cols):
        if isinstance(a, torch.Tensor):
            t = a.detach()
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(a)))
        if t.dtype == torch.float16 or t.dtype == torch.bfloat16:
            t = t.float()
        if t.dtype in (torch.int8, torch.int16, torch.uint8, torch.bool):
            t = t.to(torch.int32)
        n = self.num_envs
        if cols and t.numel() in (n, 1) and t.numel() != n * cols:
            # the reference's sensory_action space is a SCALAR Box (fov_env.py:125-129): `action_space.sample()` gives one
            # number per env, which `np.clip(loc, 0, obs - fov)` broadcasts to (a, a) (fov_env.py:166-167)
            t = t.reshape(-1, 1).expand(n, cols)
        t = t.reshape(n, cols) if cols else t.reshape(n)
        return t.to(self.device, non_blocking=True).contiguous()

    def _observe(self, action=None, action_type=None, mask=None, out=None):
        out = self._obs if out is None else out
        if self.kind == "base":
            self.pipe.observe_full(out)
        elif self.kind == "flexible" and self._ragged_packed:
            # no mask in the packed layout: every env is re-observed; an env without an action keeps its state
            self.pipe.fovea_packed(action, action_type=action_type, packed=self._packed, offsets=self._poff,
                                   loc_out=self._loc, res_out=self._res)
        elif self.kind == "flexible":
            self.pipe.fovea(action, action_type=action_type, mask=mask, out=out, loc_out=self._loc, res_out=self._res)
        else:
            self.pipe.fovea(action, mask=mask, out=out, loc_out=self._loc)
        return out

    def _out(self, t: torch.Tensor):
        return t.cpu().numpy() if self._numpy_out else t

    def _next_stage(self):
        st = getattr(self, "_stage", None)
        if st is not None and len(st) > 1:
            self._stage_i ^= 1
            cur = st[self._stage_i]
            self._h_frames, self._h_cmd, self._ev_copy = cur["frames"], cur["cmd"], cur["ev"]
            self.runner.set_frames(self._h_frames.numpy())
        self._ev_copy.synchronize()             # this set's previous screens have left the pinned buffer

    def _next_obs_buffer(self):
        self._obs_i = (self._obs_i + 1) % len(self._obs_bufs)
        self._obs = self._obs_bufs[self._obs_i]

    # below 1 MB of observations (the single-env wrappers: 113 KB) a pageable copy is as cheap as the pool's bookkeeping
    _HOST_POOL_MIN_ELEMS = 1 << 18

    def _ret_obs(self, obs):
        if self._ragged_packed:
            off = self._poff.cpu().numpy()
            res = self._res.cpu().numpy()
            flat = self._packed[:int(off[-1])]
            flat = flat.cpu().numpy() if self._numpy_out else flat.clone()   # (the packed buffer is not double-buffered)
            return [flat[int(off[i]):int(off[i + 1])].reshape(self.frame_stack, int(res[i, 0]), int(res[i, 1]))
                    for i in range(self.num_envs)]
        if self._numpy_out and self._pinned_host_obs:
            if self._h_obs is None or tuple(self._h_obs[0].shape) != tuple(obs.shape):
                self._h_obs = [torch.empty(tuple(obs.shape), dtype=obs.dtype, pin_memory=True) for _ in range(2)]
            self._h_obs_i ^= 1
            h = self._h_obs[self._h_obs_i]
            h.copy_(obs, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
            return h.numpy()
        if self._numpy_out and self._host_pool is not None and obs.numel() >= self._HOST_POOL_MIN_ELEMS:
            h = self._host_pool.take(obs.shape, obs.dtype)
            if h is not None:
                h.copy_(obs, non_blocking=True)
                torch.cuda.current_stream(self.device).synchronize()
                return self._host_pool.hand_out(h)
        if self._numpy_out:
            return self._out(obs)
        return obs.clone() if self._copy_obs else obs

    def _info(self, raw_reward):
        info = {"raw_reward": np.asarray(raw_reward, dtype=np.float64).copy(),
                "reward": self.cumulative_reward.copy(), "ep_len": self.ep_len.copy()}
        if self.kind != "base":
            # host outputs: NumPy int64 like the reference's info["fov_loc"].  Device outputs (args.device set): int64
            # DEVICE tensors - no device-to-host copy, hence no synchronisation inside step(): the next step's emulation
            # then overlaps this step's last H2D chunk and kernels
            if self._numpy_out:
                info["fov_loc"] = self._loc.cpu().numpy().astype(np.int64)
                if self.kind == "flexible":
                    info["fov_res"] = self._res.cpu().numpy().astype(np.int64)
            else:
                info["fov_loc"] = self._loc.to(torch.int64)
                if self.kind == "flexible":
                    info["fov_res"] = self._res.to(torch.int64)
        return self._extra_info(info)

    @staticmethod
    def _with_masks(info, n):
        out = {}
        for k, v in info.items():
            out[k] = v
            out["_" + k] = np.ones(n, bool)
        return out

    # ------------------------------------------------------------------ API
    def reset(self, seed=None, options=None):
        """Reset every env (the reference ignores seed/options too, atari_env.py:150-152)."""
        if self._loop is not None:
            self._next_obs_buffer()
            fov = self.kind != "base"
            self._loop.reset(self._obs, self._loc if fov else None, self._res if self.kind == "flexible" else None)
            self.cumulative_reward[:] = 0
            self.ep_len[:] = 0
            self._was_reset = True
            return self._ret_obs(self._obs), self._with_masks(self._info(np.zeros(self.num_envs)), self.num_envs)
        for st in getattr(self, "_stage", []):
            st["ev"].synchronize()
        self._ev_copy.synchronize()
        for rs in self._rsets:
            rs["ev"].synchronize()
        cmd = self.runner.reset(out=self._h_rframes.numpy())
        self._upload_reset_all(cmd)
        self._ingest()
        self.cumulative_reward[:] = 0
        self.ep_len[:] = 0
        if self.kind != "base":
            self.pipe.fovea_reset()
        # a fresh output buffer: the observation a caller still holds from the previous step() (valid until the step after
        # next) must not be overwritten by the reset observation
        self._next_obs_buffer()
        obs = self._observe()
        self._release_dset()
        self._was_reset = True
        return self._ret_obs(obs), self._with_masks(self._info(np.zeros(self.num_envs)), self.num_envs)

    def step(self, action):
        if not self._was_reset:
            raise RuntimeError("call reset() before step()")
        n = self.num_envs
        if self.kind == "base":
            motor, sens, stype = action, None, None
        else:
            motor = action["motor_action"]
            sens = self._as_device_action(action["sensory_action"], 2)
            stype = None
            if self.kind == "flexible":
                stype = self._as_device_action(action["sensory_action_type"], 0).to(torch.int32)
        if isinstance(motor, torch.Tensor):
            motor = motor.detach().cpu().numpy()
        if self._loop is not None:
            return self._step_native(motor, sens, stype)
        self._next_stage()                      # the other pinned set; waits only for the copy issued from it two steps ago
        self._next_dset()                       # the other device screen buffer
        self._next_obs_buffer()
        chunk = int(getattr(self.args, "h2d_chunk_envs", 0) or 0)
        if chunk > 0 and hasattr(self.runner, "step_begin"):
            # native runner: chunk c's screens cross PCIe while chunk c+1 is still emulating
            nc = self.runner.step_begin(motor, chunk)
            for c in range(nc):
                self.runner.step_wait(c)
                lo, hi = c * chunk, min(n, (c + 1) * chunk)
                self._d_frames[lo:hi].copy_(self._h_frames[lo:hi], non_blocking=True)
            reward, done, cmd, raw = self.runner.step_finish()
            self._h_cmd.numpy()[:] = cmd
            self._d_cmd.copy_(self._h_cmd, non_blocking=True)
            self._ev_copy.record(torch.cuda.current_stream(self.device))
        else:
            reward, done, cmd, raw = self.runner.step(motor)
            self._upload(cmd)
        if self._ragged_packed and sens is not None:
            # packed ragged observations: ingest + state / scan + crops as one ABI call, two launches (agx_step_flexible_packed)
            self.pipe.step_flexible_packed(self._d_frames, self._d_cmd, sens, action_type=stype, packed=self._packed,
                                           offsets=self._poff, loc_out=self._loc, res_out=self._res)
            obs = self._obs
        else:
            self._ingest()
            obs = self._observe(sens, stype)
        self.ep_len += 1
        self.cumulative_reward += raw                   # unclipped, fov_env.py:62
        info = self._info(raw)
        truncated = np.zeros(n, bool)                   # always False, atari_env.py:145
        infos = self._with_masks(info, n)
        if self.autoreset and done.any():
            idx = np.nonzero(done)[0]
            # env.reset() of the done envs inside the same step (SyncVectorEnv, gymnasium<1.0).  Host first (emulators, pinned
            # staging), then ONE batch of device work: uploads, the gathers of the terminal observations / infos (the kernels
            # below overwrite them), ingest of the reset screens, masked re-observation.
            mask, d_idx = self._reset_subset(idx)
            final_obs = np.empty(n, dtype=object)
            final_info = np.empty(n, dtype=object)
            if self._ragged_packed:
                cur = self._ret_obs(obs)
                fo = [cur[i].copy() if isinstance(cur[i], np.ndarray) else cur[i].clone() for i in idx]
            else:
                fo = self._out(obs.index_select(0, d_idx))
            # device-tensor info entries (fov_loc / fov_res with device outputs): one gather per key, rows handed out as views
            gathered = {key: val.index_select(0, d_idx) for key, val in info.items() if isinstance(val, torch.Tensor)}
            for k, i in enumerate(idx):
                final_obs[i] = fo[k]
                final_info[i] = {key: (gathered[key][k] if key in gathered else
                                       (val[i].copy() if isinstance(val[i], np.ndarray) else val[i]))
                                 for key, val in info.items()}
            self._ingest(self._d_rcmd)
            self.cumulative_reward[idx] = 0
            self.ep_len[idx] = 0
            if self.kind == "base":
                self._observe()
            else:
                self.pipe.fovea_reset(mask)
                self._observe(None, None, mask=mask)
            rinfo = self._info(np.zeros(n))
            for key in info:
                if isinstance(info[key], torch.Tensor):
                    infos[key] = torch.where(mask.bool().reshape((n,) + (1,) * (info[key].ndim - 1)), rinfo[key], info[key])
                else:
                    infos[key] = np.where(done.reshape((n,) + (1,) * (info[key].ndim - 1)), rinfo[key], info[key])
            infos["final_observation"] = final_obs
            infos["_final_observation"] = done.copy()
            infos["final_info"] = final_info
            infos["_final_info"] = done.copy()
        self._release_dset()
        return self._ret_obs(obs), reward, done, truncated, infos

    def _step_native(self, motor, sens, stype):
        """step() through the native loop: one C call does emulators -> staging -> H2D -> ingest -> fovea -> autoreset; what is
        left here is the bookkeeping the reference's RecordWrapper / SyncVectorEnv do in Python (counters, info dicts)."""
        from .pipeline import _DT
        from .runner import check_motor_actions
        n = self.num_envs
        fov = self.kind != "base"
        motor = check_motor_actions(motor, self.runner.num_actions).reshape(n)
        self._next_obs_buffer()
        obs = self._obs
        dt = 0
        if sens is not None:
            if sens.dtype not in _DT:
                raise TypeError(f"sensory action dtype {sens.dtype} not supported (f32/f64/i32/i64)")
            dt = _DT[sens.dtype]
        reward, raw, done, idx, fo, fl, fr = self._loop.step(motor, sens, dt, stype, obs, self._loc if fov else None,
                                                             self._res if self.kind == "flexible" else None)
        self.ep_len += 1
        self.cumulative_reward += raw                   # unclipped, fov_env.py:62
        truncated = np.zeros(n, bool)                   # always False, atari_env.py:145
        info = {"raw_reward": raw.copy(), "reward": self.cumulative_reward.copy(), "ep_len": self.ep_len.copy()}
        k = len(idx)
        final = None
        if self.autoreset and k:
            # terminal observations / infos of the envs that ended an episode: rows of the loop's side buffers (cloned: the loop
            # reuses them next step), handed out as views like the Python loop's index_select rows
            fo = fo.clone()
            gathered = {}
            if fl is not None:
                gathered["fov_loc"] = fl.to(torch.int64)
            if fr is not None:
                gathered["fov_res"] = fr.to(torch.int64)
            final_obs = np.empty(n, dtype=object)
            final_info = np.empty(n, dtype=object)
            for j, i in enumerate(idx):
                final_obs[i] = fo[j]
                fi = {key: (val[i].copy() if isinstance(val[i], np.ndarray) else val[i]) for key, val in info.items()}
                for key, val in gathered.items():
                    fi[key] = val[j]
                final_info[i] = fi
            final = (final_obs, final_info)
            self.cumulative_reward[idx] = 0
            self.ep_len[idx] = 0
            # the returned infos carry the reset values for those envs (SyncVectorEnv overwrites them with the reset infos)
            info["raw_reward"][idx] = 0
            info["reward"][idx] = 0
            info["ep_len"][idx] = 0
        if fov:
            # self._loc / self._res: the step's values, overwritten by the masked re-observation for the envs that were reset
            info["fov_loc"] = self._loc.to(torch.int64)
            if self.kind == "flexible":
                info["fov_res"] = self._res.to(torch.int64)
        infos = self._with_masks(self._extra_info(info), n)
        if final is not None:
            infos["final_observation"] = final[0]
            infos["_final_observation"] = done.copy()
            infos["final_info"] = final[1]
            infos["_final_info"] = done.copy()
        return self._ret_obs(obs), reward, done, truncated, infos

    def reset_envs(self, idx):
        """Reset only the envs in `idx` (what a caller without autoreset does after `done`)."""
        idx = [int(i) for i in idx]
        n = self.num_envs
        if self._loop is not None:
            prev = self._obs
            self._next_obs_buffer()
            if len(self._obs_bufs) > 1:
                self._obs.copy_(prev)                   # the envs that are not reset keep their observation in the fresh buffer
            fov = self.kind != "base"
            self._loop.reset_envs(idx, self._obs, self._loc if fov else None, self._res if self.kind == "flexible" else None)
            self.cumulative_reward[idx] = 0
            self.ep_len[idx] = 0
            self._was_reset = True
            return self._ret_obs(self._obs), self._with_masks(self._info(np.zeros(n)), n)
        mask, _ = self._reset_subset(idx)
        self._ingest(self._d_rcmd)
        self.cumulative_reward[idx] = 0
        self.ep_len[idx] = 0
        # a fresh output buffer, as in reset(): the terminal observation the caller holds from step() stays untouched
        prev = self._obs
        self._next_obs_buffer()
        if self.kind != "base" and len(self._obs_bufs) > 1:
            self._obs.copy_(prev)                                   # masked observe: the other envs keep their observation
        if self.kind == "base":
            obs = self._observe()
        else:
            self.pipe.fovea_reset(mask)
            obs = self._observe(None, None, mask=mask)
        self._release_dset()
        self._was_reset = True
        return self._ret_obs(obs), self._with_masks(self._info(np.zeros(n)), n)

    def render(self, index=0):
        return self.runner.render(index)

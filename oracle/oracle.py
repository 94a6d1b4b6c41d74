"""CPU oracle for the active-gym Atari + 2-D fovea observation path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The
product path (``active-gym_amd/``) never imports this module and has no CPU
fallback.

What it restates (file:line relative to the reference checkout):

* a1  ``AtariEnv._get_state``            atari_env.py:73-75   -> :func:`ale_luminance`, :func:`cv_resize_linear_u8`, :func:`get_state_u8`
* a2  ``AtariEnv._step``                 atari_env.py:119-148 -> :class:`AtariEnvOracle.step`
* a3  ``AtariEnv._reset/_reset_buffer``  atari_env.py:80-117  -> :class:`AtariEnvOracle.reset`
* a4  ``RecordWrapper``                  fov_env.py:29-67     -> :class:`RecordOracle`
* a5-a9  ``FixedFovealEnv``              fov_env.py:107-234   -> :class:`FixedFovealOracle`
* a10-a11 ``FlexibleFovealEnv``          fov_env.py:236-355   -> :class:`FlexibleFovealOracle`
* a12 ``FixedFovealPeripheralEnv``       fov_env.py:358-388   -> :class:`PeripheralOracle`

Third-party arithmetic that is NOT in the reference tree and is restated here
from its published algorithm:

* ``torchvision.transforms.Resize`` on a float tensor (unpinned version,
  reference setup.py:17) = ``torch.nn.functional.interpolate(mode="bilinear",
  align_corners=False, antialias=A)``.  :func:`resize_bilinear` restates both
  the plain and the antialiased (triangle filter) arithmetic in NumPy float64;
  ``tests/test_oracle_resize.py`` pins it against torch's own CPU kernels and
  ``tests/golden/fovea_*.npz`` pins the wrappers' use of it against the
  reference's ``fov_env.py`` executed in the build container.
* OpenCV ``cv2.resize(..., INTER_LINEAR)`` on 8-bit input (opencv-python is
  not even declared by the reference) — 11-bit fixed-point coefficients,
  ``imgproc/src/resize.cpp``.  **parity unpinned**: cv2 is absent from the
  build image and no reference fixture covers it; only known-answer tests.
* ALE ``getScreenGrayscale`` luminance, ``round(.2989 R + .5870 G + .1140 B)``
  (ALE ``ColourPalette``), **parity unpinned** for the same reason.
"""
from __future__ import annotations

import collections
from typing import Callable, Optional, Sequence, Tuple

import numpy as np

RAW_H, RAW_W = 210, 160

# ---------------------------------------------------------------------------
# a1: ALE luminance + OpenCV fixed-point bilinear  (atari_env.py:73-75)
# ---------------------------------------------------------------------------


def ale_luminance(rgb: np.ndarray) -> np.ndarray:
    """ALE ``getScreenGrayscale`` applied to an RGB screen: per pixel
    ``(uint8) round(r*0.2989 + g*0.5870 + b*0.1140)`` in C double arithmetic
    (C ``round`` = half away from zero).  rgb: u8[..., 3] -> u8[...]."""
    rgb = np.asarray(rgb)
    assert rgb.dtype == np.uint8 and rgb.shape[-1] == 3
    r = rgb[..., 0].astype(np.float64)
    g = rgb[..., 1].astype(np.float64)
    b = rgb[..., 2].astype(np.float64)
    x = (r * 0.2989 + g * 0.5870) + b * 0.1140
    fl = np.floor(x)
    out = np.where((x - fl) >= 0.5, fl + 1.0, fl)
    return out.astype(np.uint8)


def _cv_linear_tables(src: int, dst: int):
    """Per-axis tables of OpenCV's 8-bit INTER_LINEAR (resize.cpp): for every
    destination index the two source indices and the two 11-bit coefficients.

    ``inv_scale = dst/src`` (double), ``scale = 1/inv_scale``;
    ``f = (float)((d+0.5)*scale - 0.5)``; ``s = floor(f)``; ``f -= s`` (float);
    coefficients ``saturate_cast<short>(c * 2048)`` = round-half-even of the
    float product.  The x-axis clamps (s<0 -> s=0,f=0; s>=src-1 -> s=src-1,f=0)
    and the y-axis row clipping are both expressed through (i0, i1, c0, c1):
    on the y axis OpenCV keeps the coefficients and clips the row indices,
    on the x axis it zeroes the fraction — see ``axis`` users below.
    """
    inv_scale = float(dst) / float(src)
    scale = 1.0 / inv_scale
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _coef(f: np.ndarray):
    one = np.float32(1.0)
    c0 = np.rint((one - f) * np.float32(2048.0)).astype(np.int64)
    c1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return c0, c1


def cv_tables_x(src_w: int, dst_w: int):
    s, f = _cv_linear_tables(src_w, dst_w)
    f = f.copy()
    lo = s < 0
    s[lo] = 0
    f[lo] = 0
    hi = s >= src_w - 1
    s[hi] = src_w - 1
    f[hi] = 0
    a0, a1 = _coef(f)
    s1 = np.minimum(s + 1, src_w - 1)
    return s.astype(np.int32), s1.astype(np.int32), a0.astype(np.int32), a1.astype(np.int32)


def cv_tables_y(src_h: int, dst_h: int):
    s, f = _cv_linear_tables(src_h, dst_h)
    b0, b1 = _coef(f)
    s0 = np.clip(s, 0, src_h - 1)
    s1 = np.clip(s + 1, 0, src_h - 1)
    return s0.astype(np.int32), s1.astype(np.int32), b0.astype(np.int32), b1.astype(np.int32)


def cv_resize_linear_u8(src: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """``cv2.resize(src, dsize, interpolation=cv2.INTER_LINEAR)`` for
    single-channel u8 images.  ``dsize`` is OpenCV's ``(width, height)``.
    src: u8[..., H, W] (leading dims are batched) -> u8[..., dh, dw].

    Horizontal pass: ``D = S[sx]*a0 + S[sx+1]*a1`` (int32);
    vertical pass: ``u8((((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2)``.
    """
    src = np.asarray(src)
    assert src.dtype == np.uint8
    dw, dh = int(dsize[0]), int(dsize[1])
    H, W = src.shape[-2], src.shape[-1]
    x0, x1, a0, a1 = cv_tables_x(W, dw)
    y0, y1, b0, b1 = cv_tables_y(H, dh)
    s = src.astype(np.int64)
    r0 = s[..., y0, :]
    r1 = s[..., y1, :]
    h0 = r0[..., x0] * a0 + r0[..., x1] * a1
    h1 = r1[..., x0] * a0 + r1[..., x1] * a1
    b0c = b0[:, None]
    b1c = b1[:, None]
    v = (((b0c * (h0 >> 4)) >> 16) + ((b1c * (h1 >> 4)) >> 16) + 2) >> 2
    return (v & 0xFF).astype(np.uint8)


def get_state_u8(rgb: np.ndarray, obs_size: Tuple[int, int]) -> np.ndarray:
    """The integer part of ``AtariEnv._get_state`` (atari_env.py:73-75) on an
    RGB screen: luminance then ``cv2.resize(gray, obs_size)``.  ``obs_size`` is
    handed to cv2 as ``dsize=(width,height)`` exactly as the reference does, so
    the result has shape ``(obs_size[1], obs_size[0])`` — consistent only for
    square sizes."""
    return cv_resize_linear_u8(ale_luminance(rgb), (obs_size[0], obs_size[1]))


def u8_to_unit(x: np.ndarray) -> np.ndarray:
    """``state.astype(np.float32) / 255.`` (atari_env.py:75): float32 divide."""
    return np.asarray(x).astype(np.float32) / np.float32(255.0)


# ---------------------------------------------------------------------------
# torchvision Resize == torch interpolate(bilinear, align_corners=False, antialias=A)
# ---------------------------------------------------------------------------


def _taps_plain(n_in: int, n_out: int):
    """ATen ``area_pixel_compute_source_index`` (align_corners=False):
    f = max(scale*(i+0.5)-0.5, 0); i0=floor(f); i1=i0+(i0<in-1); l1=f-i0."""
    scale = float(n_in) / float(n_out)
    i = np.arange(n_out, dtype=np.float64)
    f = np.maximum(scale * (i + 0.5) - 0.5, 0.0)
    i0 = np.minimum(np.floor(f).astype(np.int64), n_in - 1)
    i1 = i0 + (i0 < n_in - 1)
    l1 = f - i0
    l0 = 1.0 - l1
    return i0, i1, l0, l1


def _weights_aa(n_in: int, n_out: int) -> np.ndarray:
    """ATen ``_compute_indices_weights_aa`` for the bilinear (triangle) filter,
    as a dense [n_out, n_in] float64 matrix."""
    scale = float(n_in) / float(n_out)
    support = scale if scale >= 1.0 else 1.0
    invscale = 1.0 / scale if scale >= 1.0 else 1.0
    Wm = np.zeros((n_out, n_in), dtype=np.float64)
    for i in range(n_out):
        center = scale * (i + 0.5)
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), n_in)
        j = np.arange(xmin, xmax, dtype=np.float64)
        w = np.maximum(0.0, 1.0 - np.abs((j - center + 0.5) * invscale))
        tot = w.sum()
        if tot != 0.0:
            w = w / tot
        Wm[i, xmin:xmax] = w
    return Wm


def resize_bilinear(x: np.ndarray, size: Sequence[int], antialias: bool) -> np.ndarray:
    """``F.interpolate(x[None], size, mode="bilinear", align_corners=False,
    antialias=antialias)[0]`` on float64 [..., H, W] in NumPy."""
    x = np.asarray(x, dtype=np.float64)
    oh, ow = int(size[0]), int(size[1])
    H, W = x.shape[-2], x.shape[-1]
    if antialias:
        Wy = _weights_aa(H, oh)
        Wx = _weights_aa(W, ow)
        t = np.einsum("...hw,jw->...hj", x, Wx)      # width pass first (ATen order)
        return np.einsum("ih,...hj->...ij", Wy, t)
    y0, y1, wy0, wy1 = _taps_plain(H, oh)
    x0, x1, wx0, wx1 = _taps_plain(W, ow)
    top = x[..., y0, :]
    bot = x[..., y1, :]
    h0 = wx0 * top[..., x0] + wx1 * top[..., x1]
    h1 = wx0 * bot[..., x0] + wx1 * bot[..., x1]
    return wy0[:, None] * h0 + wy1[:, None] * h1


def tv_resize(x: np.ndarray, size: Sequence[int], antialias: bool) -> np.ndarray:
    """``torchvision.transforms.Resize(size)(tensor[C,H,W])``: returns the
    input unchanged when the size already matches, else interpolate."""
    x = np.asarray(x, dtype=np.float64)
    if (x.shape[-2], x.shape[-1]) == (int(size[0]), int(size[1])):
        return x
    return resize_bilinear(x, size, antialias)


# ---------------------------------------------------------------------------
# a2/a3: AtariEnv control flow (atari_env.py:41-172), per env, float64 like the reference
# ---------------------------------------------------------------------------


class AtariEnvOracle:
    """Follows ``AtariEnv`` op for op over an injected emulator object with the
    atari_py surface (``act, game_over, lives, reset_game`` + either
    ``getScreenGrayscale`` or ``getScreenRGB``).

    ``noop_fn()`` stands for ``random.randrange(30)`` (atari_env.py:96, global
    ``random``) so tests can inject the recorded counts.  ``resize_fn(gray)``
    stands for ``cv2.resize(gray, obs_size, INTER_LINEAR)``; default is the
    fixed-point restatement above.
    """

    def __init__(self, ale, n_actions_minimal: Sequence[int], obs_size=(84, 84),
                 frame_stack=4, action_repeat=4, clip_reward=False,
                 noop_fn: Optional[Callable[[], int]] = None,
                 resize_fn: Optional[Callable[[np.ndarray], np.ndarray]] = None, prefer_rgb: bool = False):
        self.prefer_rgb = prefer_rgb            # luminance of getScreenRGB instead of getScreenGrayscale
        self.ale = ale
        acts = list(n_actions_minimal)
        self.actions = dict(zip(range(len(acts)), acts))   # atari_env.py:51-52
        self.lives = 0
        self.life_termination = False
        self.frame_stack = frame_stack
        self.action_repeat = action_repeat
        self.state_buffer = collections.deque([], maxlen=frame_stack)
        self.training = True                                # atari_env.py:58 (args.training ignored)
        self.obs_size = tuple(obs_size)
        self.clip_reward = clip_reward
        self.noop_fn = noop_fn or (lambda: 0)
        self.resize_fn = resize_fn or (lambda g: cv_resize_linear_u8(g, (self.obs_size[0], self.obs_size[1])))

    def _gray(self):
        if hasattr(self.ale, "getScreenGrayscale") and not self.prefer_rgb:
            g = np.asarray(self.ale.getScreenGrayscale())
            return g[..., 0] if g.ndim == 3 else g
        return ale_luminance(np.asarray(self.ale.getScreenRGB()))

    def _get_state(self):                                   # atari_env.py:73-75
        return u8_to_unit(self.resize_fn(self._gray()))

    def _reset_buffer(self):                                # atari_env.py:80-82
        for _ in range(self.frame_stack):
            self.state_buffer.append(np.zeros(self.obs_size))

    def reset(self):                                        # atari_env.py:84-117
        if self.life_termination:
            self.life_termination = False
            self.ale.act(0)
        else:
            self._reset_buffer()
            self.ale.reset_game()
            for _ in range(self.noop_fn()):
                self.ale.act(0)
                if self.ale.game_over():
                    self.ale.reset_game()
        if len(self.actions) >= 3:
            self.ale.act(1)
            if self.ale.game_over():
                self.ale.reset_game()
                self.ale.act(2)
            if self.ale.game_over():
                self.ale.reset_game()
        observation = self._get_state()
        self.state_buffer.append(observation)
        self.lives = self.ale.lives()
        state = np.stack(self.state_buffer, axis=0)
        return state, {"raw_reward": 0}

    def step(self, action):                                 # atari_env.py:119-148
        frame_buffer = np.zeros((2, *self.obs_size))
        reward, done = 0, False
        for t in range(self.action_repeat):
            reward += self.ale.act(self.actions.get(action))
            if t == 2:
                frame_buffer[0] = self._get_state()
            elif t == 3:
                frame_buffer[1] = self._get_state()
            done = self.ale.game_over()
            if done:
                break
        observation = frame_buffer.max(0)
        self.state_buffer.append(observation)
        if self.training:
            lives = self.ale.lives()
            if lives < self.lives and lives > 0:
                self.life_termination = not done
                done = True
            self.lives = lives
        state = np.stack(self.state_buffer, axis=0)
        return_reward = np.sign(reward) if self.clip_reward else reward
        return state, return_reward, done, False, {"raw_reward": reward}

    def train(self):
        self.training = True

    def eval(self):
        self.training = False


class RecordOracle:
    """``RecordWrapper`` bookkeeping (fov_env.py:29-67), recording branch omitted."""

    def __init__(self, env):
        self.env = env
        self.cumulative_reward = 0
        self.ep_len = 0

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self):
        state, info = self.env.reset()
        self.cumulative_reward = 0
        self.ep_len = 0
        info["reward"] = self.cumulative_reward
        info["ep_len"] = self.ep_len
        return state, info

    def step(self, action):
        state, r, done, trunc, info = self.env.step(action)
        self.ep_len += 1
        self.cumulative_reward += info.get("raw_reward", r)
        info["reward"] = self.cumulative_reward
        info["ep_len"] = self.ep_len
        return state, r, done, trunc, info


# ---------------------------------------------------------------------------
# a5-a12: the three fovea wrappers as pure per-env state machines over full_state
# ---------------------------------------------------------------------------


class FixedFovealOracle:
    """``FixedFovealEnv`` (fov_env.py:107-234) minus gym plumbing: owns
    ``fov_loc`` and maps (full_state, sensory_action) -> fov_state."""

    def __init__(self, obs_size, fov_size, fov_init_loc, sensory_action_mode,
                 resize_to_full, mask_out=False, sensory_action_space=None,
                 antialias=True):
        self.obs_size = tuple(obs_size)
        self.fov_size = tuple(fov_size)
        self.fov_init_loc = tuple(fov_init_loc)
        assert (np.array(self.fov_size) < np.array(self.obs_size)).all()      # :112
        self.mode = sensory_action_mode
        if self.mode == "relative":
            self.sensory_action_space = np.array(sensory_action_space)         # :116
        else:
            self.sensory_action_space = np.array(self.obs_size) - np.array(self.fov_size)  # :118
        self.resize_to_full = bool(resize_to_full)
        self.mask_out = bool(mask_out)
        self.antialias = bool(antialias)
        self.init_loc()

    def init_loc(self):                                                        # :149-150
        self.fov_loc = np.rint(np.array(self.fov_init_loc, copy=True)).astype(np.int32)

    def reset(self, full_state):                                               # :156-164
        self.init_loc()
        return self.get_fov_state(full_state)

    def _bound(self):
        return np.array(self.obs_size) - np.array(self.fov_size)

    def clip_to_valid_fov(self, loc):                                          # :166-167
        return np.rint(np.clip(loc, 0, self._bound())).astype(int)

    def clip_to_valid_sas(self, action):                                       # :169-170
        return np.rint(np.clip(action, *self.sensory_action_space)).astype(int)

    def _crop(self, full_state, hw):
        r, c = int(self.fov_loc[0]), int(self.fov_loc[1])
        return full_state[..., r:r + int(hw[0]), c:c + int(hw[1])]

    def get_fov_state(self, full_state):                                       # :172-185
        fov = self._crop(full_state, self.fov_size)
        if self.mask_out:
            mask = np.zeros_like(full_state)
            r, c = int(self.fov_loc[0]), int(self.fov_loc[1])
            mask[..., r:r + self.fov_size[0], c:c + self.fov_size[1]] = fov
            return mask
        if self.resize_to_full:
            return tv_resize(fov, self.obs_size, self.antialias)
        return fov

    def update_loc(self, action):                                              # :187-199
        action = np.asarray(action)
        if self.mode == "absolute":
            self.fov_loc = self.clip_to_valid_fov(action)
        else:
            d = self.clip_to_valid_sas(action)
            self.fov_loc = self.clip_to_valid_fov(self.fov_loc + d)

    def step(self, full_state, action):
        self.update_loc(action)
        return self.get_fov_state(full_state)


class FlexibleFovealOracle(FixedFovealOracle):
    """``FlexibleFovealEnv`` (fov_env.py:240-355)."""
    FOV_LOC, FOV_RES = 0, 1

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.init_res()

    def init_res(self):                                                        # :250-251
        self.fov_res = np.rint(np.array(self.fov_size, copy=True)).astype(np.int32)

    def reset(self, full_state):                                               # :258-268
        self.init_loc()
        self.init_res()
        return self.get_fov_state(full_state)

    def _bound(self):                                                          # :270-271
        return np.array(self.obs_size) - np.array(self.fov_res)

    def get_fov_state(self, full_state):                                       # :283-298
        fov = self._crop(full_state, self.fov_res)
        if self.fov_res[0] > self.fov_size[0]:                                 # rows only (:286)
            fov = tv_resize(fov, self.fov_size, self.antialias)                # :277
            fov = tv_resize(fov, tuple(int(v) for v in self.fov_res), self.antialias)  # :278-279
        if self.mask_out:
            mask = np.zeros_like(full_state)
            r, c = int(self.fov_loc[0]), int(self.fov_loc[1])
            mask[..., r:r + int(self.fov_res[0]), c:c + int(self.fov_res[1])] = fov
            return mask
        if self.resize_to_full:
            return tv_resize(fov, self.obs_size, self.antialias)
        return fov

    def step(self, full_state, action, action_type=0):                         # :300-330
        action = np.asarray(action)
        if isinstance(action_type, np.ndarray):
            action_type = action_type.tolist()
            if isinstance(action_type, list):
                action_type = action_type[0]
        action_type = int(action_type)
        if action_type == self.FOV_LOC:
            self.update_loc(action)
        elif action_type == self.FOV_RES:
            self.fov_res = action.copy()                                       # no clip, no round (:323)
            self.fov_loc = self.clip_to_valid_fov(self.fov_loc)
        else:
            raise NotImplementedError
        return self.get_fov_state(full_state)


class PeripheralOracle(FixedFovealOracle):
    """``FixedFovealPeripheralEnv`` (fov_env.py:358-388)."""

    def __init__(self, obs_size, fov_size, fov_init_loc, sensory_action_mode,
                 peripheral_res, sensory_action_space=None, antialias=True, **_ignored):
        super().__init__(obs_size, fov_size, fov_init_loc, sensory_action_mode,
                         resize_to_full=True, mask_out=False,
                         sensory_action_space=sensory_action_space, antialias=antialias)
        self.peripheral_res = tuple(peripheral_res)

    def get_fov_state(self, full_state):                                       # :379-388
        fov = self._crop(full_state, self.fov_size)
        per = tv_resize(full_state, self.peripheral_res, self.antialias)       # :366-368,375-377
        per = np.array(tv_resize(per, self.obs_size, self.antialias))
        r, c = int(self.fov_loc[0]), int(self.fov_loc[1])
        per[..., r:r + self.fov_size[0], c:c + self.fov_size[1]] = fov
        return per


# ---------------------------------------------------------------------------
# Batched device-pipeline model: ring of u8 slots per env, pushed by "ingest"
# ---------------------------------------------------------------------------


class RingOracle:
    """Host model of what the HIP ingest kernel maintains: per env a deque of
    ``frame_stack`` u8 frames (the f32 k/255 values of the reference's
    ``state_buffer``, kept as their integer numerators).

    ``ingest`` is one ``state_buffer.append``:
      * step (atari_env.py:121-133): max over the first ``nvalid`` of the two
        frames sampled at t==2 / t==3, zeros when none was sampled;
      * reset (atari_env.py:91,111-112): ``clear`` zero-fills the ring first
        (full reset), then the single ``_get_state()`` frame is appended
        (``nvalid == 1``).
    """

    def __init__(self, n, frame_stack=4, obs_size=(84, 84)):
        self.n, self.fs, self.obs = n, frame_stack, tuple(obs_size)
        self.ring = [collections.deque([np.zeros(self.obs, np.uint8)] * frame_stack, maxlen=frame_stack)
                     for _ in range(n)]

    def ingest(self, frames, nvalid, clear=None, skip=None):
        frames = np.asarray(frames)
        for i in range(self.n):
            if skip is not None and skip[i]:
                continue
            if clear is not None and clear[i]:
                for _ in range(self.fs):
                    self.ring[i].append(np.zeros(self.obs, np.uint8))
            nv = int(nvalid[i])
            obs = np.zeros(self.obs, np.uint8)
            for f in range(min(nv, 2)):
                obs = np.maximum(obs, get_state_u8(frames[i, f], self.obs))
            self.ring[i].append(obs)

    def stack_u8(self):
        return np.stack([np.stack(list(d), 0) for d in self.ring], 0)

    def full_state(self):
        """[N, fs, H, W] float64 holding float32-exact k/255 (atari_env.py:75,143)."""
        return u8_to_unit(self.stack_u8()).astype(np.float64)


# ---------------------------------------------------------------------------
# DMC pixel front end (reference dmc_env.py:175-186,199-240)
# ---------------------------------------------------------------------------
# Third-party arithmetic restated here: OpenCV ``cvtColor(..., COLOR_BGR2GRAY)`` for 8-bit input
# (imgproc/src/color_rgb.simd.hpp, RGB2Gray<uchar>): fixed-point BT.601 luma, channel 0 weighted as blue.
#   OpenCV 4.x : (c0*3735 + c1*19235 + c2*9798 + (1 << 14)) >> 15      (BY15, GY15, RY15, gray_shift = 15)
#   OpenCV <=3 : (c0*1868 + c1*9617  + c2*4899 + (1 << 13)) >> 14      (B2Y,  G2Y,  R2Y,  yuv_shift  = 14)
# PARITY UNPINNED: cv2 and dm_control are absent from the image and the reference holds no fixture for this
# path; which generation applies depends on the installed OpenCV (the reference pins none, setup.py:14-19).
CV_GRAY = {"cv15": (3735, 19235, 9798, 15), "cv14": (1868, 9617, 4899, 14)}


def cv_bgr2gray_u8(img: np.ndarray, mode: str = "cv15") -> np.ndarray:
    """u8[..., 3] -> u8[...]; the reference hands this an RGB render, so R gets the blue weight."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.shape[-1] == 3
    k0, k1, k2, sh = CV_GRAY[mode]
    c = img.astype(np.int64)
    return ((c[..., 0] * k0 + c[..., 1] * k1 + c[..., 2] * k2 + (1 << (sh - 1))) >> sh).astype(np.uint8)


class DMCEnvOracle:
    """``DMCEnv`` (dmc_env.py:79-240) over any object with the dm_control surface the reference uses
    (``reset/step -> time_step``, ``physics.render``, ``physics.get_state``, ``action_spec``)."""

    def __init__(self, dmc_env, obs_size=(84, 84), frame_stack=3, action_repeat=4, clip_reward=False, camera_id=0,
                 gray_mode="cv15"):
        self.dmc_env, self.obs_size = dmc_env, tuple(obs_size)
        self.frame_stack, self.action_repeat, self.clip_reward = frame_stack, action_repeat, clip_reward
        self.camera_id, self.gray_mode = camera_id, gray_mode
        self.state_buffer = collections.deque([], maxlen=frame_stack)
        spec = dmc_env.action_spec()
        z = np.zeros(int(np.prod(spec.shape)), dtype=np.float32)
        self.true_low = (spec.minimum + z).astype(np.float32)            # _spec_to_box, dmc_env.py:27-47
        self.true_high = (spec.maximum + z).astype(np.float32)

    def _convert_action(self, action):                                  # dmc_env.py:166-173
        action = np.asarray(action).astype(np.float64)
        true_delta = self.true_high - self.true_low
        norm_delta = np.float32(1.0) - np.float32(-1.0)
        action = (action - np.float32(-1.0)) / norm_delta
        action = action * true_delta + self.true_low
        return action.astype(np.float32)

    def _get_obs(self):                                                 # dmc_env.py:175-186
        h, w = self.obs_size
        obs = self.dmc_env.physics.render(height=h, width=w, camera_id=self.camera_id)
        obs = cv_bgr2gray_u8(np.asarray(obs, dtype=np.uint8), self.gray_mode)
        return obs.astype(np.float32) / np.float32(255.0)

    def _info(self, ts, raw_reward=0):                                  # dmc_env.py:189-192
        return {"internal_state": self.dmc_env.physics.get_state().copy(), "discount": ts.discount, "raw_reward": raw_reward}

    def reset(self):                                                    # dmc_env.py:199-212
        for _ in range(self.frame_stack):
            self.state_buffer.append(np.zeros(self.obs_size))
        ts = self.dmc_env.reset()
        self.state_buffer.append(self._get_obs())
        return np.stack(self.state_buffer, axis=0), self._info(ts)

    def step(self, action):                                             # dmc_env.py:214-240
        action = np.asarray(action)
        assert action.shape == self.true_low.shape and np.all(action >= -1) and np.all(action <= 1)
        action = self._convert_action(action)
        assert np.all(action >= self.true_low) and np.all(action <= self.true_high)
        reward = 0
        for _ in range(self.action_repeat):
            ts = self.dmc_env.step(action)
            reward += ts.reward or 0
            done = ts.last()
            if done:
                break
        self.state_buffer.append(self._get_obs())
        ret = np.sign(reward) if self.clip_reward else reward
        return np.stack(self.state_buffer, axis=0), ret, done, False, self._info(ts, raw_reward=reward)

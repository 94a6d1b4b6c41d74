"""Trajectory recording of ``RecordWrapper`` (reference fov_env.py:34-37,70-102)
and ``AtariEnv.render`` resizing (reference atari_env.py:165-169).

The record buffer has the reference's keys (``rgb, state, action, reward, done,
truncated, info, return_reward`` + ``fov_size / fov_loc [/ fov_res /
peripheral_res]`` added by the fovea wrappers, fov_env.py:152-154,253-256,
370-373) and is saved with ``torch.save`` to ``<name>.pt``.  The reference
writes the frames to ``<name>.mp4`` with ``cv2.VideoWriter``; OpenCV is not a
dependency here, so the frames go to ``<name>.rgb.npy`` unless cv2 is
importable."""
from __future__ import annotations

import copy

import numpy as np
import torch


def resize_rgb_linear(rgb: np.ndarray, size) -> np.ndarray:
    """cv2.resize(rgb, size, INTER_LINEAR) for u8 HxWx3 with OpenCV's 11-bit fixed-point arithmetic
    (the same per-channel integer formula as the gray ingest kernel); host side, used for recording only."""
    dw, dh = int(size[0]), int(size[1])
    H, W = rgb.shape[:2]

    def axis(src, dst, is_x):
        scale = 1.0 / (float(dst) / float(src))
        d = np.arange(dst, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if is_x:
            lo, hi = s < 0, s >= src - 1
            s[lo] = 0
            f[lo] = 0
            s[hi] = src - 1
            f[hi] = 0
            i0, i1 = s, np.minimum(s + 1, src - 1)
        else:
            i0, i1 = np.clip(s, 0, src - 1), np.clip(s + 1, 0, src - 1)
        c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        c1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return i0, i1, c0, c1

    x0, x1, a0, a1 = axis(W, dw, True)
    y0, y1, b0, b1 = axis(H, dh, False)
    s = rgb.astype(np.int64)
    r0, r1 = s[y0], s[y1]
    h0 = r0[:, x0] * a0[None, :, None] + r0[:, x1] * a1[None, :, None]
    h1 = r1[:, x0] * a0[None, :, None] + r1[:, x1] * a1[None, :, None]
    v = (((b0[:, None, None] * (h0 >> 4)) >> 16) + ((b1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return (v & 0xFF).astype(np.uint8)


class Recorder:
    def __init__(self, wrapper):
        self.w = wrapper

    def _new_buffer(self):
        w = self.w
        w.prev_record_buffer = copy.deepcopy(w.record_buffer)
        w.record_buffer = {"rgb": [], "state": [], "action": [], "reward": [], "done": [], "truncated": [],
                           "info": [], "return_reward": []}

    def _save(self, state, action=None, reward=None, done=None, truncated=None, info=None, rgb=None,
              return_reward=None):                                           # fov_env.py:70-88
        b = self.w.record_buffer
        if (done is not None) and (not done):
            b["state"].append(state)
            b["rgb"].append(rgb)
        if action is not None:
            b["action"].append(action)
        if reward is not None:
            b["reward"].append(reward)
        if done is not None and len(b["state"]) > 1:
            b["done"].append(done)
        if truncated is not None:
            b["truncated"].append(truncated)
        if info is not None and len(b["state"]) > 1:
            b["info"].append(info)
        if return_reward is not None:
            b["return_reward"].append(return_reward)

    def on_reset(self, state, info, fovea=None):
        """`state` is the FULL-frame state of the wrapped base env (the reference's RecordWrapper sits below
        the fovea wrapper, fov_env.py:43-56), float64 like the reference's."""
        rgb = self.w.env.render()
        self._new_buffer()
        self._save(np.asarray(state, dtype=np.float64), done=False, info=info, rgb=rgb)
        if fovea is not None:
            # key order as the reference's reset_record_buffer builds it: fov_size, [peripheral_res,] fov_loc [, fov_res]
            # (fov_env.py:152-154, 253-256, 370-373)
            b = self.w.record_buffer
            b["fov_size"] = fovea.fov_size
            if hasattr(fovea, "peripheral_res"):
                b["peripheral_res"] = fovea.peripheral_res
            b["fov_loc"] = [info["fov_loc"]]
            if "fov_res" in info:
                b["fov_res"] = [info["fov_res"]]

    def on_step(self, state, action, cum_reward, done, truncated, info, return_reward, fovea=None):
        rgb = self.w.env.render()
        self._save(np.asarray(state, dtype=np.float64), action, cum_reward, done, truncated, info, rgb=rgb,
                   return_reward=return_reward)
        if fovea is not None and not done:
            b = self.w.record_buffer
            b["fov_loc"].append(info["fov_loc"])
            if "fov_res" in info:
                b["fov_res"].append(info["fov_res"])

    def save(self, file_path: str):                                          # fov_env.py:90-102
        buf = self.w.prev_record_buffer
        if buf is None:
            raise RuntimeError("no finished episode to save: the previous buffer is filled at the next reset()")
        frames = buf["rgb"]
        try:
            import cv2  # type: ignore
            video_path = file_path.replace(".pt", ".mp4")
            size = frames[0].shape[:2][::-1]
            vw = cv2.VideoWriter(video_path, cv2.VideoWriter_fourcc(*"mp4v"), 30, size)
            for f in frames:
                vw.write(f)
            vw.release()
        except ImportError:
            video_path = file_path.replace(".pt", ".rgb.npy")
            np.save(video_path, np.stack(frames))
        buf["rgb"] = video_path
        buf["state"] = [0] * len(buf["reward"])
        torch.save(buf, file_path)

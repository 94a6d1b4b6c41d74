"""Helpers to replay the committed golden vectors (tests/golden/*.npz)."""
from __future__ import annotations

import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fovea_case_names():
    return sorted(os.path.basename(p)[len("fovea_"):-4] for p in glob.glob(os.path.join(GOLDEN, "fovea_*.npz")))


def atari_case_names():
    return sorted(os.path.basename(p)[len("atari_"):-4] for p in glob.glob(os.path.join(GOLDEN, "atari_*.npz")))


def load_fovea(name):
    z = np.load(os.path.join(GOLDEN, f"fovea_{name}.npz"), allow_pickle=False)
    c = {k: z[k] for k in z.files}
    for k in ("kind", "mode", "out_f64_dtype"):
        c[k] = str(c[k])
    for k in ("frame_stack",):
        c[k] = int(c[k])
    for k in ("resize_to_full", "mask_out", "antialias", "ragged"):
        c[k] = bool(c[k])
    steps = c["actions"].shape[0]
    if c["ragged"]:
        c["outs"] = [c[f"out_{i}"] for i in range(steps + 1)]
    else:
        c["outs"] = list(c["out"])
    c["steps"] = steps
    return c


def load_atari(name):
    z = np.load(os.path.join(GOLDEN, f"atari_{name}.npz"), allow_pickle=False)
    c = {k: z[k] for k in z.files}
    for k in ("seed", "frame_stack", "action_repeat", "n_actions"):
        c[k] = int(c[k])
    for k in ("clip_reward", "training", "fixed_fov"):
        c[k] = bool(c[k])
    return c


def unit64(u8):
    """float64 array holding float32(k)/255 — what the reference's state_buffer holds."""
    return (np.asarray(u8).astype(np.float32) / np.float32(255.0)).astype(np.float64)


def golden_tol(case):
    """Goldens stored as float32 lose <= half an ulp of 1.0; float64 ones are exact."""
    return 1e-7 if case["outs"][0].dtype == np.float32 else 1e-12

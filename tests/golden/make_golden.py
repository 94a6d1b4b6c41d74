#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by EXECUTING the reference's
own ``active_gym/fov_env.py`` and ``active_gym/atari_env.py`` (read from
/root/reference, never copied) in the build container.

The reference imports four third-party modules that are absent from the image
(cv2, gymnasium, torchvision, atari_py).  They are replaced *for this script
only* by minimal stand-ins pre-seeded into ``sys.modules``:

* ``gymnasium``  — ``Env``, ``Wrapper`` (attribute forwarding), ``spaces.Box /
  Discrete / Dict`` as plain data holders.  No arithmetic.
* ``torchvision.transforms.Resize(size)`` — unsqueeze -> ``torch.nn.functional.
  interpolate(size, mode="bilinear", align_corners=False, antialias=A)`` ->
  squeeze, with torchvision's same-size early return.  That IS the backend of
  torchvision's tensor path and torch is in the image; ``A`` (version
  dependent default) is emitted both ways.
* ``cv2`` — ``resize`` = identity on an already obs-sized screen.  The atari
  goldens therefore pin the reference's CONTROL FLOW only (frame sampling at
  t==2/3, zero fill, max, deque order, life-loss, resets, reward bookkeeping);
  OpenCV's fixed-point arithmetic stays "parity unpinned".
* ``atari_py`` — ``tests/fake_ale.ScriptedALE`` (an event script, not an emulator).

Only data (inputs + the reference's outputs) is written; run from the repo root:
    python tests/golden/make_golden.py
"""
from __future__ import annotations

import importlib.util
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("AGX_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REPO, "tests"))
from fake_ale import ScriptedALE  # noqa: E402

_STATE = {"antialias": True, "next_ale": None}


# ------------------------------------------------------------------ stand-ins
def _install_standins():
    # gymnasium
    gym = types.ModuleType("gymnasium")

    class Env:
        @property
        def unwrapped(self):
            return self

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

        @property
        def unwrapped(self):
            return self.env.unwrapped

        def __getattr__(self, name):
            if name.startswith("__") or name == "env":
                raise AttributeError(name)
            return getattr(self.env, name)

    class Box:
        """Data holder; with array bounds or a shape it also offers what dmc_env.py calls: seed / contains."""

        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype
            if shape is not None or np.ndim(low) > 0:
                shp = tuple(shape) if shape is not None else np.shape(low)
                self.shape = shp
                self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shp).copy()
                self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shp).copy()

        def seed(self, seed=None):
            return [seed]

        def contains(self, x):                  # gymnasium.spaces.Box.contains
            x = np.asarray(x)
            return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self.shape
                        and np.all(x >= self.low) and np.all(x <= self.high))

    class Discrete:
        def __init__(self, n):
            self.n = n

        def sample(self):
            return random.randrange(self.n)

    class Dict(dict):
        def __init__(self, d):
            super().__init__(d)

    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Box, spaces.Discrete, spaces.Dict = Box, Discrete, Dict
    gym.Env, gym.Wrapper, gym.spaces = Env, Wrapper, spaces
    gym.Space = object                       # annotation only (dmc_env.py:27)
    sys.modules["gymnasium"] = gym
    sys.modules["gymnasium.spaces"] = spaces

    # torchvision.transforms.Resize
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class Resize(torch.nn.Module):
        def __init__(self, size):
            super().__init__()
            self.size = tuple(int(s) for s in size)

        def forward(self, img):
            if tuple(img.shape[-2:]) == self.size:
                return img
            out = torch.nn.functional.interpolate(
                img.unsqueeze(0), size=self.size, mode="bilinear",
                align_corners=False, antialias=_STATE["antialias"])
            return out.squeeze(0)

    tvt.Resize = Resize
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt

    # cv2: the reference is always executed over the stand-in below (the control-flow goldens keep their meaning on
    # every machine); if the REAL library is importable (it is not in the build image) it is kept aside for make_cv2(),
    # whose arithmetic goldens pin the oracle's restatement of INTER_LINEAR and BGR2GRAY
    try:
        import cv2 as _real_cv2  # type: ignore
        _STATE["cv2_real"] = _real_cv2
    except ImportError:
        _STATE["cv2_real"] = None
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1

    def resize(img, dsize, interpolation=None):
        img = np.asarray(img)
        if img.ndim == 3 and img.shape[-1] == 1:
            img = img[..., 0]
        assert img.shape == (dsize[1], dsize[0]), (img.shape, dsize)
        return img.copy()

    cv2.resize = resize

    # VideoWriter: records what the reference writes (no encoding) - used by the record goldens only
    class VideoWriter:
        log = []

        def __init__(self, path, fourcc, fps, size):
            self.entry = {"path": path, "fourcc": fourcc, "fps": fps, "size": tuple(size), "frames": []}
            VideoWriter.log.append(self.entry)

        def write(self, frame):
            self.entry["frames"].append(np.asarray(frame).copy())

        def release(self):
            self.entry["released"] = True

    # cvtColor: the oracle's restatement of OpenCV 4.x BGR2GRAY (the DMC goldens pin CONTROL FLOW only)
    sys.path.insert(0, REPO)
    from oracle import oracle as _O
    cv2.COLOR_BGR2GRAY = 6
    cv2.cvtColor = lambda img, code: _O.cv_bgr2gray_u8(np.asarray(img), "cv15")
    cv2.VideoWriter = VideoWriter
    cv2.VideoWriter_fourcc = lambda *c: "".join(c)
    sys.modules["cv2"] = cv2

    _install_rest(gym)


def _install_rest(gym):
    # dm_control.suite / dm_env.specs (dmc_env.py:15-16)
    import fake_dmc
    dmc = types.ModuleType("dm_control")
    suite = types.ModuleType("dm_control.suite")
    suite.load = lambda **kw: _STATE["next_dmc"](kw)
    dmc.suite = suite
    sys.modules["dm_control"], sys.modules["dm_control.suite"] = dmc, suite
    dme = types.ModuleType("dm_env")
    specs = types.ModuleType("dm_env.specs")
    specs.Array, specs.BoundedArray = fake_dmc.Array, fake_dmc.BoundedArray
    dme.specs = specs
    sys.modules["dm_env"], sys.modules["dm_env.specs"] = dme, specs

    # atari_py
    ap = types.ModuleType("atari_py")
    ap.ALEInterface = lambda: _STATE["next_ale"]
    ap.get_game_path = lambda game: game
    sys.modules["atari_py"] = ap


def _load_reference():
    pkg = types.ModuleType("refpkg")
    pkg.__path__ = [os.path.join(REF, "active_gym")]
    sys.modules["refpkg"] = pkg
    mods = {}
    for name in ("fov_env", "atari_env", "dmc_env"):
        spec = importlib.util.spec_from_file_location(
            f"refpkg.{name}", os.path.join(REF, "active_gym", f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"refpkg.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["fov_env"], mods["atari_env"], mods["dmc_env"]


# ------------------------------------------------------------------ fovea goldens
class _Args:
    def __init__(self, **kw):
        self.record = False
        self.mask_out = False
        for k, v in kw.items():
            setattr(self, k, v)


def _make_states(rng, steps, fs, obs):
    """u8 numerators; reference sees float64 holding float32(k)/255."""
    return rng.integers(0, 256, size=(steps + 1, fs) + tuple(obs), dtype=np.uint8)


def _unit(u8):
    return (u8.astype(np.float32) / np.float32(255.0)).astype(np.float64)


def _fovea_case(fov_env, gym, kind, name, obs, fov, fs, mode, resize_to_full, mask_out,
                antialias, steps, seed, per=None, init_loc=(0, 0), sas=(-10.0, 10.0),
                out_dtype=np.float32, int_actions=False):
    rng = np.random.default_rng(seed)
    states_u8 = _make_states(rng, steps, fs, obs)
    _STATE["antialias"] = antialias

    class FakeBase(gym.Env):
        def __init__(self):
            self.obs_size = tuple(obs)
            self.frame_stack = fs
            self.action_space = gym.spaces.Discrete(4)
            self.t = 0

        def reset(self, seed=None, options=None):
            self.t = 0
            return _unit(states_u8[0]), {"raw_reward": 0}

        def step(self, action):
            self.t += 1
            return _unit(states_u8[self.t]), 1.0, False, False, {"raw_reward": 1.0}

        def render(self):
            return None

    args = _Args(fov_size=tuple(fov), fov_init_loc=tuple(init_loc), sensory_action_mode=mode,
                 sensory_action_space=sas, resize_to_full=resize_to_full, mask_out=mask_out,
                 peripheral_res=per)
    base = fov_env.RecordWrapper(FakeBase(), args)
    cls = {"fixed": fov_env.FixedFovealEnv, "flex": fov_env.FlexibleFovealEnv,
           "per": fov_env.FixedFovealPeripheralEnv}[kind]
    env = cls(base, args)

    hi = np.array(obs) - np.array(fov)
    actions, types_, outs, locs, ress = [], [], [], [], []
    o, info = env.reset()
    outs.append(np.asarray(o))
    locs.append(info["fov_loc"].copy())
    ress.append(np.asarray(info.get("fov_res", fov)).copy())
    cur_res = np.array(fov)
    for t in range(steps):
        atype = 0
        if kind == "flex" and rng.random() < 0.5:
            atype = 1
            # integer resolutions in [6, obs]: rows above and below fov rows, cols independent
            a = np.array([rng.integers(6, obs[0] + 1), rng.integers(6, obs[1] + 1)])
            if t == 1:
                a = np.array([fov[0] + 7, max(6, fov[1] - 5)])     # rows > fov rows, cols < fov cols
            if t == 3:
                a = np.array([obs[0], obs[1]])                     # full-frame window
        elif mode == "absolute":
            a = rng.uniform(-5.0, max(hi) + 6.0, size=2)
            if t % 3 == 1:
                a = np.floor(a) + 0.5                              # exercise round-half-even
            if int_actions:
                a = np.rint(a).astype(np.int64)
        else:
            a = rng.uniform(sas[0] - 4.0, sas[1] + 4.0, size=2)
            if t % 3 == 1:
                a = np.floor(a) + 0.5
        act = {"motor_action": 0, "sensory_action": a}
        if kind == "flex":
            act["sensory_action_type"] = np.array((atype,))
        o, r, d, tr, info = env.step(act)
        actions.append(np.asarray(a, dtype=np.float64))
        types_.append(atype)
        outs.append(np.asarray(o))
        locs.append(np.asarray(info["fov_loc"]).copy())
        ress.append(np.asarray(info.get("fov_res", fov)).copy())
        assert info["ep_len"] == t + 1 and info["reward"] == float(t + 1)
    rec = {
        "kind": kind, "obs_size": np.array(obs), "fov_size": np.array(fov), "frame_stack": fs,
        "mode": mode, "resize_to_full": resize_to_full, "mask_out": mask_out, "antialias": antialias,
        "peripheral_res": np.array(per if per else (0, 0)), "init_loc": np.array(init_loc),
        "sas": np.array(sas, dtype=np.float64), "states_u8": states_u8,
        "actions": np.array(actions), "action_types": np.array(types_, dtype=np.int64),
        "fov_loc": np.array(locs, dtype=np.int64), "fov_res": np.array(ress, dtype=np.int64),
        "out_f64_dtype": str(outs[0].dtype),
    }
    assert all(x.dtype == np.float64 for x in outs), "reference emits float64 (SURVEY §8a-Q1)"
    ragged = len({x.shape for x in outs}) > 1
    rec["ragged"] = ragged
    if ragged:
        for i, x in enumerate(outs):
            rec[f"out_{i}"] = x.astype(out_dtype)
    else:
        rec["out"] = np.stack(outs).astype(out_dtype)
    path = os.path.join(HERE, f"fovea_{name}.npz")
    np.savez_compressed(path, **rec)
    return path


def make_fovea(fov_env, gym):
    paths = []
    n = 0
    std = dict(obs=(84, 84), fov=(30, 30))
    for mode in ("absolute", "relative"):
        for tag, rtf, mo in (("resize", True, False), ("mask", False, True), ("raw", False, False)):
            n += 1
            paths.append(_fovea_case(fov_env, gym, "fixed", f"fixed_{tag}_{mode[:3]}", fs=2, mode=mode,
                                     resize_to_full=rtf, mask_out=mo, antialias=True, steps=5,
                                     seed=100 + n, init_loc=(10.4, 20.5), **std))
    # the headline config verbatim (fs=4), antialias off (== on for upscaling), int actions
    paths.append(_fovea_case(fov_env, gym, "fixed", "fixed_resize_abs_fs4_aa0", fs=4, mode="absolute",
                             resize_to_full=True, mask_out=False, antialias=False, steps=3, seed=200,
                             int_actions=True, **std))
    # mask_out wins over resize_to_full (fov_env.py:176-183)
    paths.append(_fovea_case(fov_env, gym, "fixed", "fixed_maskwins_abs", fs=2, mode="absolute",
                             resize_to_full=True, mask_out=True, antialias=True, steps=3, seed=201, **std))
    for aa in (False, True):
        paths.append(_fovea_case(fov_env, gym, "per", f"per_aa{int(aa)}", fs=2, mode="absolute",
                                 resize_to_full=False, mask_out=True, antialias=aa, steps=4, seed=300 + aa,
                                 per=(20, 20), **std))
        for tag, rtf, mo in (("resize", True, False), ("mask", False, True), ("raw", False, False)):
            paths.append(_fovea_case(fov_env, gym, "flex", f"flex_{tag}_aa{int(aa)}", fs=2, mode="absolute",
                                     resize_to_full=rtf, mask_out=mo, antialias=aa, steps=8,
                                     seed=400 + 10 * aa + len(tag), **std))
    paths.append(_fovea_case(fov_env, gym, "flex", "flex_resize_rel_aa1", fs=2, mode="relative",
                             resize_to_full=True, mask_out=False, antialias=True, steps=8, seed=450, **std))
    # small / non-square geometry, full float64 outputs
    small = dict(obs=(36, 48), fov=(10, 16), out_dtype=np.float64)
    paths.append(_fovea_case(fov_env, gym, "fixed", "small_fixed_resize", fs=3, mode="absolute",
                             resize_to_full=True, mask_out=False, antialias=True, steps=4, seed=500, **small))
    for aa in (False, True):
        paths.append(_fovea_case(fov_env, gym, "per", f"small_per_aa{int(aa)}", fs=3, mode="relative",
                                 resize_to_full=True, mask_out=False, antialias=aa, steps=4, seed=510 + aa,
                                 per=(9, 7), sas=(-6.0, 6.0), **small))
        paths.append(_fovea_case(fov_env, gym, "flex", f"small_flex_resize_aa{int(aa)}", fs=3, mode="absolute",
                                 resize_to_full=True, mask_out=False, antialias=aa, steps=8, seed=520 + aa,
                                 **small))
    return paths


# ------------------------------------------------------------------ atari control-flow goldens
def _atari_case(atari_env, name, seed, obs=(16, 16), fs=4, ar=4, clip=False, training=True,
                steps=260, n_actions=4, fixed_fov=False):
    ale = ScriptedALE(seed=seed, screen_hw=obs, n_actions=n_actions)
    _STATE["next_ale"] = ale
    random.seed(seed)
    noops = []
    orig = random.randrange

    def rec_randrange(n):
        v = orig(n)
        noops.append(v)
        return v

    atari_env.random.randrange = rec_randrange
    try:
        kw = dict(frame_stack=fs, action_repeat=ar, clip_reward=clip)
        if fixed_fov:
            kw.update(fov_size=(6, 6), fov_init_loc=(2, 3), sensory_action_mode="absolute",
                      resize_to_full=True)
        args = atari_env.AtariEnvArgs(game="scripted", seed=seed, obs_size=tuple(obs), **kw)
        env = atari_env.AtariFixedFovealEnv(args) if fixed_fov else atari_env.AtariBaseEnv(args)
        if not training:
            env.eval()
        arng = np.random.default_rng(seed + 1)
        motor = arng.integers(0, n_actions, size=steps)
        sens = arng.uniform(-2, 14, size=(steps, 2))
        states, rewards, dones, ep_len, cum, is_reset, fov_loc = [], [], [], [], [], [], []
        s, info = env.reset()
        states.append(s); rewards.append(0.0); dones.append(False)
        ep_len.append(info["ep_len"]); cum.append(info["reward"]); is_reset.append(True)
        fov_loc.append(info.get("fov_loc", np.zeros(2)))
        for t in range(steps):
            a = int(motor[t])
            if fixed_fov:
                s, r, d, tr, info = env.step({"motor_action": a, "sensory_action": sens[t]})
            else:
                s, r, d, tr, info = env.step(a)
            assert tr is False
            states.append(s); rewards.append(float(r)); dones.append(bool(d))
            ep_len.append(info["ep_len"]); cum.append(info["reward"]); is_reset.append(False)
            fov_loc.append(info.get("fov_loc", np.zeros(2)))
            if d:
                s, info = env.reset()
                states.append(s); rewards.append(0.0); dones.append(False)
                ep_len.append(info["ep_len"]); cum.append(info["reward"]); is_reset.append(True)
                fov_loc.append(info.get("fov_loc", np.zeros(2)))
    finally:
        atari_env.random.randrange = orig
    states = np.stack(states)
    assert states.dtype == np.float64
    rec = dict(seed=seed, obs_size=np.array(obs), frame_stack=fs, action_repeat=ar, clip_reward=clip,
               training=training, n_actions=n_actions, fixed_fov=fixed_fov, motor=motor, sens=sens,
               noops=np.array(noops, dtype=np.int64), rewards=np.array(rewards), dones=np.array(dones),
               ep_len=np.array(ep_len, dtype=np.int64), cum_reward=np.array(cum, dtype=np.float64),
               is_reset=np.array(is_reset), fov_loc=np.array(fov_loc, dtype=np.int64))
    if fixed_fov:
        rec["states_f64"] = states
    else:
        u8 = np.rint(states * 255.0).astype(np.uint8)
        assert np.array_equal((u8.astype(np.float32) / np.float32(255.0)).astype(np.float64), states), \
            "base-env states must be float32-exact k/255"
        rec["states_u8"] = u8
    path = os.path.join(HERE, f"atari_{name}.npz")
    np.savez_compressed(path, **rec)
    return path


def make_atari(atari_env):
    return [
        _atari_case(atari_env, "train_ar4", seed=11),
        _atari_case(atari_env, "eval_ar4_clip", seed=12, clip=True, training=False),
        _atari_case(atari_env, "train_ar3_fs3", seed=13, ar=3, fs=3, steps=120),
        _atari_case(atari_env, "train_ar1", seed=14, ar=1, steps=60),
        _atari_case(atari_env, "train_ar6_2act", seed=15, ar=6, n_actions=2, steps=160),
        _atari_case(atari_env, "fixedfov_train_ar4", seed=16, steps=100, fixed_fov=True),
    ]


# ------------------------------------------------------------------ record-buffer goldens
def _record_case(fov_env, gym, kind, name, seed, tmpdir):
    """RecordWrapper(record=True) under each fovea wrapper (fov_env.py:34-37,51-55,64-102,152-154,161-163,
    207,218-220,253-256,265-267,334-335,352-354,370-373): two episodes + the start of a third, then
    save_record_to_file() of the finished one.  Emits the driving sequence and both buffers."""
    import cv2
    obs, fov, fs = (12, 12), (4, 4), 2
    rng = np.random.default_rng(seed)
    T = 16
    states_u8 = rng.integers(0, 256, size=(T + 4, fs) + obs, dtype=np.uint8)
    rgbs = rng.integers(0, 256, size=(T + 4, 6, 5, 3), dtype=np.uint8)
    base_rewards = rng.integers(-2, 5, size=T + 4).astype(np.float64)
    done_at = {4, 11}                      # global step indices that end an episode
    _STATE["antialias"] = True
    clock = {"g": 0}                       # counts every reset/step of the base env -> index into states/rgbs

    class FakeBase(gym.Env):
        def __init__(self):
            self.obs_size, self.frame_stack = obs, fs
            self.action_space = gym.spaces.Discrete(4)

        def reset(self, seed=None, options=None):
            clock["g"] += 1
            return _unit(states_u8[clock["g"]]), {"raw_reward": 0}

        def step(self, action):
            clock["g"] += 1
            g = clock["g"]
            return _unit(states_u8[g]), float(np.sign(base_rewards[g])), (g in done_at), False, {"raw_reward": float(base_rewards[g])}

        def render(self):
            return rgbs[clock["g"]]

    args = _Args(fov_size=fov, fov_init_loc=(1, 2), sensory_action_mode="absolute", sensory_action_space=(-3.0, 3.0),
                 resize_to_full=True, mask_out=False, peripheral_res=(5, 5), record=True)
    base = fov_env.RecordWrapper(FakeBase(), args)
    env = base if kind == "base" else {"fixed": fov_env.FixedFovealEnv, "flex": fov_env.FlexibleFovealEnv,
                                       "per": fov_env.FixedFovealPeripheralEnv}[kind](base, args)
    clock["g"] = -1
    drive = []                              # (is_reset, motor, sens0, sens1, type)
    env.reset()
    drive.append((1, 0, 0, 0, 0))
    while clock["g"] < T - 1:
        motor = int(rng.integers(0, 4))
        sens = rng.integers(-1, 10, size=2)
        typ = int(rng.integers(0, 2)) if kind == "flex" else 0
        if typ == 1:
            sens = rng.integers(2, 9, size=2)
        if kind == "base":
            _, _, d, _, _ = env.step(motor)
        else:
            # the flexible env stores a FOV_RES action unrounded (fov_env.py:322) and slices with it: integers there
            act = {"motor_action": motor, "sensory_action": sens.astype(np.int64 if kind == "flex" else np.float64)}
            if kind == "flex":
                act["sensory_action_type"] = np.array((typ,))
            _, _, d, _, _ = env.step(act)
        drive.append((0, motor, int(sens[0]), int(sens[1]), typ))
        if d:
            env.reset()
            drive.append((1, 0, 0, 0, 0))
    rec = {"kind": kind, "obs_size": np.array(obs), "fov_size": np.array(fov), "frame_stack": fs, "init_loc": np.array((1, 2)),
           "peripheral_res": np.array((5, 5)), "states_u8": states_u8, "rgbs": rgbs, "base_rewards": base_rewards,
           "done_at": np.array(sorted(done_at)), "drive": np.array(drive, dtype=np.int64)}

    def dump(tag, buf):
        rec[f"{tag}_keys"] = np.array(sorted(buf.keys()))
        for k in ("rgb", "state", "action", "reward", "done", "truncated", "info", "return_reward", "fov_loc", "fov_res"):
            if k in buf:
                rec[f"{tag}_len_{k}"] = len(buf[k])
        rec[f"{tag}_rgb"] = np.stack(buf["rgb"])
        rec[f"{tag}_state"] = np.stack(buf["state"])
        assert rec[f"{tag}_state"].dtype == np.float64
        for k in ("action", "reward", "done", "truncated", "return_reward"):
            rec[f"{tag}_{k}"] = np.array(buf[k])
        for k in ("fov_loc", "fov_res"):
            if k in buf:
                rec[f"{tag}_{k}"] = np.array(buf[k], dtype=np.int64)
        for k in ("fov_size", "peripheral_res"):
            if k in buf:
                rec[f"{tag}_{k}"] = np.array(buf[k])
        infos = buf["info"]
        for k in ("raw_reward", "reward", "ep_len"):
            rec[f"{tag}_info_{k}"] = np.array([i[k] for i in infos], dtype=np.float64)
        rec[f"{tag}_info_keys"] = np.array(sorted(infos[-1].keys())) if infos else np.array([])
        if infos and "fov_loc" in infos[-1]:
            rec[f"{tag}_info_fov_loc"] = np.array([i["fov_loc"] for i in infos], dtype=np.int64)

    dump("prev", base.prev_record_buffer)
    dump("cur", base.record_buffer)
    # save_record_to_file of the finished episode
    cv2.VideoWriter.log.clear()
    pt = os.path.join(tmpdir, f"{name}.pt")
    n_rgb = len(base.prev_record_buffer["rgb"])
    base.save_record_to_file(pt)
    vw = cv2.VideoWriter.log[-1]
    saved = torch.load(pt, weights_only=False)            # written two lines above by this script
    rec["save_video_suffix"] = os.path.basename(vw["path"])[len(name):]
    rec["save_fourcc"], rec["save_fps"], rec["save_size"] = vw["fourcc"], vw["fps"], np.array(vw["size"])
    rec["save_frames"] = len(vw["frames"])
    assert len(vw["frames"]) == n_rgb and vw.get("released")
    rec["saved_keys"] = np.array(sorted(saved.keys()))
    rec["saved_rgb_is_path"] = isinstance(saved["rgb"], str) and saved["rgb"] == vw["path"]
    rec["saved_state"] = np.array(saved["state"])
    path = os.path.join(HERE, f"record_{name}.npz")
    np.savez_compressed(path, **rec)
    return path


def make_record(fov_env, gym):
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        return [_record_case(fov_env, gym, k, k, 900 + i, tmp) for i, k in enumerate(("base", "fixed", "flex", "per"))]


# ------------------------------------------------------------------ record buffers through the WHOLE Atari stack
def _record_atari_case(atari_env, kind, seed):
    """The reference's own AtariBaseEnv / AtariFixedFovealEnv / AtariFlexibleFovealEnv / AtariFixedFovealPeripheralEnv with
    record=True (atari_env.py:174-192, fov_env.py:34-37,51-102,152-154,161-163,218-220,253-256,265-267,352-354,370-373)
    over a ScriptedALE with real-size 210x160 screens: what the record buffers hold after two finished episodes and the
    start of a third - frames (render(): atari_env.py:165-169), full states, actions, rewards, dones, infos, fov_loc /
    fov_res - for the drop-in's device path to reproduce (tests/test_gpu_env.py::test_record_buffers_match_reference_run).
    For THIS case only cv2.resize is the oracle's restatement of OpenCV's 8-bit INTER_LINEAR (the control flow and the
    buffer contents are the reference's; that arithmetic stays "parity unpinned")."""
    import zlib
    import cv2
    from oracle import oracle as _O
    obs, fov, fs = (84, 84), (30, 30), 2
    ale = ScriptedALE(seed=seed, screen_hw=(210, 160), n_actions=4, start_lives=2, p_life=0.015, p_over=0.01)
    _STATE["next_ale"] = ale
    _STATE["antialias"] = True
    random.seed(seed)
    noops = []
    orig_rr, orig_resize = random.randrange, cv2.resize

    def rec_randrange(n):
        v = orig_rr(n)
        noops.append(v)
        return v

    def resize(img, dsize, interpolation=None):
        img = np.asarray(img)
        if img.ndim == 3 and img.shape[-1] == 1:
            return _O.cv_resize_linear_u8(img[..., 0], dsize)
        if img.ndim == 3:
            return np.stack([_O.cv_resize_linear_u8(np.ascontiguousarray(img[..., c]), dsize) for c in range(img.shape[-1])], -1)
        return _O.cv_resize_linear_u8(img, dsize)

    atari_env.random.randrange = rec_randrange
    cv2.resize = resize
    try:
        args = atari_env.AtariEnvArgs(game="scripted", seed=seed, obs_size=obs, frame_stack=fs, action_repeat=4, record=True,
                                      fov_size=fov, fov_init_loc=(3, 5), sensory_action_mode="absolute", resize_to_full=True,
                                      mask_out=False, peripheral_res=(20, 20))
        env = {"base": atari_env.AtariBaseEnv, "fixed": atari_env.AtariFixedFovealEnv, "flex": atari_env.AtariFlexibleFovealEnv,
               "per": atari_env.AtariFixedFovealPeripheralEnv}[kind](args)
        base = env
        while not hasattr(base, "prev_record_buffer"):
            base = base.env
        rng = np.random.default_rng(seed + 1)
        drive = [(1, 0, 0, 0, 0)]
        env.reset()
        episodes = 0
        for t in range(400):
            motor = int(rng.integers(0, 4))
            sens = rng.integers(-4, 70, size=2)
            typ = int(rng.integers(0, 2)) if kind == "flex" else 0
            if typ == 1:
                sens = rng.integers(8, 70, size=2)
            if kind == "base":
                _, _, d, _, _ = env.step(motor)
            else:
                act = {"motor_action": motor, "sensory_action": sens.astype(np.int64 if kind == "flex" else np.float64)}
                if kind == "flex":
                    act["sensory_action_type"] = np.array((typ,))
                _, _, d, _, _ = env.step(act)
            drive.append((0, motor, int(sens[0]), int(sens[1]), typ))
            if d:
                episodes += 1
                env.reset()
                drive.append((1, 0, 0, 0, 0))
                if episodes == 2:
                    break
        assert episodes == 2, "raise the step cap or the event rates"
        for _ in range(3):                       # the start of a third episode stays in record_buffer
            motor = int(rng.integers(0, 4))
            sens = rng.integers(0, 50, size=2)
            if kind == "base":
                _, _, d, _, _ = env.step(motor)
            else:
                act = {"motor_action": motor, "sensory_action": sens.astype(np.int64 if kind == "flex" else np.float64)}
                if kind == "flex":
                    act["sensory_action_type"] = np.array((0,))
                _, _, d, _, _ = env.step(act)
            drive.append((0, motor, int(sens[0]), int(sens[1]), 0))
            if d:
                break
    finally:
        atari_env.random.randrange = orig_rr
        cv2.resize = orig_resize
    rec = {"kind": kind, "seed": seed, "obs_size": np.array(obs), "fov_size": np.array(fov), "frame_stack": fs, "action_repeat": 4,
           "init_loc": np.array((3, 5)), "peripheral_res": np.array((20, 20)), "start_lives": 2, "p_life": 0.015, "p_over": 0.01,
           "noops": np.array(noops, dtype=np.int64), "drive": np.array(drive, dtype=np.int64)}

    def dump(tag, buf):
        rec[f"{tag}_keys"] = np.array(list(buf.keys()))                      # insertion order, as the reference builds it
        for k in ("rgb", "state", "action", "reward", "done", "truncated", "info", "return_reward", "fov_loc", "fov_res"):
            if k in buf:
                rec[f"{tag}_len_{k}"] = len(buf[k])
        rgb = np.stack(buf["rgb"])
        assert rgb.dtype == np.uint8 and rgb.shape[1:] == (256, 256, 3)
        rec[f"{tag}_rgb_crc"] = np.array([zlib.crc32(np.ascontiguousarray(f).tobytes()) for f in rgb], dtype=np.int64)
        rec[f"{tag}_rgb_first"] = rgb[0][::8, ::8].copy()                    # a thumbnail for diagnosis
        st = np.stack(buf["state"])
        assert st.dtype == np.float64
        u8 = np.rint(st * 255.0).astype(np.uint8)
        assert np.array_equal((u8.astype(np.float32) / np.float32(255.0)).astype(np.float64), st), "states must be k/255"
        rec[f"{tag}_state_crc"] = np.array([zlib.crc32(np.ascontiguousarray(f).tobytes()) for f in u8], dtype=np.int64)
        rec[f"{tag}_state_last_u8"] = u8[-1]
        for k in ("action", "reward", "done", "truncated", "return_reward"):
            rec[f"{tag}_{k}"] = np.array(buf[k])
        for k in ("fov_loc", "fov_res"):
            if k in buf:
                rec[f"{tag}_{k}"] = np.array(buf[k], dtype=np.int64)
        for k in ("fov_size", "peripheral_res"):
            if k in buf:
                rec[f"{tag}_{k}"] = np.array(buf[k])
        infos = buf["info"]
        for k in ("raw_reward", "reward", "ep_len"):
            rec[f"{tag}_info_{k}"] = np.array([i[k] for i in infos], dtype=np.float64)
        rec[f"{tag}_info_keys"] = np.array(list(infos[-1].keys())) if infos else np.array([])
        for k in ("fov_loc", "fov_res"):
            if infos and k in infos[-1]:
                rec[f"{tag}_info_{k}"] = np.array([i[k] for i in infos], dtype=np.int64)

    dump("prev", base.prev_record_buffer)
    dump("cur", base.record_buffer)
    path = os.path.join(HERE, f"record_atari_{kind}.npz")
    np.savez_compressed(path, **rec)
    return path


def make_record_atari(atari_env):
    return [_record_atari_case(atari_env, k, 950 + i) for i, k in enumerate(("base", "fixed", "flex", "per"))]


# ------------------------------------------------------------------ spaces / attributes
def make_spaces(fov_env, gym):
    """What each wrapper hands to gymnasium's space constructors and the attributes callers read
    (fov_env.py:110-147,236-251,358-367), for square / non-square / relative / degenerate geometries."""
    import json
    out = {}

    def space(sp):
        if isinstance(sp, dict):
            return {k: space(v) for k, v in sp.items()}
        if hasattr(sp, "n"):
            return {"type": "Discrete", "n": int(sp.n)}
        lo, hi = np.asarray(sp.low), np.asarray(sp.high)
        assert lo.min() == lo.max() and hi.min() == hi.max()              # uniform bounds everywhere in fov_env.py
        return {"type": "Box", "low": lo.min().item(), "high": hi.min().item(),
                "shape": None if sp.shape is None else list(sp.shape), "dtype": np.dtype(sp.dtype).name}

    for kind in ("fixed", "flex", "per"):
        for tag, obs, fov, mode, sas, rtf, mo in (
                ("abs84", (84, 84), (30, 30), "absolute", None, True, False),
                ("abs_nonsq", (36, 48), (10, 16), "absolute", None, False, False),
                ("abs_mask", (36, 48), (10, 16), "absolute", None, False, True),
                ("rel", (84, 84), (30, 30), "relative", (-10.0, 10.0), True, False),
                ("rel_degenerate", (84, 84), (30, 30), "relative", (3, 3), False, False)):

            class FakeBase(gym.Env):
                def __init__(self):
                    self.obs_size, self.frame_stack = tuple(obs), 4
                    self.action_space = gym.spaces.Discrete(6)

            args = _Args(fov_size=tuple(fov), fov_init_loc=(2.5, 3.5), sensory_action_mode=mode, sensory_action_space=sas,
                         resize_to_full=rtf, mask_out=mo, peripheral_res=(9, 7))
            base = fov_env.RecordWrapper(FakeBase(), args)
            env = {"fixed": fov_env.FixedFovealEnv, "flex": fov_env.FlexibleFovealEnv,
                   "per": fov_env.FixedFovealPeripheralEnv}[kind](base, args)
            e = {"action_space": space(env.action_space), "observation_space": space(env.observation_space),
                 "sensory_action_space": np.asarray(env.sensory_action_space).tolist(),
                 "fov_loc": np.asarray(env.fov_loc).tolist(), "fov_loc_dtype": str(np.asarray(env.fov_loc).dtype),
                 "fov_size": list(env.fov_size), "mask_out": bool(env.mask_out)}
            if kind == "flex":
                e["fov_res"] = np.asarray(env.fov_res).tolist()
            if kind == "per":
                e["peripheral_res"] = list(env.peripheral_res)
                e["resize_to_full"] = bool(env.resize_to_full)
            out[f"{kind}_{tag}"] = e
    path = os.path.join(HERE, "spaces.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    return [path]


# ------------------------------------------------------------------ DMC control-flow goldens
def _dmc_case(dmc_env, name, seed, obs=(12, 16), fs=3, ar=4, clip=False, steps=40, fixed_fov=False, episode_len=23):
    """The reference's DMCEnv (+ RecordWrapper [+ FixedFovealEnv]) over tests/fake_dmc.ScriptedDMC: pins action
    conversion, action repeat with early break, `reward or 0`, clipping, zero-fill + append, info keys."""
    import fake_dmc
    seen = {}

    def load(kw):
        seen.update(kw)
        return fake_dmc.ScriptedDMC(seed, episode_len=episode_len)

    _STATE["next_dmc"] = load
    _STATE["antialias"] = True
    kw = dict(frame_stack=fs, action_repeat=ar, clip_reward=clip)
    if fixed_fov:
        kw.update(fov_size=(4, 6), fov_init_loc=(1, 2), sensory_action_mode="absolute", resize_to_full=True)
    args = dmc_env.DMCEnvArgs(domain_name="scripted", task_name="t", seed=seed, obs_size=tuple(obs), **kw)
    env = dmc_env.DMCFixedFovealEnv(args) if fixed_fov else dmc_env.DMCBaseEnv(args)
    assert seen["task_kwargs"]["random"] == seed
    rng = np.random.default_rng(seed + 1)
    motor = rng.uniform(-1, 1, size=(steps, 2)).astype(np.float32)
    motor[3] = (-1.0, 1.0)
    sens = rng.uniform(-2, 12, size=(steps, 2))
    states, rewards, raws, dones, disc, internal, is_reset, ep_len, cum, fov_loc, cur = [], [], [], [], [], [], [], [], [], [], []

    def push(s, r, d, info, rs):
        states.append(np.asarray(s)); rewards.append(float(r)); dones.append(bool(d)); is_reset.append(rs)
        raws.append(float(info["raw_reward"]))
        disc.append(np.nan if info["discount"] is None else float(info["discount"]))
        internal.append(np.asarray(info["internal_state"]).copy())
        ep_len.append(info["ep_len"]); cum.append(float(info["reward"]))
        fov_loc.append(np.asarray(info.get("fov_loc", np.zeros(2))))
        cur.append(np.asarray(env.unwrapped.current_state).copy())

    s, info = env.reset()
    push(s, 0.0, False, info, True)
    for t in range(steps):
        if fixed_fov:
            s, r, d, tr, info = env.step({"motor_action": motor[t], "sensory_action": sens[t]})
        else:
            s, r, d, tr, info = env.step(motor[t])
        assert tr is False and sorted(k for k in info if k != "fov_loc") == ["discount", "ep_len", "internal_state", "raw_reward", "reward"]
        push(s, r, d, info, False)
        if d:
            s, info = env.reset()
            push(s, 0.0, False, info, True)
    # quirk: the zero frames of _reset_buffer are float64, the rendered frames float32 (dmc_env.py:183,195-197), so
    # np.stack yields float64 only while a zero frame is still in the deque
    state_is_f64 = np.array([x.dtype == np.float64 for x in states])
    assert all(x.dtype in (np.float32, np.float64) for x in states)
    states = np.stack([x.astype(np.float64) for x in states])
    base = env.unwrapped
    rec = dict(seed=seed, state_is_f64=state_is_f64, obs_size=np.array(obs), frame_stack=fs, action_repeat=ar, clip_reward=clip, fixed_fov=fixed_fov,
               episode_len=episode_len, motor=motor, sens=sens, rewards=np.array(rewards), raw_rewards=np.array(raws),
               dones=np.array(dones), discount=np.array(disc), internal_state=np.stack(internal), is_reset=np.array(is_reset),
               ep_len=np.array(ep_len, dtype=np.int64), cum_reward=np.array(cum), fov_loc=np.array(fov_loc, dtype=np.int64),
               current_state=np.stack(cur), true_low=base._true_action_space.low, true_high=base._true_action_space.high,
               obs_space_shape=np.array(base.observation_space.shape), state_space_shape=np.array(base.state_space.shape),
               reward_range=np.array(base.reward_range))
    if fixed_fov:
        rec["states_f64"] = states
    else:
        u8 = np.rint(states * 255.0).astype(np.uint8)
        assert np.array_equal((u8.astype(np.float32) / np.float32(255.0)).astype(np.float64), states)
        rec["states_u8"] = u8
    path = os.path.join(HERE, f"dmc_{name}.npz")
    np.savez_compressed(path, **rec)
    return path


def make_dmc(dmc_env):
    return [
        _dmc_case(dmc_env, "ar4_fs3", seed=21),
        _dmc_case(dmc_env, "ar4_clip", seed=22, clip=True),
        _dmc_case(dmc_env, "ar1_fs2", seed=23, ar=1, fs=2, steps=30, episode_len=11),
        _dmc_case(dmc_env, "ar3_fs4_break", seed=24, ar=3, fs=4, steps=30, episode_len=10),   # episode ends mid-repeat
        _dmc_case(dmc_env, "fixedfov", seed=25, steps=30, fixed_fov=True),
    ]


# ------------------------------------------------------------------ third-party arithmetic (only where cv2 exists)
def make_cv2():
    """Inputs and cv2's own outputs for the two OpenCV operations the path uses; written only on a machine that has
    cv2 (tests/test_oracle_resize.py::test_cv2_goldens_if_present checks the oracle against them when they exist)."""
    cv2 = _STATE.get("cv2_real")
    if cv2 is None:
        return []
    rng = np.random.default_rng(4242)
    rec = {"cv2_version": cv2.__version__}
    for i, ((h, w), (oh, ow)) in enumerate((((210, 160), (84, 84)), ((210, 160), (64, 64)), ((210, 160), (96, 96)))):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        rec[f"resize_in_{i}"] = img
        rec[f"resize_out_{i}"] = cv2.resize(img, (ow, oh), interpolation=cv2.INTER_LINEAR)
    rgb = rng.integers(0, 256, (84, 84, 3), dtype=np.uint8)
    rec["gray_in"] = rgb
    rec["gray_out"] = cv2.cvtColor(rgb, cv2.COLOR_BGR2GRAY)
    path = os.path.join(HERE, "cv2_arithmetic.npz")
    np.savez_compressed(path, **rec)
    return [path]


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f"reference checkout not found at {REF}; goldens are generated in the build container only")
    _install_standins()
    fov_env, atari_env, dmc_env = _load_reference()
    gym = sys.modules["gymnasium"]
    if sys.argv[1:] == ["record_atari"]:          # only the whole-stack record goldens (the other files stay byte-identical)
        paths = make_record_atari(atari_env)
    else:
        paths = (make_fovea(fov_env, gym) + make_atari(atari_env) + make_record(fov_env, gym) + make_record_atari(atari_env) +
                 make_dmc(dmc_env) + make_spaces(fov_env, gym) + make_cv2())
    total = 0
    for p in paths:
        sz = os.path.getsize(p)
        total += sz
        print(f"{os.path.relpath(p, REPO):60s} {sz/1024:9.1f} KiB")
    print(f"total {total/1e6:.2f} MB in {len(paths)} files")


if __name__ == "__main__":
    main()

"""CPU: the recording logic (active_gym/record.py: Recorder) replayed against buffers the reference's own
RecordWrapper + fovea wrappers produced (tests/golden/record_*.npz, made by tests/golden/make_golden.py from
fov_env.py:34-37,51-55,64-102,152-154,161-163,207,218-220,253-256,265-267,334-335,352-354,370-373).
The stub below makes exactly the calls the product's single-env classes make (fov_env.py of this package:
``rec.on_reset(full_state, info, fovea=self)`` / ``rec.on_step(full_state, motor, cum_reward, done, False, info,
return_reward, fovea=self)``); fov_loc / fov_res come from the oracle's wrappers (pinned by their own goldens)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _unit(u8):
    return (u8.astype(np.float32) / np.float32(255.0)).astype(np.float64)


class _Base:
    def __init__(self, g):
        self.g = g
        self.t = -1

    def render(self):
        return self.g["rgbs"][self.t]


class _Wrapper:
    def __init__(self, env):
        self.env = env
        self.record_buffer = None
        self.prev_record_buffer = None


class _Fovea:
    pass


def _replay(kind):
    from active_gym.record import Recorder
    g = np.load(os.path.join(GOLD, f"record_{kind}.npz"))
    base = _Base(g)
    w = _Wrapper(base)
    rec = Recorder(w)
    fov = None
    handle = None
    if kind != "base":
        kw = dict(obs_size=tuple(g["obs_size"]), fov_size=tuple(g["fov_size"]), fov_init_loc=tuple(g["init_loc"]),
                  sensory_action_mode="absolute", sensory_action_space=(-3.0, 3.0))
        if kind == "fixed":
            fov = O.FixedFovealOracle(resize_to_full=True, mask_out=False, **kw)
        elif kind == "flex":
            fov = O.FlexibleFovealOracle(resize_to_full=True, mask_out=False, **kw)
        else:
            fov = O.PeripheralOracle(peripheral_res=tuple(g["peripheral_res"]), **kw)
        handle = _Fovea()
        handle.fov_size = tuple(int(v) for v in g["fov_size"])
        if kind == "per":
            handle.peripheral_res = tuple(int(v) for v in g["peripheral_res"])
    done_at = set(g["done_at"].tolist())
    cum, ep_len = 0.0, 0
    for is_reset, motor, s0, s1, typ in g["drive"].tolist():
        base.t += 1
        state = _unit(g["states_u8"][base.t])
        if is_reset:
            cum, ep_len = 0.0, 0
            info = {"raw_reward": 0, "reward": cum, "ep_len": ep_len}
            if fov is not None:
                fov.reset(state)
        else:
            raw = float(g["base_rewards"][base.t])
            cum += raw
            ep_len += 1
            info = {"raw_reward": raw, "reward": cum, "ep_len": ep_len}
            if fov is not None:
                a = np.array((s0, s1))
                if kind == "flex":
                    fov.step(state, a, np.array((typ,)))
                else:
                    fov.step(state, a.astype(np.float64))
        if fov is not None:
            info["fov_loc"] = np.asarray(fov.fov_loc).copy()
            if kind == "flex":
                info["fov_res"] = np.asarray(fov.fov_res).copy()
        if is_reset:
            rec.on_reset(state, info, fovea=handle)
        else:
            rec.on_step(state, motor, cum, base.t in done_at, False, info, float(np.sign(raw)), fovea=handle)
    return g, w, rec


def _check(g, tag, buf):
    assert sorted(buf.keys()) == g[f"{tag}_keys"].tolist()
    for k in ("rgb", "state", "action", "reward", "done", "truncated", "info", "return_reward", "fov_loc", "fov_res"):
        if f"{tag}_len_{k}" in g.files:
            assert len(buf[k]) == int(g[f"{tag}_len_{k}"]), (tag, k)
    assert np.array_equal(np.stack(buf["rgb"]), g[f"{tag}_rgb"])
    st = np.stack(buf["state"])
    assert st.dtype == np.float64 and np.array_equal(st, g[f"{tag}_state"])
    for k in ("action", "reward", "done", "truncated", "return_reward"):
        assert np.array_equal(np.array(buf[k]), g[f"{tag}_{k}"]), (tag, k)
    for k in ("fov_loc", "fov_res", "fov_size", "peripheral_res"):
        if f"{tag}_{k}" in g.files:
            assert np.array_equal(np.array(buf[k], dtype=np.int64), g[f"{tag}_{k}"]), (tag, k)
    for k in ("raw_reward", "reward", "ep_len"):
        assert np.array_equal(np.array([i[k] for i in buf["info"]], dtype=np.float64), g[f"{tag}_info_{k}"])
    assert sorted(buf["info"][-1].keys()) == g[f"{tag}_info_keys"].tolist()
    if f"{tag}_info_fov_loc" in g.files:
        assert np.array_equal(np.array([i["fov_loc"] for i in buf["info"]], dtype=np.int64), g[f"{tag}_info_fov_loc"])


@pytest.mark.parametrize("kind", ["base", "fixed", "flex", "per"])
def test_record_buffers_match_reference(kind):
    g, w, rec = _replay(kind)
    _check(g, "prev", w.prev_record_buffer)
    _check(g, "cur", w.record_buffer)


@pytest.mark.parametrize("kind", ["base", "flex"])
def test_save_record_to_file_matches_reference(kind, tmp_path, monkeypatch):
    g, w, rec = _replay(kind)
    log = []

    class VideoWriter:
        def __init__(self, path, fourcc, fps, size):
            self.e = {"path": path, "fourcc": fourcc, "fps": fps, "size": tuple(size), "frames": 0, "released": False}
            log.append(self.e)

        def write(self, frame):
            self.e["frames"] += 1

        def release(self):
            self.e["released"] = True

    cv2 = types.ModuleType("cv2")
    cv2.VideoWriter = VideoWriter
    cv2.VideoWriter_fourcc = lambda *c: "".join(c)
    monkeypatch.setitem(sys.modules, "cv2", cv2)
    path = str(tmp_path / f"{kind}.pt")
    rec.save(path)
    e = log[-1]
    assert os.path.basename(e["path"]) == kind + str(g["save_video_suffix"])
    assert e["fourcc"] == str(g["save_fourcc"]) and e["fps"] == int(g["save_fps"]) and e["size"] == tuple(g["save_size"].tolist())
    assert e["frames"] == int(g["save_frames"]) and e["released"]
    saved = torch.load(path, weights_only=False)        # written by the line above
    assert sorted(saved.keys()) == g["saved_keys"].tolist()
    assert saved["rgb"] == e["path"] and bool(g["saved_rgb_is_path"])
    assert np.array_equal(np.array(saved["state"]), g["saved_state"])


def test_save_without_cv2_writes_frames_as_npy(tmp_path, monkeypatch):
    g, w, rec = _replay("fixed")
    monkeypatch.setitem(sys.modules, "cv2", None)          # import cv2 -> ImportError
    path = str(tmp_path / "ep.pt")
    rec.save(path)
    saved = torch.load(path, weights_only=False)
    assert saved["rgb"].endswith("ep.rgb.npy") and np.load(saved["rgb"]).shape[0] == int(g["save_frames"])

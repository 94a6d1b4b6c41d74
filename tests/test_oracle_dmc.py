"""CPU: the oracle's DMC restatement (oracle/oracle.py: DMCEnvOracle, cv_bgr2gray_u8) against goldens produced by
the reference's own dmc_env.py (tests/golden/dmc_*.npz, tests/golden/make_golden.py) over the scripted stand-in
tests/fake_dmc.py.  The goldens pin the CONTROL FLOW (action conversion, action repeat with early break,
`reward or 0`, clipping, zero-fill + append, info) - OpenCV's BGR2GRAY arithmetic itself is "parity unpinned"
(cv2 absent; only the classic known answers below)."""
import glob
import os

import numpy as np
import pytest

from fake_dmc import ScriptedDMC
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLD, "dmc_*.npz")))


def make_chain(g):
    env = O.DMCEnvOracle(ScriptedDMC(int(g["seed"]), episode_len=int(g["episode_len"])), obs_size=tuple(g["obs_size"]),
                         frame_stack=int(g["frame_stack"]), action_repeat=int(g["action_repeat"]),
                         clip_reward=bool(g["clip_reward"]))
    rec = O.RecordOracle(env)
    fov = None
    if bool(g["fixed_fov"]):
        fov = O.FixedFovealOracle(obs_size=tuple(g["obs_size"]), fov_size=(4, 6), fov_init_loc=(1, 2),
                                  sensory_action_mode="absolute", resize_to_full=True)
    return env, rec, fov


def test_cases_present():
    assert len(CASES) == 5


@pytest.mark.parametrize("name", CASES)
def test_dmc_oracle_matches_reference_golden(name):
    g = np.load(os.path.join(GOLD, f"dmc_{name}.npz"))
    env, rec, fov = make_chain(g)
    assert np.array_equal(env.true_low, g["true_low"]) and np.array_equal(env.true_high, g["true_high"])
    k = 0

    def check(s, r, d, info):
        nonlocal k
        if fov is None:
            want = (g["states_u8"][k].astype(np.float32) / np.float32(255)).astype(np.float64)
            assert np.array_equal(s, want), k
            # dtype quirk of the reference: float64 only while a zero frame of _reset_buffer is still stacked
            assert (s.dtype == np.float64) == bool(g["state_is_f64"][k]) and s.dtype in (np.float32, np.float64)
        else:
            # once the DMC state is float32 (see the dtype quirk below) torchvision resizes in float32; the oracle
            # always resizes in float64: agreement to float32 rounding there, to 1e-12 while the state is float64
            tol = 1e-12 if bool(g["state_is_f64"][k]) else 5e-7
            assert np.abs(s - g["states_f64"][k]).max() < tol, k
            assert np.array_equal(fov.fov_loc, g["fov_loc"][k])
        assert float(r) == g["rewards"][k] and bool(d) == bool(g["dones"][k])
        assert float(info["raw_reward"]) == g["raw_rewards"][k]
        disc = np.nan if info["discount"] is None else info["discount"]
        assert (np.isnan(disc) and np.isnan(g["discount"][k])) or disc == g["discount"][k]
        assert np.array_equal(info["internal_state"], g["internal_state"][k])
        assert info["ep_len"] == g["ep_len"][k] and float(info["reward"]) == g["cum_reward"][k]
        k += 1

    s, info = rec.reset()
    check(fov.reset(s) if fov else s, 0.0, False, info)
    t = 0
    while k < len(g["rewards"]):
        assert not g["is_reset"][k]
        s, r, d, tr, info = rec.step(g["motor"][t])
        check(fov.step(s, g["sens"][t]) if fov else s, r, d, info)
        t += 1
        if d:
            assert g["is_reset"][k]
            s, info = rec.reset()
            check(fov.reset(s) if fov else s, 0.0, False, info)
    assert t == len(g["motor"]) and g["dones"].sum() >= 1


def test_bgr2gray_known_answers_and_identities():
    # the classic OpenCV answers for pure channels (blue weight on channel 0): 29 / 150 / 76
    px = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0]], np.uint8)
    for mode in ("cv15", "cv14"):
        assert O.cv_bgr2gray_u8(px, mode).tolist() == [29, 150, 76, 255, 0]
        v = np.arange(256, dtype=np.uint8)
        assert np.array_equal(O.cv_bgr2gray_u8(np.stack([v, v, v], -1), mode), v)     # weights sum to 1 << shift
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    a, b = O.cv_bgr2gray_u8(img, "cv15").astype(int), O.cv_bgr2gray_u8(img, "cv14").astype(int)
    assert np.abs(a - b).max() <= 1 and (a != b).any()                                 # the generations differ by <= 1 LSB
    ref = img[..., 0] * 0.114 + img[..., 1] * 0.587 + img[..., 2] * 0.299
    assert np.abs(a - ref).max() <= 0.51

"""CPU: pin the oracle (oracle/oracle.py) against vectors produced by the
reference's own fov_env.py / atari_env.py (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from fake_ale import ScriptedALE
from golden_util import (atari_case_names, fovea_case_names, golden_tol, load_atari,
                         load_fovea, unit64)
from oracle import oracle as O


def make_fovea_oracle(c):
    kw = dict(obs_size=tuple(c["obs_size"]), fov_size=tuple(c["fov_size"]),
              fov_init_loc=tuple(c["init_loc"]), sensory_action_mode=c["mode"],
              sensory_action_space=tuple(c["sas"]), antialias=c["antialias"])
    if c["kind"] == "fixed":
        return O.FixedFovealOracle(resize_to_full=c["resize_to_full"], mask_out=c["mask_out"], **kw)
    if c["kind"] == "flex":
        return O.FlexibleFovealOracle(resize_to_full=c["resize_to_full"], mask_out=c["mask_out"], **kw)
    return O.PeripheralOracle(peripheral_res=tuple(c["peripheral_res"]), **kw)


@pytest.mark.parametrize("name", fovea_case_names())
def test_fovea_oracle_matches_reference(name):
    c = load_fovea(name)
    orc = make_fovea_oracle(c)
    tol = golden_tol(c)
    out = orc.reset(unit64(c["states_u8"][0]))
    assert out.shape == c["outs"][0].shape
    np.testing.assert_allclose(out, c["outs"][0], rtol=0, atol=tol)
    assert np.array_equal(orc.fov_loc, c["fov_loc"][0])
    for t in range(c["steps"]):
        fs = unit64(c["states_u8"][t + 1])
        if c["kind"] == "flex":
            a = c["actions"][t]
            if c["action_types"][t] == 1:
                a = a.astype(np.int64)
            out = orc.step(fs, a, np.array((c["action_types"][t],)))
            assert np.array_equal(np.asarray(orc.fov_res), c["fov_res"][t + 1])
        else:
            out = orc.step(fs, c["actions"][t])
        assert np.array_equal(orc.fov_loc, c["fov_loc"][t + 1]), (t, orc.fov_loc, c["fov_loc"][t + 1])
        assert out.shape == c["outs"][t + 1].shape
        np.testing.assert_allclose(out, c["outs"][t + 1], rtol=0, atol=tol)


def _identity_resize(g):
    return np.asarray(g)


@pytest.mark.parametrize("name", atari_case_names())
def test_atari_control_flow_matches_reference(name):
    c = load_atari(name)
    obs = tuple(c["obs_size"])
    ale = ScriptedALE(seed=c["seed"], screen_hw=obs, n_actions=c["n_actions"])
    noops = list(c["noops"])
    env = O.AtariEnvOracle(ale, ale.getMinimalActionSet(), obs_size=obs, frame_stack=c["frame_stack"],
                           action_repeat=c["action_repeat"], clip_reward=c["clip_reward"],
                           noop_fn=lambda: int(noops.pop(0)), resize_fn=_identity_resize)
    if not c["training"]:
        env.eval()
    rec = O.RecordOracle(env)
    fov = None
    if c["fixed_fov"]:
        fov = O.FixedFovealOracle(obs, (6, 6), (2, 3), "absolute", resize_to_full=True, antialias=True)
    want = c["states_f64"] if c["fixed_fov"] else unit64(c["states_u8"])

    def check(i, s, r, d, info, loc=None):
        np.testing.assert_allclose(s, want[i], rtol=0, atol=1e-12 if c["fixed_fov"] else 0)
        assert float(r) == c["rewards"][i] and bool(d) == bool(c["dones"][i])
        assert info["ep_len"] == c["ep_len"][i] and float(info["reward"]) == c["cum_reward"][i]
        if loc is not None:
            assert np.array_equal(loc, c["fov_loc"][i])

    i = 0
    s, info = rec.reset()
    if fov:
        s = fov.reset(s)
    check(i, s, 0.0, False, info, fov.fov_loc if fov else None)
    assert c["is_reset"][i]
    for t in range(len(c["motor"])):
        i += 1
        s, r, d, tr, info = rec.step(int(c["motor"][t]))
        if fov:
            s = fov.step(s, c["sens"][t])
        assert not c["is_reset"][i]
        check(i, s, r, d, info, fov.fov_loc if fov else None)
        if d:
            i += 1
            s, info = rec.reset()
            if fov:
                s = fov.reset(s)
            assert c["is_reset"][i]
            check(i, s, 0.0, False, info, fov.fov_loc if fov else None)
    assert i + 1 == len(c["dones"]) and not noops

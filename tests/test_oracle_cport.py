"""CPU: the C restatement (oracle/cport.c) agrees with the NumPy oracle: u8 stages bit-exact,
float64 resize to rounding."""
import importlib.util
import os

import numpy as np

from oracle import oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cport():
    spec = importlib.util.spec_from_file_location("agx_oracle_build", os.path.join(REPO, "oracle", "build_oracle.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.build()
    from oracle import cport
    return cport


def test_c_get_state_matches_numpy_oracle():
    cp = _cport()
    rng = np.random.default_rng(0)
    for obs in ((84, 84), (64, 64), (96, 96)):
        rgb = rng.integers(0, 256, (210, 160, 3), dtype=np.uint8)
        assert np.array_equal(cp.get_state(rgb, obs), O.get_state_u8(rgb, obs))


def test_c_step_matches_numpy_oracle():
    cp = _cport()
    rng = np.random.default_rng(1)
    n = 3
    eb = cp.EnvBatch(n)
    ring = O.RingOracle(n, 4, (84, 84))
    fov = [O.FixedFovealOracle((84, 84), (30, 30), (0, 0), "absolute", resize_to_full=True) for _ in range(n)]
    for step in range(4):
        frames = rng.integers(0, 256, (n, 2, 210, 160, 3), dtype=np.uint8)
        act = rng.uniform(-5, 60, (n, 2))
        act[0] = np.floor(act[0]) + 0.5
        nvalid = np.array([2, 1, step % 3])
        out, loc = eb.step_fixed(frames, act, nvalid)
        ring.ingest(frames, nvalid)
        assert np.array_equal(eb.ring, ring.stack_u8())
        full = ring.full_state()
        for i in range(n):
            want = fov[i].step(full[i], act[i])
            assert np.array_equal(loc[i], fov[i].fov_loc)
            np.testing.assert_allclose(out[i], want, rtol=0, atol=1e-14)


def test_c_ingest_with_command_bytes_matches_numpy_oracle():
    """agxo_ingest / agxo_fovea_fixed (the split form tests/test_gpu_fullsize.py steps 1024 envs through): nvalid, CLEAR and
    SKIP as agx_ingest's command byte encodes them, against RingOracle."""
    cp = _cport()
    rng = np.random.default_rng(2)
    n = 6
    eb = cp.EnvBatch(n)
    ring = O.RingOracle(n, 4, (84, 84))
    fov = [O.FixedFovealOracle((84, 84), (30, 30), (0, 0), "absolute", resize_to_full=True) for _ in range(n)]
    for step in range(5):
        frames = rng.integers(0, 256, (n, 2, 210, 160, 3), dtype=np.uint8)
        nvalid = rng.integers(0, 3, n)
        clear = (rng.random(n) < 0.3).astype(np.uint8)
        skip = (rng.random(n) < 0.3).astype(np.uint8) if step else np.zeros(n, np.uint8)
        nvalid[clear == 1] = 1
        cmd = (nvalid | clear * 4 | skip * 8).astype(np.uint8)
        eb.ingest(frames, cmd)
        ring.ingest(frames, nvalid, clear=clear, skip=skip)
        assert np.array_equal(eb.ring, ring.stack_u8()), step
        act = rng.uniform(-5, 60, (n, 2))
        out, loc = eb.fovea_fixed(act)
        full = ring.full_state()
        for i in range(n):
            want = fov[i].step(full[i], act[i])
            assert np.array_equal(loc[i], fov[i].fov_loc)
            np.testing.assert_allclose(out[i], want, rtol=0, atol=1e-14)

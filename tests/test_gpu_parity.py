"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU
oracle on the same seeded inputs, against the committed golden vectors, and —
at the benchmark's full size — through size-independent properties.

Bars: integer / byte / index work bit-exact; float resize within 1e-5
(BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from golden_util import fovea_case_names, load_fovea, unit64
from oracle import oracle as O

pytestmark = pytest.mark.gpu
FLOAT_TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _pipe(**kw):
    from active_gym import ObsPipeline
    return ObsPipeline(**kw)


def _t(x, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


def test_native_library_is_loaded():
    from active_gym import _native as nat
    lib = nat.lib()
    assert lib.agx_abi_version() == nat.ABI_VERSION
    maps = open("/proc/self/maps").read()
    assert "libagx.so" in maps
    # ... and it is built from the sources in this tree (a stale .so travelling to the GPU box would make every test here
    # a test of something else): agx_build_info() carries the hash build.py takes over the kernel and ABI sources
    import importlib.util
    spec = importlib.util.spec_from_file_location("agx_build", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                            "active-gym_amd", "build.py"))
    bld = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bld)
    if not os.environ.get("AGX_LIB"):
        assert nat.build_info().endswith(" src " + bld.source_hash()), (nat.build_info(), bld.source_hash())


# ---------------------------------------------------------------- K0 / unit conversion
def test_unit_division_all_256_values_bit_exact(dev):
    p = _pipe(num_envs=1, kind="base", obs_size=(16, 16), frame_stack=1)
    st = np.arange(256, dtype=np.uint8).reshape(1, 1, 16, 16)
    p.set_stack_u8(_t(st, dev))
    got = p.observe_full().cpu().numpy()
    want = st.astype(np.float32) / np.float32(255.0)
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ---------------------------------------------------------------- K1
def _tie_pixels():
    """RGB triples on exact .5 luminance ties (2989r+5870g+1140b = 5000 mod 10000)."""
    out = []
    for r in range(0, 256, 5):
        for g in range(256):
            for b in range(0, 256, 2):
                if (2989 * r + 5870 * g + 1140 * b) % 10000 == 5000:
                    out.append((r, g, b))
    return np.array(out, dtype=np.uint8)


def test_ingest_matches_oracle_sequence(dev):
    N, fs = 7, 4
    rng = np.random.default_rng(42)
    p = _pipe(num_envs=N, kind="base", obs_size=(84, 84), frame_stack=fs)
    ring = O.RingOracle(N, fs, (84, 84))
    ties = _tie_pixels()
    assert len(ties) > 50
    for step in range(7):
        frames = rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8)
        # sprinkle luminance ties and extremes
        idx = rng.integers(0, len(ties), size=(N, 2, 210, 160))
        tie_mask = rng.random((N, 2, 210, 160)) < 0.02
        frames[tie_mask] = ties[idx[tie_mask]]
        frames[0, 0, :20] = 255
        frames[0, 1, :20] = 0
        nvalid = rng.integers(0, 3, N)
        if step == 0:
            nvalid[:] = 2
        clear = (rng.random(N) < 0.2).astype(np.uint8)
        skip = (rng.random(N) < 0.2).astype(np.uint8)
        if step == 0:
            skip[:] = 0
        nvalid[clear == 1] = 1            # a reset appends one _get_state() frame
        cmd = (nvalid | (clear * 0x04) | (skip * 0x08)).astype(np.uint8)
        p.ingest(_t(frames, dev), _t(cmd, dev))
        ring.ingest(frames, nvalid, clear=clear, skip=skip)
        got = p.stack_u8().cpu().numpy()
        assert np.array_equal(got, ring.stack_u8()), f"step {step}"
    full = p.observe_full().cpu().numpy()
    assert np.array_equal(full, O.u8_to_unit(ring.stack_u8()))


def test_ingest_luminance_ties_exhaustive_rows(dev):
    """Every tie triple, laid out as constant 2x2-source blocks, must come out as ALE's double rounding."""
    ties = _tie_pixels()
    N = 1
    frames = np.zeros((N, 2, 210, 160, 3), np.uint8)
    flat = frames[0, 0].reshape(-1, 3)
    reps = np.repeat(ties, 1, axis=0)
    flat[:] = reps[np.arange(flat.shape[0]) % len(reps)]
    frames[0, 1] = frames[0, 0]
    p = _pipe(num_envs=N, kind="base", obs_size=(84, 84), frame_stack=2)
    p.ingest(_t(frames, dev), _t(np.array([2], np.uint8), dev))
    got = p.stack_u8().cpu().numpy()[0, -1]
    assert np.array_equal(got, O.get_state_u8(frames[0, 0], (84, 84)))


@pytest.mark.parametrize("obs", [(84, 84), (64, 64), (96, 96), (40, 40)])
def test_ingest_other_square_sizes(dev, obs):
    N = 3
    rng = np.random.default_rng(obs[0])
    frames = rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8)
    p = _pipe(num_envs=N, kind="base", obs_size=obs, frame_stack=2)
    ring = O.RingOracle(N, 2, obs)
    cmd = np.array([2, 1, 2], np.uint8)
    p.ingest(_t(frames, dev), _t(cmd, dev))
    ring.ingest(frames, cmd & 3)
    assert np.array_equal(p.stack_u8().cpu().numpy(), ring.stack_u8())


@pytest.mark.parametrize("gray", [False, True])
@pytest.mark.parametrize("obs", [(84, 84), (64, 64), (96, 96), (40, 40), (128, 128), (48, 48)])
def test_ingest_compact_equals_whole_screens(dev, obs, gray):
    """agx_ingest_compact / agx_ingest_gray_raw_compact: only the screen rows cv2.resize reads (agx_source_rows), packed - the ring
    must equal the whole-screen path's bit for bit, for the band12 form (84: packed rows 2 dy, 2 dy + 1) and for the general
    kernel with its packed row table (every other size; 128 reads overlapping row pairs), with random commands."""
    N, fs = 9, 3
    rng = np.random.default_rng(obs[0] + gray)
    a = _pipe(num_envs=N, kind="base", obs_size=obs, frame_stack=fs)
    b = _pipe(num_envs=N, kind="base", obs_size=obs, frame_stack=fs)
    rows = b.source_rows()
    ty = O.cv_tables_y(210, obs[0])
    want_rows = np.unique(np.concatenate([np.asarray(ty[0]), np.asarray(ty[1])]))
    assert np.array_equal(rows, want_rows), "agx_source_rows vs the oracle's cv2 row table"
    if obs == (84, 84):
        assert len(rows) == 168
    for step in range(4):
        frames = rng.integers(0, 256, (N, 2, 210, 160) + (() if gray else (3,)), dtype=np.uint8)
        nvalid = rng.integers(0, 3, N)
        clear = (rng.random(N) < 0.2).astype(np.uint8)
        skip = (rng.random(N) < 0.2).astype(np.uint8) if step else np.zeros(N, np.uint8)
        nvalid[clear == 1] = 1
        cmd = _t((nvalid | clear * 4 | skip * 8).astype(np.uint8), dev)
        comp = np.ascontiguousarray(frames[:, :, rows])
        if gray:
            a.ingest_gray_raw(_t(frames, dev), cmd)
            b.ingest_gray_raw_compact(_t(comp, dev), cmd)
        else:
            a.ingest(_t(frames, dev), cmd)
            b.ingest_compact(_t(comp, dev), cmd)
        assert torch.equal(a.stack_u8(), b.stack_u8()), step
    assert a.stack_u8().any()


def test_ingest_rejects_non_square_and_bad_shapes(dev):
    from active_gym import _native as nat
    p = _pipe(num_envs=2, kind="base", obs_size=(36, 48), frame_stack=2)
    frames = torch.zeros((2, 2, 210, 160, 3), dtype=torch.uint8, device=dev)
    cmd = torch.full((2,), 2, dtype=torch.uint8, device=dev)
    with pytest.raises(nat.AgxError, match="not square"):
        p.ingest(frames, cmd)
    with pytest.raises(ValueError):
        p.ingest(frames[:1], cmd)
    with pytest.raises(TypeError):
        p.ingest(frames.float(), cmd)
    with pytest.raises(ValueError):
        _pipe(num_envs=2, kind="fixed", obs_size=(84, 84), fov_size=(84, 30))


def test_ingest_gray_matches_oracle(dev):
    N, fs, obs = 5, 3, (36, 48)
    rng = np.random.default_rng(7)
    p = _pipe(num_envs=N, kind="base", obs_size=obs, frame_stack=fs)
    dq = [[np.zeros(obs, np.uint8)] * fs for _ in range(N)]
    for step in range(5):
        small = rng.integers(0, 256, (N, 2) + obs, dtype=np.uint8)
        nvalid = rng.integers(0, 3, N)
        clear = (rng.random(N) < 0.25).astype(np.uint8)
        skip = (rng.random(N) < 0.25).astype(np.uint8)
        cmd = (nvalid | clear * 4 | skip * 8).astype(np.uint8)
        p.ingest_gray(_t(small, dev), _t(cmd, dev))
        for i in range(N):
            if skip[i]:
                continue
            if clear[i]:
                dq[i] = [np.zeros(obs, np.uint8)] * fs
            o = np.zeros(obs, np.uint8)
            for f in range(nvalid[i]):
                o = np.maximum(o, small[i, f])
            dq[i] = dq[i][1:] + [o]
        want = np.stack([np.stack(d) for d in dq])
        assert np.array_equal(p.stack_u8().cpu().numpy(), want)


# ---------------------------------------------------------------- golden replay (K2, K3, K4)
def _pipe_for_case(c, n=1):
    kind = {"fixed": "fixed", "flex": "flexible", "per": "peripheral"}[c["kind"]]
    return _pipe(num_envs=n, kind=kind, obs_size=tuple(int(v) for v in c["obs_size"]),
                 frame_stack=c["frame_stack"], fov_size=tuple(int(v) for v in c["fov_size"]),
                 fov_init_loc=tuple(float(v) for v in c["init_loc"]), sensory_action_mode=c["mode"],
                 sensory_action_space=tuple(float(v) for v in c["sas"]), resize_to_full=c["resize_to_full"],
                 mask_out=c["mask_out"], peripheral_res=tuple(int(v) for v in c["peripheral_res"]),
                 antialias=c["antialias"])


def _compare(c, got, want, res=None):
    """got: f32 [fs,h,w] from the device; want: golden (f32/f64, maybe ragged)."""
    exact = c["kind"] != "per" and (c["mask_out"] or not c["resize_to_full"])
    if c["kind"] == "flex" and not c["mask_out"] and not c["resize_to_full"]:
        rh, rw = int(res[0]), int(res[1])
        assert want.shape[-2:] == (rh, rw)
        pad = got.copy()
        pad[:, :rh, :rw] = 0
        assert not pad.any(), "padding of the ragged raw window must be zero"
        got = got[:, :rh, :rw]
    assert got.shape == want.shape, (got.shape, want.shape)
    np.testing.assert_allclose(got.astype(np.float64), want.astype(np.float64), rtol=0, atol=FLOAT_TOL)
    return exact


@pytest.mark.parametrize("name", fovea_case_names())
def test_fovea_golden_replay(dev, name):
    c = load_fovea(name)
    p = _pipe_for_case(c)
    flex = c["kind"] == "flex"
    states = c["states_u8"]
    p.set_stack_u8(_t(states[0][None], dev))
    p.fovea_reset()
    r = p.fovea()
    res0 = r[2].cpu().numpy()[0] if flex else None
    _compare(c, r[0].cpu().numpy()[0], c["outs"][0], res0)
    assert np.array_equal(r[1].cpu().numpy()[0], c["fov_loc"][0])
    worst = 0.0
    for t in range(c["steps"]):
        p.set_stack_u8(_t(states[t + 1][None], dev))
        a = _t(c["actions"][t][None], dev)                     # float64, as the reference received it
        if flex:
            at = _t(np.array([c["action_types"][t]], np.int32), dev)
            obs, loc, res = p.fovea(a, action_type=at)
            assert np.array_equal(res.cpu().numpy()[0], c["fov_res"][t + 1]), t
            rr = res.cpu().numpy()[0]
        else:
            obs, loc = p.fovea(a)
            rr = None
        assert np.array_equal(loc.cpu().numpy()[0], c["fov_loc"][t + 1]), (t, loc, c["fov_loc"][t + 1])
        g = obs.cpu().numpy()[0]
        w = c["outs"][t + 1]
        exact = _compare(c, g, w, rr)
        if exact and not (flex and int(rr[0]) > int(c["fov_size"][0])):
            gg = g[:, :w.shape[-2], :w.shape[-1]] if g.shape != w.shape else g
            assert np.array_equal(gg, w.astype(np.float32)), "crop / paste must be bit-exact"
        worst = max(worst, float(np.abs(g[..., :w.shape[-2], :w.shape[-1]].astype(np.float64) - w).max()))
    assert worst <= FLOAT_TOL


# ---------------------------------------------------------------- batched vs oracle
@pytest.mark.parametrize("mode", ["absolute", "relative"])
@pytest.mark.parametrize("out", ["resize", "mask", "raw"])
def test_fixed_batched_vs_oracle(dev, mode, out):
    N, fs = 33, 4
    rng = np.random.default_rng(hash((mode, out)) & 0xFFFF)
    kw = dict(obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(27, 27), sensory_action_mode=mode,
              sensory_action_space=(-10.0, 10.0), resize_to_full=(out == "resize"), mask_out=(out == "mask"))
    p = _pipe(num_envs=N, kind="fixed", frame_stack=fs, **kw)
    orcs = [O.FixedFovealOracle(**kw) for _ in range(N)]
    for step in range(4):
        st = rng.integers(0, 256, (N, fs, 84, 84), dtype=np.uint8)
        p.set_stack_u8(_t(st, dev))
        if mode == "absolute":
            a = rng.uniform(-5, 60, (N, 2)).astype(np.float32)
            a[::5] = np.floor(a[::5]) + 0.5
        else:
            a = rng.uniform(-14, 14, (N, 2)).astype(np.float32)
            a[::4] = np.floor(a[::4]) + 0.5
        a[1] = np.nan if step == 2 else a[1]
        obs, loc = p.fovea(_t(a, dev))
        obs, loc = obs.cpu().numpy(), loc.cpu().numpy()
        for i in range(N):
            ai = a[i]
            if np.isnan(ai).any():
                continue                       # NaN is normalised by the ABI; the reference is undefined
            want = orcs[i].step(unit64(st[i]), ai)
            assert np.array_equal(loc[i], orcs[i].fov_loc), (step, i, ai, loc[i], orcs[i].fov_loc)
            if out == "resize":
                np.testing.assert_allclose(obs[i], want, rtol=0, atol=FLOAT_TOL)
            else:
                assert np.array_equal(obs[i], want.astype(np.float32))
        if step == 2:                          # resync the NaN env
            orcs[1].fov_loc = loc[1].astype(np.int64)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.int32, torch.int64])
def test_action_dtypes_and_half_to_even(dev, dtype):
    N = 8
    p = _pipe(num_envs=N, kind="fixed", obs_size=(84, 84), fov_size=(30, 30), frame_stack=1, resize_to_full=False)
    vals = np.array([[12.5, 13.5], [0.5, 1.5], [-3, 60.7], [53.5, 54.5], [54.49, 2.5], [7, 9], [100, -100], [22.5, 23.5]])
    if dtype in (torch.int32, torch.int64):
        vals = np.rint(vals)
    a = torch.tensor(vals, dtype=dtype, device=dev)
    _, loc = p.fovea(a)
    want = np.rint(np.clip(a.cpu().numpy(), 0, 54)).astype(int)
    assert np.array_equal(loc.cpu().numpy(), want)


@pytest.mark.parametrize("kind", ["fixed", "flexible"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.int32, torch.float64])
def test_action_tensor_at_an_element_aligned_address(dev, kind, dtype):
    """The env state reaches K2 / K4 through scalar loads (s_load_dwordx2): an action tensor that starts one element into its
    storage - aligned to its element size only, not to the 8 / 16 bytes of an env's pair - must give what an aligned copy gives."""
    N = 37
    rng = np.random.default_rng(5)
    kw = dict(num_envs=N, kind=kind, obs_size=(84, 84), fov_size=(30, 30), frame_stack=2, resize_to_full=True)
    a_, b_ = _pipe(**kw), _pipe(**kw)
    st = _t(rng.integers(0, 256, (N, 2, 84, 84), dtype=np.uint8), dev)
    a_.set_stack_u8(st); b_.set_stack_u8(st)
    vals = rng.uniform(-5, 70, (N, 2))
    flat = torch.zeros(2 * N + 1, dtype=dtype, device=dev)
    flat[1:] = torch.tensor(vals.reshape(-1), dtype=dtype, device=dev)
    odd = flat[1:].view(N, 2)                                   # contiguous, data_ptr = storage + one element
    assert odd.is_contiguous() and odd.data_ptr() % (2 * odd.element_size()) != 0
    even = odd.clone()
    extra = {}
    if kind == "flexible":
        extra = dict(action_type=_t(rng.integers(0, 2, N).astype(np.int32), dev))
    ra = a_.fovea(odd, **extra)
    rb = b_.fovea(even, **extra)
    for x, y in zip(ra, rb):
        assert torch.equal(x, y)
    a_.close(); b_.close()


def test_mask_leaves_envs_untouched(dev):
    N = 6
    rng = np.random.default_rng(3)
    p = _pipe(num_envs=N, kind="fixed", obs_size=(84, 84), fov_size=(30, 30), frame_stack=2, resize_to_full=True)
    p.set_stack_u8(_t(rng.integers(0, 256, (N, 2, 84, 84), dtype=np.uint8), dev))
    a = _t(rng.uniform(0, 54, (N, 2)).astype(np.float32), dev)
    obs, loc = p.fovea(a)
    obs0, loc0 = obs.clone(), loc.clone()
    mask = _t(np.array([1, 0, 1, 0, 0, 1], np.uint8), dev)
    a2 = _t(rng.uniform(0, 54, (N, 2)).astype(np.float32), dev)
    sentinel = torch.full_like(obs, -7.0)
    obs2, loc2 = p.fovea(a2, mask=mask, out=sentinel, loc_out=loc.clone())
    m = mask.bool().cpu().numpy()
    assert (obs2[~mask.bool()] == -7.0).all() and (obs2[mask.bool()] != -7.0).any()
    st_loc, _ = p.fov_state()
    assert np.array_equal(st_loc.cpu().numpy()[~m], loc0.cpu().numpy()[~m])
    assert np.array_equal(loc2.cpu().numpy()[~m], loc0.cpu().numpy()[~m])
    # reset only env 0
    p.fovea_reset(_t(np.array([1, 0, 0, 0, 0, 0], np.uint8), dev))
    st_loc2, _ = p.fov_state()
    assert st_loc2[0].tolist() == [0, 0] and np.array_equal(st_loc2[1:].cpu().numpy(), st_loc.cpu().numpy()[1:])


@pytest.mark.parametrize("aa", [False, True])
def test_peripheral_batched_vs_oracle(dev, aa):
    N, fs = 9, 4
    rng = np.random.default_rng(11 + aa)
    kw = dict(obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute",
              peripheral_res=(20, 20), antialias=aa)
    p = _pipe(num_envs=N, kind="peripheral", frame_stack=fs, **kw)
    orcs = [O.PeripheralOracle(**kw) for _ in range(N)]
    for step in range(3):
        st = rng.integers(0, 256, (N, fs, 84, 84), dtype=np.uint8)
        p.set_stack_u8(_t(st, dev))
        a = rng.uniform(-5, 60, (N, 2))
        obs, loc = p.fovea(_t(a, dev))
        obs, loc = obs.cpu().numpy(), loc.cpu().numpy()
        for i in range(N):
            want = orcs[i].step(unit64(st[i]), a[i])
            assert np.array_equal(loc[i], orcs[i].fov_loc)
            np.testing.assert_allclose(obs[i], want, rtol=0, atol=FLOAT_TOL)
            r, c = loc[i]
            assert np.array_equal(obs[i][:, r:r + 30, c:c + 30], want[:, r:r + 30, c:c + 30].astype(np.float32))


@pytest.mark.parametrize("aa", [False, True])
@pytest.mark.parametrize("out", ["resize", "mask", "raw"])
def test_flexible_batched_vs_oracle(dev, aa, out):
    N, fs = 17, 2
    rng = np.random.default_rng(23 + aa + len(out))
    kw = dict(obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(5, 6), sensory_action_mode="absolute",
              resize_to_full=(out == "resize"), mask_out=(out == "mask"), antialias=aa)
    p = _pipe(num_envs=N, kind="flexible", frame_stack=fs, **kw)
    orcs = [O.FlexibleFovealOracle(**kw) for _ in range(N)]
    for step in range(6):
        st = rng.integers(0, 256, (N, fs, 84, 84), dtype=np.uint8)
        p.set_stack_u8(_t(st, dev))
        types = rng.integers(0, 2, N).astype(np.int32)
        a = np.where(types[:, None] == 1, rng.integers(10, 61, (N, 2)), rng.integers(-5, 80, (N, 2))).astype(np.int64)
        if step == 1:
            types[0], a[0] = 1, (84, 84)
            types[1], a[1] = 1, (31, 10)
        obs, loc, res = p.fovea(_t(a, dev), action_type=_t(types, dev))
        obs, loc, res = obs.cpu().numpy(), loc.cpu().numpy(), res.cpu().numpy()
        for i in range(N):
            want = orcs[i].step(unit64(st[i]), a[i], np.array((types[i],)))
            assert np.array_equal(loc[i], orcs[i].fov_loc) and np.array_equal(res[i], orcs[i].fov_res)
            got = obs[i]
            if out == "raw":
                rh, rw = res[i]
                assert not got[:, rh:, :].any() and not got[:, :, rw:].any()
                got = got[:, :rh, :rw]
            np.testing.assert_allclose(got, want, rtol=0, atol=FLOAT_TOL)


# ---------------------------------------------------------------- full benchmark size: properties
def test_full_size_properties(dev):
    """N=1024 (BASELINE.json configs[1]): size-independent checks instead of the slow oracle."""
    N, fs = 1024, 4
    g = torch.Generator(device="cpu").manual_seed(1234)
    p = _pipe(num_envs=N, kind="fixed", obs_size=(84, 84), fov_size=(30, 30), frame_stack=fs, resize_to_full=True)
    pm = _pipe(num_envs=N, kind="fixed", obs_size=(84, 84), fov_size=(30, 30), frame_stack=fs, mask_out=True)
    pr = _pipe(num_envs=N, kind="fixed", obs_size=(84, 84), fov_size=(30, 30), frame_stack=fs)
    # (1) constant-colour frames -> constant observation of that luminance, every env, every pixel
    cols = torch.randint(0, 256, (N, 3), generator=g, dtype=torch.uint8)
    frames = cols.view(N, 1, 1, 1, 3).expand(N, 2, 210, 160, 3).contiguous().to(dev)
    cmd = torch.full((N,), 2, dtype=torch.uint8, device=dev)
    lum = torch.from_numpy(O.ale_luminance(cols.numpy())).to(dev)
    for q in (p, pm, pr):
        for _ in range(fs):
            q.ingest(frames, cmd)
        st = q.stack_u8()
        assert (st == lum.view(N, 1, 1, 1)).all()
    a = (torch.rand((N, 2), generator=g) * 65 - 5).to(dev)
    obs, loc = p.fovea(a)
    want = (lum.float() / 255.0).view(N, 1, 1, 1)
    assert (obs - want).abs().max().item() <= 2e-7          # interpolating a constant gives the constant
    # (2) random frames: ingest twice with identical inputs == idempotent newest frame; max-pool commutes
    fr = torch.randint(0, 256, (N, 2, 210, 160, 3), generator=g, dtype=torch.uint8).to(dev)
    p.ingest(fr, cmd)
    s1 = p.stack_u8()[:, -1].clone()
    p.ingest(fr.flip(1).contiguous(), cmd)
    s2 = p.stack_u8()
    assert torch.equal(s2[:, -1], s1) and torch.equal(s2[:, -2], s1)
    one = torch.full((N,), 1, dtype=torch.uint8, device=dev)
    p.ingest(fr, one)
    a0 = p.stack_u8()[:, -1].clone()
    p.ingest(fr.flip(1).contiguous(), one)
    a1 = p.stack_u8()[:, -1]
    assert torch.equal(torch.maximum(a0, a1), s1)            # max of single-frame results == 2-frame result
    # (3) the three output modes agree with each other on the window
    for q in (pm, pr):
        q.set_stack_u8(p.stack_u8())
    obs_m, loc_m = pm.fovea(a)
    obs_r, loc_r = pr.fovea(a)
    obs_z, loc_z = p.fovea(a)
    assert torch.equal(loc_m, loc_r) and torch.equal(loc_m, loc_z)
    full = p.observe_full()
    ar = torch.arange(30, device=dev)
    rows = (loc_m[:, 0:1] + ar).long()
    cols_ = (loc_m[:, 1:2] + ar).long()
    win = full[torch.arange(N, device=dev)[:, None, None, None], torch.arange(fs, device=dev)[None, :, None, None],
               rows[:, None, :, None], cols_[:, None, None, :]]
    assert torch.equal(obs_r, win)
    expect_m = torch.zeros_like(obs_m)
    expect_m[torch.arange(N, device=dev)[:, None, None, None], torch.arange(fs, device=dev)[None, :, None, None],
             rows[:, None, :, None], cols_[:, None, None, :]] = win
    assert torch.equal(obs_m, expect_m)
    # resize output is a convex combination of the window: bounded by its min/max, corners exact
    lo = win.amin((2, 3), keepdim=True)
    hi = win.amax((2, 3), keepdim=True)
    assert (obs_z >= lo - 1e-6).all() and (obs_z <= hi + 1e-6).all()
    assert torch.allclose(obs_z[..., 0, 0], win[..., 0, 0], atol=1e-6) and \
        torch.allclose(obs_z[..., -1, -1], win[..., -1, -1], atol=1e-6)
    # (4) torch's own bilinear on the same window (float32 reference of the float kernel)
    ref = torch.nn.functional.interpolate(win, size=(84, 84), mode="bilinear", align_corners=False)
    assert (obs_z - ref).abs().max().item() <= FLOAT_TOL


# ---------------------------------------------------------------- ABI error behaviour
def test_abi_error_paths(dev):
    import ctypes as C
    from active_gym import _native as nat
    lib = nat.lib()

    def make(**over):
        cfg = nat.AgxConfig()
        cfg.struct_size = C.sizeof(nat.AgxConfig)
        cfg.device, cfg.num_envs, cfg.kind = 0, 4, nat.KIND_FIXED
        cfg.raw_h, cfg.raw_w, cfg.obs_h, cfg.obs_w, cfg.frame_stack = 210, 160, 84, 84, 4
        cfg.fov_h, cfg.fov_w, cfg.out_mode, cfg.action_mode, cfg.antialias = 30, 30, nat.OUT_RESIZE, nat.MODE_ABSOLUTE, 1
        for k, v in over.items():
            setattr(cfg, k, v)
        ctx = C.c_void_p()
        rc = lib.agx_create(C.byref(cfg), C.byref(ctx))
        return rc, ctx

    for over, frag in [(dict(num_envs=0), "num_envs"), (dict(num_envs=70000), "num_envs"), (dict(raw_h=200), "raw screen"),
                       (dict(obs_w=86), "obs_size"), (dict(frame_stack=0), "frame_stack"), (dict(frame_stack=17), "frame_stack"),
                       (dict(fov_h=84), "fov_size"), (dict(fov_w=0), "fov_size"), (dict(kind=9), "kind"),
                       (dict(out_mode=5), "out_mode"), (dict(action_mode=3), "action_mode"), (dict(device=99), "device"),
                       (dict(action_mode=nat.MODE_RELATIVE, sas_lo=2.0, sas_hi=1.0), "sensory_action_space"),
                       (dict(kind=nat.KIND_PERIPHERAL, per_h=0, per_w=20), "peripheral_res")]:
        rc, ctx = make(**over)
        assert rc == nat.E_INVALID and not ctx.value, over
        assert frag in nat.last_error(None), (over, nat.last_error(None))
    cfg_bad = dict(init_loc=(C.c_double * 2)(60.0, 0.0))
    rc, ctx = make(**cfg_bad)
    assert rc == nat.E_INVALID and "fov_init_loc" in nat.last_error(None)

    rc, ctx = make()
    assert rc == nat.OK and ctx.value
    obs = torch.empty((4, 4, 84, 84), device=dev)
    P = C.c_void_p
    # wrong wrapper kind for this context
    assert lib.agx_fovea_peripheral(ctx, None, 0, None, P(obs.data_ptr()), None, None) == nat.E_STATE
    assert "kind" in nat.last_error(ctx)
    assert lib.agx_fovea_flexible(ctx, None, 0, None, None, P(obs.data_ptr()), None, None, None) == nat.E_STATE
    # null buffers / bad dtype
    assert lib.agx_fovea_fixed(ctx, None, 0, None, None, None, None) == nat.E_INVALID
    assert lib.agx_ingest(ctx, None, None, None) == nat.E_INVALID
    assert lib.agx_fovea_fixed(ctx, P(obs.data_ptr()), 7, None, P(obs.data_ptr()), None, None) == nat.E_INVALID
    assert "dtype" in nat.last_error(ctx)
    assert lib.agx_algorithmic_bytes(ctx, nat.K_FOVEA) == 4 * 4 * (900 + 4 * 7056)
    assert lib.agx_algorithmic_bytes(ctx, nat.K_INGEST) == 4 * (2 * 168 * 480 + 7056)
    assert lib.agx_algorithmic_bytes(ctx, 99) == nat.E_INVALID
    assert lib.agx_destroy(ctx) == nat.OK
    # base context has no fovea
    rc, ctx = make(kind=nat.KIND_BASE)
    assert rc == nat.OK
    assert lib.agx_fovea_reset(ctx, None, None) == nat.E_STATE and lib.agx_fovea_fixed(ctx, None, 0, None, P(obs.data_ptr()), None, None) == nat.E_STATE
    assert lib.agx_destroy(ctx) == nat.OK and lib.agx_destroy(None) == nat.OK
    torch.cuda.synchronize()


def test_large_and_odd_batches(dev):
    """N not a multiple of anything, and a batch big enough to exceed one round of workgroups."""
    for N in (1, 3, 257, 2000):
        rng = np.random.default_rng(N)
        p = _pipe(num_envs=N, kind="fixed", obs_size=(84, 84), fov_size=(30, 30), frame_stack=4, resize_to_full=True)
        st = torch.randint(0, 256, (N, 4, 84, 84), dtype=torch.uint8, device=dev)
        p.set_stack_u8(st)
        a = torch.rand((N, 2), device=dev) * 54
        obs, loc = p.fovea(a)
        want_loc = torch.round(a.clamp(0, 54)).to(torch.int32)            # round-half-even like rint
        assert torch.equal(loc, want_loc)
        ar = torch.arange(30, device=dev)
        full = st.float() / 255.0
        idx = torch.arange(N, device=dev)
        win = full[idx[:, None, None, None], torch.arange(4, device=dev)[None, :, None, None],
                   (loc[:, 0:1] + ar).long()[:, None, :, None], (loc[:, 1:2] + ar).long()[:, None, None, :]]
        ref = torch.nn.functional.interpolate(win, size=(84, 84), mode="bilinear", align_corners=False)
        assert (obs - ref).abs().max().item() <= FLOAT_TOL
        if N <= 3:
            frames = rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8)
            p.ingest(_t(frames, dev), torch.full((N,), 2, dtype=torch.uint8, device=dev))
            newest = p.stack_u8()[:, -1].cpu().numpy()
            for i in range(N):
                assert np.array_equal(newest[i], np.maximum(O.get_state_u8(frames[i, 0], (84, 84)), O.get_state_u8(frames[i, 1], (84, 84))))


# ---------------------------------------------------------------- fused step == ingest + fovea
@pytest.mark.parametrize("geom", ["headline", "generic"])
def test_step_fixed_equals_separate_calls(dev, geom):
    """agx_step_fixed (one call per env step) against agx_ingest + agx_fovea_fixed and the oracle.  (Its other launch forms -
    fused, split, one workgroup per env - exist in the experiments build only: tests/test_gpu_variants.py.)"""
    N, fs = 37, 4
    rng = np.random.default_rng(77)
    fov = (30, 30) if geom == "headline" else (26, 34)
    kw = dict(num_envs=N, kind="fixed", obs_size=(84, 84), fov_size=fov, frame_stack=fs, resize_to_full=True,
              fov_init_loc=(3, 4), sensory_action_mode="relative", sensory_action_space=(-12.0, 12.0))
    a = _pipe(**kw)                                 # through agx_step_fixed
    b = _pipe(**kw)                                 # through the two stand-alone entry points
    ring = O.RingOracle(N, fs, (84, 84))
    orcs = [O.FixedFovealOracle(obs_size=(84, 84), fov_size=fov, fov_init_loc=(3, 4), sensory_action_mode="relative",
                                sensory_action_space=(-12.0, 12.0), resize_to_full=True) for _ in range(N)]
    for step in range(9):
        frames = _t(rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8), dev)
        nvalid = rng.integers(0, 3, N)
        clear = (rng.random(N) < 0.25).astype(np.uint8)
        skip = (rng.random(N) < 0.2).astype(np.uint8)
        nvalid[clear == 1] = 1
        cmd = _t((nvalid | clear * 4 | skip * 8).astype(np.uint8), dev)
        act_np = rng.uniform(-15, 15, (N, 2))
        act = _t(act_np, dev)
        oa, la = a.step_fixed(frames, cmd, act)
        b.ingest(frames, cmd)
        ob, lb = b.fovea(act)
        assert torch.equal(la, lb), step
        assert torch.equal(oa, ob), f"step {step}: agx_step_fixed and the separate calls must agree bit for bit"
        assert torch.equal(a.stack_u8(), b.stack_u8())
        ring.ingest(frames.cpu().numpy(), nvalid, clear=clear, skip=skip)
        full = ring.full_state()
        oa_np, la_np = oa.cpu().numpy(), la.cpu().numpy()
        for i in range(0, N, 5):
            want = orcs[i].step(full[i], act_np[i])
            assert np.array_equal(la_np[i], orcs[i].fov_loc)
            np.testing.assert_allclose(oa_np[i], want, rtol=0, atol=FLOAT_TOL)
        for i in range(N):
            if i % 5:
                orcs[i].update_loc(act_np[i])


# ---------------------------------------------------------------- K1g: ALE grayscale screens in
@pytest.mark.parametrize("obs", [(84, 84), (64, 64)])
def test_ingest_gray_raw_matches_oracle_and_rgb_path(dev, obs):
    """agx_ingest_gray_raw takes getScreenGrayscale screens (what the reference itself resizes, atari_env.py:74):
    bit-exact against the oracle's OpenCV restatement on arbitrary gray input, and - when the gray screens are
    ALE's luminance of RGB screens - identical to the RGB path's ring."""
    N, fs = 21, 4
    g = _pipe(num_envs=N, kind="base", obs_size=obs, frame_stack=fs)
    c = _pipe(num_envs=N, kind="base", obs_size=obs, frame_stack=fs)
    rng = np.random.default_rng(31)
    ring = np.zeros((N, fs) + obs, np.uint8)
    for step in range(6):
        rgb = rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8)
        gray = O.ale_luminance(rgb) if step % 2 == 0 else rng.integers(0, 256, (N, 2, 210, 160), dtype=np.uint8)
        nvalid = rng.integers(0, 3, N)
        clear = (rng.random(N) < 0.2).astype(np.uint8)
        skip = (rng.random(N) < 0.15).astype(np.uint8)
        nvalid[clear == 1] = 1
        cmd = _t((nvalid | clear * 4 | skip * 8).astype(np.uint8), dev)
        g.ingest_gray_raw(_t(gray, dev), cmd)
        for i in range(N):
            if skip[i]:
                continue
            if clear[i]:
                ring[i] = 0
            new = np.zeros(obs, np.uint8)
            for f in range(int(nvalid[i])):
                new = np.maximum(new, O.cv_resize_linear_u8(gray[i, f], obs))
            ring[i] = np.concatenate([ring[i, 1:], new[None]], 0)
        assert np.array_equal(g.stack_u8().cpu().numpy(), ring), step
    # same screens through both front ends
    a = _pipe(num_envs=N, kind="base", obs_size=obs, frame_stack=fs)
    b = _pipe(num_envs=N, kind="base", obs_size=obs, frame_stack=fs)
    for step in range(3):
        rgb = rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8)
        cmd = _t(np.full(N, 2 | (4 if step == 0 else 0), np.uint8), dev)
        a.ingest(_t(rgb, dev), cmd)
        b.ingest_gray_raw(_t(O.ale_luminance(rgb), dev), cmd)
        assert torch.equal(a.stack_u8(), b.stack_u8())
    assert g.algorithmic_bytes("ingest_gray_raw") * 3 - 2 * N * obs[0] * obs[1] == g.algorithmic_bytes("ingest")
    with pytest.raises(ValueError):
        g.ingest_gray_raw(_t(np.zeros((N, 2, 210, 160, 3), np.uint8), dev), cmd)
    for p_ in (g, c, a, b):
        p_.close()


# ---------------------------------------------------------------- fallback kernels == the tuned ones
@pytest.mark.parametrize("obs", [(84, 84), (48, 48)])
def test_general_ingest_kernel_matches_band12(dev, obs, monkeypatch):
    """AGX_INGEST_NO_FULL routes the ingest through k_ingest<256>, the kernel every geometry outside the band12 plan gets:
    same ring, bit for bit, incl. exact .5 luminance ties, clears, skips and short steps."""
    N, fs = 23, 3
    kw = dict(num_envs=N, kind="base", obs_size=obs, frame_stack=fs)
    monkeypatch.delenv("AGX_INGEST_NO_FULL", raising=False)
    d = _pipe(**kw)
    monkeypatch.setenv("AGX_INGEST_NO_FULL", "1")
    g = _pipe(**kw)
    rng = np.random.default_rng(13)
    ties = _tie_pixels()
    for step in range(6):
        fr = rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8)
        fr[step % N].reshape(-1, 3)[: len(ties)] = ties
        nvalid = rng.integers(0, 3, N)
        clear = (rng.random(N) < 0.2).astype(np.uint8)
        skip = (rng.random(N) < 0.15).astype(np.uint8)
        nvalid[clear == 1] = 1
        cmd = _t((nvalid | clear * 4 | skip * 8).astype(np.uint8), dev)
        d.ingest(_t(fr, dev), cmd)
        g.ingest(_t(fr, dev), cmd)
        assert torch.equal(d.stack_u8(), g.stack_u8()), step
    d.close()
    g.close()


@pytest.mark.parametrize("knob", ["AGX_FOVEA_GENERIC", "V2"])
@pytest.mark.parametrize("kind", ["peripheral", "flexible"])
def test_generic_fallback_kernel_matches_tuned(dev, kind, knob, monkeypatch):
    """AGX_FOVEA_GENERIC routes K3 / K4 through k_fovea_generic (the fallback for geometries whose tables do not fit
    the tuned kernels' LDS plan), AGX_PER_V2 / AGX_FLEX_V2 through the pass-by-pass tuned forms (k_fovea_peripheral2 /
    k_fovea_flexible2: the fallbacks for tap counts outside the composed-operator kernels' plan); same results up to
    float summation order."""
    if knob == "V2":
        knob = "AGX_PER_V2" if kind == "peripheral" else "AGX_FLEX_V2"
    N, fs = 19, 4
    kw = dict(num_envs=N, kind=kind, obs_size=(84, 84), fov_size=(30, 30), frame_stack=fs, resize_to_full=True,
              fov_init_loc=(0, 0), sensory_action_mode="absolute")
    if kind == "peripheral":
        kw["peripheral_res"] = (20, 20)
    for k_ in ("AGX_FOVEA_GENERIC", "AGX_PER_V2", "AGX_FLEX_V2"):
        monkeypatch.delenv(k_, raising=False)
    d = _pipe(**kw)
    monkeypatch.setenv(knob, "1")
    g = _pipe(**kw)
    rng = np.random.default_rng(12)
    for step in range(6):
        st = _t(rng.integers(0, 256, (N, fs, 84, 84), dtype=np.uint8), dev)
        d.set_stack_u8(st)
        g.set_stack_u8(st)
        if kind == "flexible":
            types = rng.integers(0, 2, N)
            a = np.where(types[:, None] == 1, rng.integers(8, 80, (N, 2)), rng.integers(-5, 60, (N, 2))).astype(np.int64)
            rd = d.fovea(_t(a, dev), action_type=_t(types.astype(np.int32), dev))
            rg = g.fovea(_t(a, dev), action_type=_t(types.astype(np.int32), dev))
        else:
            a = rng.uniform(-5, 60, (N, 2)).astype(np.float32)
            rd, rg = d.fovea(_t(a, dev)), g.fovea(_t(a, dev))
        for x, y in zip(rd[1:], rg[1:]):
            assert torch.equal(x, y)
        assert (rd[0] - rg[0]).abs().max().item() <= 2e-6, step
    d.close()
    g.close()


@pytest.mark.parametrize("generic", [False, True])
def test_large_geometries_are_refused_at_create_or_run(dev, generic, monkeypatch):
    """Found by tools/fuzz_gpu.py: agx_create must size the LDS of the kernel that will actually run (tuned or generic
    fallback), so that a geometry is either refused there with a message or launches - never an 'invalid argument' at
    the first fovea call."""
    from active_gym import _native as nat
    if generic:
        monkeypatch.setenv("AGX_FOVEA_GENERIC", "1")
    else:
        monkeypatch.delenv("AGX_FOVEA_GENERIC", raising=False)
    rng = np.random.default_rng(5)
    built = refused = 0
    for obs, fov, per in (((128, 128), (64, 64), (100, 100)), ((128, 128), (100, 90), (127, 127)), ((124, 128), (20, 30), (300, 60)),
                          ((96, 96), (40, 40), (50, 50)), ((256, 256), (64, 64), (32, 32))):
        try:
            p = _pipe(num_envs=3, kind="peripheral", obs_size=obs, fov_size=fov, fov_init_loc=(0, 0), frame_stack=2,
                      sensory_action_mode="absolute", resize_to_full=True, peripheral_res=per)
        except nat.AgxError as e:
            assert "LDS" in str(e) or "peripheral_res" in str(e)
            refused += 1
            continue
        p.set_stack_u8(_t(rng.integers(0, 256, (3, 2) + obs, dtype=np.uint8), dev))
        o, loc = p.fovea(_t(rng.uniform(0, 60, (3, 2)).astype(np.float32), dev))
        assert torch.isfinite(o).all()
        built += 1
        p.close()
    assert built >= 1 and refused >= 1


def test_profile_next_stamps_kernel_scoped_events(dev):
    """agx_profile_next: the next launch of a kernel family stamps a start / stop HIP event pair with its own begin /
    end (hipExtLaunchKernelGGL); one-shot; results unchanged."""
    N = 256
    kw = dict(num_envs=N, kind="fixed", obs_size=(84, 84), fov_size=(30, 30), frame_stack=4, resize_to_full=True,
              fov_init_loc=(0, 0), sensory_action_mode="absolute")
    a, b = _pipe(**kw), _pipe(**kw)
    rng = np.random.default_rng(3)
    frames = _t(rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8), dev)
    cmd = _t(np.full(N, 2, np.uint8), dev)
    act = _t(rng.uniform(0, 54, (N, 2)).astype(np.float32), dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for e in ev:
        e.record()                       # creates the handles
    torch.cuda.synchronize()
    for it in range(3):
        if it == 1:
            a.profile_next("ingest", ev[0], ev[1])
            a.profile_next("fovea", ev[2], ev[3])
        a.ingest(frames, cmd)
        oa, la = a.fovea(act)
        b.ingest(frames, cmd)
        ob, lb = b.fovea(act)
        assert torch.equal(oa, ob) and torch.equal(la, lb) and torch.equal(a.stack_u8(), b.stack_u8())
    torch.cuda.synchronize()
    t_ing, t_fov = ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])
    assert 0.002 < t_ing < 5.0 and 0.002 < t_fov < 5.0, (t_ing, t_fov)          # ms: a real kernel interval each
    assert ev[1].elapsed_time(ev[2]) >= 0.0                                      # ingest ended before fovea began
    from active_gym import _native as nat
    with pytest.raises(nat.AgxError):
        nat.check(nat.lib().agx_profile_next(a._ctx, 99, None, None), a._ctx)
    a.close()
    b.close()


def test_c_abi_demo_matches_python_binding(dev, tmp_path):
    """examples/c_abi_demo.cpp drives libagx from plain C++ (hipMalloc buffers, no torch, no Python); the same LCG
    inputs through the ctypes binding must give the same ring / fov_loc / observation checksums."""
    import json
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not on this box")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_demo")
    libdir = os.path.join(repo, "active-gym_amd", "lib")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-I", os.path.join(repo, "include"),
                    os.path.join(repo, "examples", "c_abi_demo.cpp"), "-o", exe, "-L", libdir, "-lagx",
                    f"-Wl,-rpath,{libdir}"], check=True, timeout=600)
    N, steps, fs = 24, 4, 4
    got = json.loads(subprocess.run([exe, str(N), str(steps)], check=True, capture_output=True, text=True, timeout=120)
                     .stdout.strip().splitlines()[-1])

    def lcg_block(state, count):
        """count consecutive LCG outputs, vectorised by jumping: x_{k} = A_k x_0 + C_k (mod 2^32)."""
        A = np.empty(count, np.uint64)
        Cc = np.empty(count, np.uint64)
        a, c, m = 1664525, 1013904223, 0xFFFFFFFF
        # doubling construction of (A_k, C_k)
        A[0], Cc[0] = a, c
        n = 1
        while n < count:
            k = min(n, count - n)
            An, Cn = int(A[n - 1]), int(Cc[n - 1])
            A[n:n + k] = (A[:k] * np.uint64(An)) & np.uint64(m)
            Cc[n:n + k] = (A[:k] * np.uint64(Cn) + Cc[:k]) & np.uint64(m)
            n += k
        out = (A * np.uint64(state) + Cc) & np.uint64(m)
        return out.astype(np.uint32), int(out[-1])

    p = _pipe(num_envs=N, kind="fixed", obs_size=(84, 84), fov_size=(30, 30), frame_stack=fs, resize_to_full=True,
              fov_init_loc=(0, 0), sensory_action_mode="absolute")
    state = 12345
    nb = N * 2 * 210 * 160 * 3
    for t in range(steps):
        r, state = lcg_block(state, nb)
        frames = (r >> np.uint32(24)).astype(np.uint8).reshape(N, 2, 210, 160, 3)
        cmd = np.array([(1 | 4) if t == 0 else (1 if i % 7 == 3 else 2) for i in range(N)], np.uint8)
        r, state = lcg_block(state, 2 * N)
        act = ((r >> np.uint32(16)).astype(np.float32) * np.float32(65.0 / 65536.0) - np.float32(5.0)).reshape(N, 2)
        p.ingest(_t(frames, dev), _t(cmd, dev))
        obs, loc = p.fovea(_t(act, dev))
    ring = p.stack_u8().cpu().numpy().reshape(-1)
    obs = obs.cpu().numpy().reshape(-1)
    loc = loc.cpu().numpy().reshape(-1).astype(np.uint64)

    def fnv(words):
        h = 1469598103934665603
        for w in words.tolist():
            h = ((h ^ w) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h

    from active_gym import _native as nat
    assert got["abi"] == nat.ABI_VERSION and got["N"] == N and got["steps"] == steps
    assert got["ring_sum"] == int(ring.astype(np.uint64).sum()) and got["ring_hash"] == fnv(ring)
    assert got["loc_sum"] == int((loc * np.arange(1, 2 * N + 1, dtype=np.uint64)).sum())
    assert got["obs_hash"] == fnv(obs.view(np.uint32)), "float observations must agree bit for bit"
    assert abs(got["obs_sum"] - float(obs.astype(np.float64).sum())) < 1e-6
    p.close()


# ---------------------------------------------------------------- streams, several contexts, lifetime
def test_non_default_stream_and_interleaved_contexts(dev):
    N = 12
    rng = np.random.default_rng(5)
    kw = dict(num_envs=N, kind="fixed", obs_size=(84, 84), fov_size=(30, 30), frame_stack=2, resize_to_full=True)
    a, b = _pipe(**kw), _pipe(**kw)
    fa = _t(rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8), dev)
    fb = _t(rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8), dev)
    cmd = torch.full((N,), 2, dtype=torch.uint8, device=dev)
    act = _t(rng.uniform(0, 54, (N, 2)).astype(np.float32), dev)
    ref = _pipe(**kw)
    ref.ingest(fa, cmd)
    want_a, _ = ref.fovea(act)
    ref2 = _pipe(**kw)
    ref2.ingest(fb, cmd)
    want_b, _ = ref2.fovea(act)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s1):
        a.ingest(fa, cmd)
    with torch.cuda.stream(s2):
        b.ingest(fb, cmd)
    with torch.cuda.stream(s1):
        oa, _ = a.fovea(act)
    with torch.cuda.stream(s2):
        ob, _ = b.fovea(act)
    s1.synchronize()
    s2.synchronize()
    assert torch.equal(oa, want_a) and torch.equal(ob, want_b)


def test_create_destroy_does_not_leak(dev):
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    for kind, extra in (("fixed", dict(resize_to_full=True)), ("peripheral", dict(peripheral_res=(20, 20))),
                        ("flexible", dict(resize_to_full=True)), ("base", dict())):
        for _ in range(6):
            p = _pipe(num_envs=512, kind=kind, obs_size=(84, 84), fov_size=(30, 30), frame_stack=4, **extra)
            p.close()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free0 - free1 < 64 << 20, f"leaked {(free0 - free1) >> 20} MiB"


# ---------------------------------------------------------------- K4 raw crops, packed (ragged) output
@pytest.mark.parametrize("name", [n for n in fovea_case_names() if n.startswith("flex_raw")])
def test_flexible_raw_packed_golden_replay(dev, name):
    """agx_fovea_flexible_packed against the reference-run goldens: the ragged [fs, rh, rw] crops themselves
    (fov_env.py:283-298), no padding."""
    c = load_fovea(name)
    p = _pipe_for_case(c)
    states = c["states_u8"]
    p.set_stack_u8(_t(states[0][None], dev))
    p.fovea_reset()
    packed, off, loc, res = p.fovea_packed()
    fs = c["frame_stack"]

    def view():
        o, r = off.cpu().numpy(), res.cpu().numpy()[0]
        assert o[0] == 0 and o[1] == fs * r[0] * r[1]
        return packed[:int(o[1])].cpu().numpy().reshape(fs, int(r[0]), int(r[1])), r
    g, r = view()
    w = c["outs"][0]
    assert g.shape == w.shape
    np.testing.assert_allclose(g.astype(np.float64), w.astype(np.float64), rtol=0, atol=FLOAT_TOL)
    for t in range(c["steps"]):
        p.set_stack_u8(_t(states[t + 1][None], dev))
        a = _t(c["actions"][t][None], dev)
        at = _t(np.array([c["action_types"][t]], np.int32), dev)
        packed, off, loc, res = p.fovea_packed(a, action_type=at, packed=packed, offsets=off)
        assert np.array_equal(res.cpu().numpy()[0], c["fov_res"][t + 1]) and np.array_equal(loc.cpu().numpy()[0], c["fov_loc"][t + 1])
        g, r = view()
        w = c["outs"][t + 1]
        assert g.shape == w.shape, (t, g.shape, w.shape)
        np.testing.assert_allclose(g.astype(np.float64), w.astype(np.float64), rtol=0, atol=FLOAT_TOL)
        if int(r[0]) <= int(c["fov_size"][0]):
            assert np.array_equal(g, w.astype(np.float32)), "an unsqueezed crop must be bit-exact"


@pytest.mark.parametrize("aa", [False, True])
@pytest.mark.parametrize("generic", [False, True])
def test_flexible_packed_equals_padded(dev, aa, generic, monkeypatch):
    """The packed layout holds exactly the valid [0:rh, 0:rw] part of the padded raw-crop buffer, env after env, and the
    capacity guard never writes past the buffer."""
    for k_ in ("AGX_FOVEA_GENERIC", "AGX_FLEX_V2"):
        monkeypatch.delenv(k_, raising=False)
    if generic:
        monkeypatch.setenv("AGX_FOVEA_GENERIC", "1")
    N, fs = (37, 3) if generic else (700, 3)        # 700 envs: three blocks of the two-level offset scan (256 envs each)
    kw = dict(num_envs=N, kind="flexible", obs_size=(84, 84), fov_size=(30, 30), frame_stack=fs, fov_init_loc=(3, 4),
              sensory_action_mode="absolute", antialias=aa)
    a_, b_ = _pipe(**kw), _pipe(**kw)
    rng = np.random.default_rng(3 + aa)
    for step in range(5):
        st = _t(rng.integers(0, 256, (N, fs, 84, 84), dtype=np.uint8), dev)
        a_.set_stack_u8(st)
        b_.set_stack_u8(st)
        types = rng.integers(0, 2, N).astype(np.int32)
        act = np.where(types[:, None] == 1, rng.integers(1, 85, (N, 2)), rng.integers(-5, 90, (N, 2))).astype(np.float64)
        pad, loc_a, res_a = a_.fovea(_t(act, dev), action_type=_t(types, dev))
        packed, off, loc_b, res_b = b_.fovea_packed(_t(act, dev), action_type=_t(types, dev))
        assert torch.equal(loc_a, loc_b) and torch.equal(res_a, res_b)
        o, r = off.cpu().numpy(), res_b.cpu().numpy()
        assert o[0] == 0 and np.array_equal(np.diff(o), fs * r[:, 0].astype(np.int64) * r[:, 1])
        flat, pad = packed.cpu().numpy(), pad.cpu().numpy()
        for i in range(N):
            got = flat[o[i]:o[i + 1]].reshape(fs, r[i, 0], r[i, 1])
            assert np.array_equal(got, pad[i, :, :r[i, 0], :r[i, 1]]), (step, i)
    # capacity guard: a buffer that holds only the first envs; the canary behind it must survive
    total = int(off.cpu()[-1])
    cut = int(off.cpu()[N // 2])
    small = torch.full((cut + 8,), -7.0, dtype=torch.float32, device=dev)
    _, off2, _, _ = b_.fovea_packed(None, packed=small[:cut])
    assert int(off2.cpu()[-1]) == total and (small[cut:] == -7.0).all()
    assert np.array_equal(small[:cut].cpu().numpy(), flat[:cut])
    a_.close()
    b_.close()

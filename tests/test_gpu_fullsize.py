"""BASELINE.json configs[2] and configs[3] at their full size (N = 1024 per GPU, 84x84 obs, 30x30 fovea, frame_stack 4):
K3 (FixedFovealPeripheralEnv, peripheral_res 20x20) and K4 (FlexibleFovealEnv, per-env ragged windows) through the C
ABI, checked against torch's own `interpolate` (the backend of torchvision's Resize, float64 like the reference feeds
it: fov_env.py:276-298, 366-388) and against the NumPy clip / rint rules of the reference (fov_env.py:166-170,270-271,
300-324).  Bars: indices and pasted / cropped pixels bit-exact, resized pixels within 1e-5."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
FLOAT_TOL = 1e-5
N, FS, OBS, FOV = 1024, 4, 84, 30


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _pipe(**kw):
    from active_gym import ObsPipeline
    return ObsPipeline(num_envs=N, obs_size=(OBS, OBS), fov_size=(FOV, FOV), frame_stack=FS, fov_init_loc=(0, 0),
                       sensory_action_mode="absolute", **kw)


def _resize(x, size, aa):
    """torchvision Resize(size)(x) on float64 [.., H, W]: unchanged when the size matches, else interpolate."""
    if tuple(x.shape[-2:]) == tuple(size):
        return x
    return F.interpolate(x, size=size, mode="bilinear", align_corners=False, antialias=aa)


@pytest.mark.parametrize("aa", [True, False])
def test_peripheral_full_size(dev, aa):
    g = torch.Generator(device="cpu").manual_seed(77 + aa)
    p = _pipe(kind="peripheral", peripheral_res=(20, 20), antialias=aa)
    for step in range(3):
        st = torch.randint(0, 256, (N, FS, OBS, OBS), generator=g, dtype=torch.uint8).to(dev)
        p.set_stack_u8(st)
        a = torch.rand((N, 2), generator=g) * 65 - 5
        if step == 1:
            a[:8] = torch.tensor([[0.5, 1.5], [2.5, 53.5], [54.0, 54.0], [-3.0, 80.0], [12.5, 12.49], [53.51, 0.0], [27.0, 27.0], [1e9, -1e9]])
        obs, loc = p.fovea(a.to(dev))
        want_loc = np.rint(np.clip(a.numpy().astype(np.float64), 0, OBS - FOV)).astype(np.int32)       # fov_env.py:166-167
        assert np.array_equal(loc.cpu().numpy(), want_loc)
        full = p.observe_full()                                                                       # float32 k/255 (exact, tested elsewhere)
        ref = _resize(_resize(full.double(), (20, 20), aa), (OBS, OBS), aa)                           # fov_env.py:375-377
        ar = torch.arange(FOV, device=dev)
        rows = (loc[:, 0:1] + ar).long()[:, None, :, None]
        cols = (loc[:, 1:2] + ar).long()[:, None, None, :]
        ni = torch.arange(N, device=dev)[:, None, None, None]
        ci = torch.arange(FS, device=dev)[None, :, None, None]
        assert torch.equal(obs[ni, ci, rows, cols], full[ni, ci, rows, cols]), "pasted fovea must be bit-exact"
        ref[ni, ci, rows, cols] = full[ni, ci, rows, cols].double()                                   # fov_env.py:383-387
        err = (obs.double() - ref).abs().max().item()
        assert err <= FLOAT_TOL, (step, err)
    p.close()


@pytest.mark.parametrize("aa", [True, False])
def test_flexible_full_size(dev, aa):
    rng = np.random.default_rng(5 + aa)
    p = _pipe(kind="flexible", resize_to_full=True, antialias=aa)
    loc_w = np.zeros((N, 2), np.int64)                                    # the reference's state, NumPy rules
    res_w = np.full((N, 2), FOV, np.int64)
    worst = 0.0
    for step in range(4):
        st = torch.from_numpy(rng.integers(0, 256, (N, FS, OBS, OBS), dtype=np.uint8)).to(dev)
        p.set_stack_u8(st)
        types = rng.integers(0, 2, N).astype(np.int32)
        a = np.where(types[:, None] == 1, rng.integers(10, 61, (N, 2)).astype(np.float64), rng.uniform(-5, 80, (N, 2)))
        if step == 2:                                                    # edges of the resolution range
            types[:6] = 1
            a[:6] = [(84, 84), (31, 10), (1, 1), (30, 84), (84, 1), (60.5, 59.5)]
        obs, loc, res = p.fovea(torch.from_numpy(a).to(dev), action_type=torch.from_numpy(types).to(dev))
        # fov_env.py:300-324 (+ the ABI's documented rint / clamp of FOV_RES values)
        is_res = types == 1
        res_w = np.where(is_res[:, None], np.rint(np.clip(a, 1, OBS)).astype(np.int64), res_w)
        loc_w = np.where(is_res[:, None], np.rint(np.clip(loc_w, 0, OBS - res_w)).astype(np.int64),
                         np.rint(np.clip(a, 0, OBS - res_w)).astype(np.int64))
        assert np.array_equal(res.cpu().numpy(), res_w) and np.array_equal(loc.cpu().numpy(), loc_w), step
        full = p.observe_full().double()
        errs = []
        for i in range(N):
            (r, c), (rh, rw) = loc_w[i], res_w[i]
            x = full[i:i + 1, :, r:r + rh, c:c + rw]
            if rh > FOV:                                                  # rows only, fov_env.py:286
                x = _resize(_resize(x, (FOV, FOV), aa), (int(rh), int(rw)), aa)
            x = _resize(x, (OBS, OBS), aa)
            errs.append((obs[i:i + 1].double() - x).abs().max())
        worst = max(worst, torch.stack(errs).max().item())
        assert worst <= FLOAT_TOL, (step, worst)
    p.close()


def test_large_batch_every_env_is_reached(dev):
    """N = 20,000 envs in one context (gridDim.y carries the env index; the ABI allows up to 65,535): constant-colour
    screens give every env - first, last, and a sample in between - its own constant observation, and fov_loc follows the
    NumPy rule for all of them."""
    from active_gym import ObsPipeline
    from oracle import oracle as O
    n = 20000
    p = ObsPipeline(num_envs=n, kind="fixed", obs_size=(OBS, OBS), fov_size=(FOV, FOV), frame_stack=2, fov_init_loc=(0, 0),
                    sensory_action_mode="absolute", resize_to_full=True, device=dev)
    g = torch.Generator(device="cpu").manual_seed(9)
    cols = torch.randint(0, 256, (n, 3), generator=g, dtype=torch.uint8)
    frames = cols.view(n, 1, 1, 1, 3).to(dev).expand(n, 2, 210, 160, 3).contiguous()
    cmd = torch.full((n,), 2, dtype=torch.uint8, device=dev)
    for _ in range(2):
        p.ingest(frames, cmd)
    del frames
    a = torch.rand((n, 2), generator=g) * 65 - 5
    obs, loc = p.fovea(a.to(dev))
    lum = torch.from_numpy(O.ale_luminance(cols.numpy())).to(dev)
    want = (lum.float() / 255.0).view(n, 1, 1, 1)
    assert (obs - want).abs().max().item() <= 2e-7
    assert torch.equal(p.stack_u8(), lum.view(n, 1, 1, 1).expand(n, 2, OBS, OBS))
    assert np.array_equal(loc.cpu().numpy(), np.rint(np.clip(a.numpy().astype(np.float64), 0, OBS - FOV)).astype(np.int32))
    p.close()


def test_ingest_and_fixed_fovea_full_size_vs_c_oracle(dev):
    """BASELINE.json configs[1] at the size the metric is quoted on: 1024 envs of RANDOM RGB screens with random nvalid / CLEAR /
    SKIP commands, three steps.  K1's ring must equal oracle/cport.c's bit for bit (VERDICT r03 Weak 4: full size was checked by
    properties only, random-pixel bit-exactness at N = 7); K2 (resize_to_full) on that very ring within 1e-5, fov_loc exact.
    Reference: atari_env.py:73-75,121-133 and fov_env.py:166-183."""
    from oracle import cport
    assert cport.available(), "oracle/_build/liboracle.so missing (__graft_entry__.build())"
    from active_gym import ObsPipeline
    rng = np.random.default_rng(20241)
    p = ObsPipeline(num_envs=N, kind="fixed", obs_size=(OBS, OBS), fov_size=(FOV, FOV), frame_stack=FS, fov_init_loc=(0, 0),
                    sensory_action_mode="absolute", resize_to_full=True)
    eb = cport.EnvBatch(N, frame_stack=FS, obs=(OBS, OBS), fov=(FOV, FOV))
    worst = 0.0
    for step in range(3):
        frames = rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8)
        nvalid = rng.integers(0, 3, N)
        nvalid[rng.random(N) < 0.8] = 2
        clear = (rng.random(N) < (0.05 if step else 0.5)).astype(np.uint8)
        skip = (rng.random(N) < 0.05).astype(np.uint8) if step else np.zeros(N, np.uint8)
        nvalid[clear == 1] = 1
        cmd = (nvalid | clear * 4 | skip * 8).astype(np.uint8)
        act = rng.uniform(-5, 60, (N, 2))
        act[::7] = np.floor(act[::7]) + 0.5                                   # exact .5 ties: half to even
        p.ingest(torch.from_numpy(frames).to(dev), torch.from_numpy(cmd).to(dev))
        obs, loc = p.fovea(torch.from_numpy(act).to(dev))                    # float64 actions, as the reference receives them
        eb.ingest(frames, cmd)
        want, want_loc = eb.fovea_fixed(act)
        got_ring = p.stack_u8().cpu().numpy()
        bad = np.nonzero((got_ring != eb.ring).reshape(N, -1).any(1))[0]
        assert len(bad) == 0, f"step {step}: K1 ring differs from oracle/cport.c for envs {bad[:8]} (cmd {cmd[bad[:8]]})"
        assert np.array_equal(loc.cpu().numpy(), want_loc), step
        err = float(np.abs(obs.cpu().numpy().astype(np.float64) - want).max())
        worst = max(worst, err)
        assert err <= FLOAT_TOL, (step, err)
    assert (eb.ring != 0).any() and worst > 0.0
    p.close()

"""GPU: the drop-in env API end to end (host runner -> pinned staging -> H2D ->
HIP ingest + fovea kernels) against the oracle's per-env chain
``FovealOracle(RecordOracle(AtariEnvOracle(emulator)))`` driven by the same
scripted emulators, same actions, same no-op draws.

Bars: u8 stack / fov_loc / fov_res / rewards / dones / counters exact; float
observations within 1e-5 (bit-exact for crop / mask modes)."""
import numpy as np
import pytest
import torch

from fake_ale import ScriptedALE
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _args(**kw):
    from active_gym import AtariEnvArgs
    base = dict(game="scripted", seed=7, obs_size=(84, 84), frame_source=lambda a, i: ScriptedALE(seed=300 + i, n_actions=4,
                                                                                                 p_life=0.05, p_over=0.01))
    base.update(kw)
    return AtariEnvArgs(**base)


class _Noops:
    """Two identical deterministic no-op streams (product runner / oracle envs)."""

    def __init__(self, seed):
        self.seq = np.random.default_rng(seed).integers(0, 30, size=10000).tolist()
        self.a = iter(self.seq)
        self.b = iter(self.seq)


def _oracle_env(i, args, noop_iter, kind, antialias=True):
    if getattr(args, "frame_source", None) == "native":       # the C++ runner's scripted emulator, mirrored in Python
        from lcg_ale import LcgALE
        ale = LcgALE(args.seed + i, args.scripted_actions, args.scripted_lives, args.scripted_p_life, args.scripted_p_over)
    else:
        ale = ScriptedALE(seed=300 + i, n_actions=4, p_life=0.05, p_over=0.01)
    env = O.AtariEnvOracle(ale, ale.getMinimalActionSet(), obs_size=(84, 84), frame_stack=args.frame_stack,
                           action_repeat=args.action_repeat, clip_reward=args.clip_reward,
                           noop_fn=lambda: int(next(noop_iter)),
                           prefer_rgb=getattr(args, "frame_format", "rgb") != "gray")   # gray: ale.getScreenGrayscale(), as the reference
    rec = O.RecordOracle(env)
    if kind == "base":
        return rec, None
    kw = dict(obs_size=(84, 84), fov_size=tuple(args.fov_size), fov_init_loc=tuple(args.fov_init_loc),
              sensory_action_mode=args.sensory_action_mode,
              sensory_action_space=getattr(args, "sensory_action_space", None), antialias=antialias)
    if kind == "fixed":
        fov = O.FixedFovealOracle(resize_to_full=args.resize_to_full, mask_out=args.mask_out, **kw)
    elif kind == "flexible":
        fov = O.FlexibleFovealOracle(resize_to_full=args.resize_to_full, mask_out=args.mask_out, **kw)
    else:
        fov = O.PeripheralOracle(peripheral_res=tuple(args.peripheral_res), **kw)
    return rec, fov


@pytest.mark.parametrize("kind,extra", [
    ("fixed", dict(resize_to_full=True)),
    ("fixed", dict(resize_to_full=False, mask_out=True, sensory_action_mode="relative", sensory_action_space=(-10.0, 10.0))),
    ("peripheral", dict(resize_to_full=False, peripheral_res=(20, 20))),
    ("flexible", dict(resize_to_full=True)),
    ("base", dict()),
    # the same chain fed by the native C++ host runner (libagx_runner.so) instead of the Python one
    ("fixed", dict(resize_to_full=True, frame_source="native", scripted_actions=4, scripted_lives=3, scripted_p_life=50,
                   scripted_p_over=10, h2d_chunk_envs=2)),        # chunked: H2D of chunk c overlaps emulation of c+1
    ("flexible", dict(resize_to_full=False, mask_out=True, frame_source="native", scripted_actions=4, scripted_lives=2,
                      scripted_p_life=50, scripted_p_over=10, clip_reward=True)),
    # frame_format="gray": the emulator's own grayscale screens travel (getScreenGrayscale, atari_env.py:74); the
    # scripted emulator's gray is a channel mean, i.e. independent of the luminance restatement
    ("fixed", dict(resize_to_full=True, frame_format="gray")),
    ("peripheral", dict(resize_to_full=False, peripheral_res=(20, 20), frame_format="gray", frame_source="native",
                        scripted_actions=4, scripted_lives=3, scripted_p_life=50, scripted_p_over=10, h2d_chunk_envs=3)),
    # device outputs + the native runner: the NATIVE STEP LOOP (agx_loop_step: one C call per step, autoreset inside) - every
    # kind, RGB and gray screens, compact and whole-screen staging; and the Python loop with device outputs beside it
    ("fixed", dict(resize_to_full=True, frame_source="native", device="cuda:0", scripted_actions=4, scripted_lives=3,
                   scripted_p_life=50, scripted_p_over=10)),
    ("fixed", dict(resize_to_full=True, frame_source="native", device="cuda:0", scripted_actions=4, scripted_lives=3,
                   scripted_p_life=50, scripted_p_over=10, compact_rows=False, clip_reward=True)),
    ("fixed", dict(resize_to_full=True, frame_source="native", device="cuda:0", scripted_actions=4, scripted_lives=3,
                   scripted_p_life=50, scripted_p_over=10, native_loop=False)),
    ("flexible", dict(resize_to_full=False, mask_out=True, frame_source="native", device="cuda:0", scripted_actions=4,
                      scripted_lives=2, scripted_p_life=50, scripted_p_over=10)),
    ("peripheral", dict(resize_to_full=False, peripheral_res=(20, 20), frame_format="gray", frame_source="native", device="cuda:0",
                        scripted_actions=4, scripted_lives=3, scripted_p_life=50, scripted_p_over=10)),
    ("base", dict(frame_source="native", device="cuda:0", frame_format="gray", scripted_actions=6, scripted_lives=3,
                  scripted_p_life=50, scripted_p_over=10)),
])
def test_vec_env_matches_oracle_with_autoreset(kind, extra):
    from active_gym import AtariVecEnv
    N, STEPS = 5, 60
    kw = dict(fov_size=(30, 30), fov_init_loc=(3.5, 4.49), sensory_action_mode="absolute", resize_to_full=True)
    kw.update(extra)
    args = _args(**kw)
    noops = _Noops(1)
    env = AtariVecEnv(args, N, kind=kind, noop_fn=lambda: int(next(noops.a)))
    on_device = getattr(args, "device", None) is not None
    want_loop = on_device and getattr(args, "native_loop", True) and getattr(args, "frame_source", None) == "native"
    assert (env._loop is not None) == bool(want_loop)
    n_act = int(getattr(args, "scripted_actions", 4))

    def _np(x):
        return x.cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)

    orcs = [_oracle_env(i, args, noops.b, kind) for i in range(N)]
    rng = np.random.default_rng(3)

    def fov_obs(i, state, a=None, t=None, reset=False):
        rec, fov = orcs[i]
        if fov is None:
            return state
        if reset:
            return fov.reset(state)
        if kind == "flexible":
            return fov.step(state, a, np.array((t,)))
        return fov.step(state, a)

    obs, infos = env.reset()
    assert isinstance(obs, torch.Tensor) == on_device
    obs = _np(obs)
    want = []
    for i in range(N):                      # same env order as the runner draws its no-ops
        s, info = orcs[i][0].reset()
        want.append(fov_obs(i, s, reset=True))
    assert obs.dtype == np.float32 and obs.shape[0] == N
    np.testing.assert_allclose(obs, np.stack(want), rtol=0, atol=TOL)
    n_done = 0
    for step in range(STEPS):
        motor = rng.integers(0, n_act, N)
        if args.__dict__.get("sensory_action_mode") == "relative":
            sens = rng.uniform(-14, 14, (N, 2))
        else:
            sens = rng.uniform(-5, 60, (N, 2))
        types = rng.integers(0, 2, N)
        if kind == "flexible":
            sens = np.where(types[:, None] == 1, rng.integers(8, 70, (N, 2)), np.rint(sens)).astype(np.int64)
        act = motor if kind == "base" else {"motor_action": motor, "sensory_action": sens}
        if kind == "flexible":
            act["sensory_action_type"] = types.reshape(N, 1)
        obs, rew, term, trunc, infos = env.step(act)
        assert not trunc.any()
        obs = _np(obs)
        if kind != "base":
            infos["fov_loc"] = _np(infos["fov_loc"])
            if kind == "flexible":
                infos["fov_res"] = _np(infos["fov_res"])
        for i in range(N):
            rec, fov = orcs[i]
            s, r, d, tr, info = rec.step(int(motor[i]))
            a_i = sens[i]
            if kind == "flexible" and types[i] == 1:
                a_i = np.clip(a_i, 1, 84)   # the ABI clamps FOV_RES into [1, obs]; the reference stores it raw
            o = fov_obs(i, s, a_i, int(types[i]))
            assert float(rew[i]) == float(r) and bool(term[i]) == bool(d), (step, i)
            if d:
                n_done += 1
                assert infos["_final_observation"][i] and infos["_final_info"][i]
                np.testing.assert_allclose(_np(infos["final_observation"][i]), o, rtol=0, atol=TOL)
                fi = infos["final_info"][i]
                assert fi["ep_len"] == info["ep_len"] and fi["reward"] == info["reward"] and fi["raw_reward"] == info["raw_reward"]
                if fov is not None:
                    assert np.array_equal(_np(fi["fov_loc"]), fov.fov_loc)
                    if kind == "flexible":
                        assert np.array_equal(_np(fi["fov_res"]), fov.fov_res)
                s, info = rec.reset()
                o = fov_obs(i, s, reset=True)
            np.testing.assert_allclose(obs[i], o, rtol=0, atol=TOL, err_msg=f"step {step} env {i}")
            assert infos["ep_len"][i] == info["ep_len"] and infos["reward"][i] == info["reward"]
            if fov is not None:
                assert np.array_equal(infos["fov_loc"][i], fov.fov_loc), (step, i)
                if kind == "flexible":
                    assert np.array_equal(infos["fov_res"][i], fov.fov_res)
        if kind != "base" and step % 10 == 0:
            assert np.array_equal(env.fov_loc, np.stack([o_[1].fov_loc for o_ in orcs]))
    assert n_done >= 2, "the scripted emulators should have produced terminals"
    # the device u8 stack equals the oracle's state_buffer numerators
    st = env.pipe.stack_u8().cpu().numpy()
    for i in range(N):
        full = np.stack(orcs[i][0].env.state_buffer, 0)
        assert np.array_equal(st[i], np.rint(full * 255).astype(np.uint8))
    env.close()


@pytest.mark.parametrize("factory,kind,extra", [
    ("AtariFixedFovealEnv", "fixed", dict(resize_to_full=True)),
    ("AtariFixedFovealEnv", "fixed", dict(resize_to_full=False)),
    ("AtariFlexibleFovealEnv", "flexible", dict(resize_to_full=False)),
    ("AtariFixedFovealPeripheralEnv", "peripheral", dict(resize_to_full=False, peripheral_res=(20, 20))),
    ("AtariBaseEnv", "base", dict()),
])
def test_single_env_drop_in(factory, kind, extra):
    import active_gym
    kw = dict(fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute", resize_to_full=True,
              frame_stack=3, clip_reward=True)
    kw.update(extra)
    args = _args(**kw)
    noops = _Noops(2)
    env = getattr(active_gym, factory)(args)
    env.unwrapped._core.runner.noop_fn = lambda: int(next(noops.a))
    rec, fov = _oracle_env(0, args, noops.b, kind)
    rng = np.random.default_rng(9)
    if kind == "base":
        obs, info = env.reset()
    else:
        obs, info = env.reset()
        assert isinstance(info["fov_loc"], np.ndarray) and info["fov_loc"].shape == (2,)
    s, oinfo = rec.reset()
    want = s if fov is None else fov.reset(s)
    assert obs.shape == want.shape and obs.dtype == np.float32
    np.testing.assert_allclose(obs, want, rtol=0, atol=TOL)
    assert info["ep_len"] == 0 and info["reward"] == 0
    done_seen = 0
    for step in range(80):
        m = int(rng.integers(0, 4))
        t = int(rng.integers(0, 2))
        sa = rng.integers(10, 50, 2) if (kind == "flexible" and t == 1) else rng.uniform(-3, 58, 2)
        if kind == "base":
            obs, r, d, tr, info = env.step(m)
        else:
            a = {"motor_action": m, "sensory_action": sa if step % 2 else torch.from_numpy(np.asarray(sa))}
            if kind == "flexible":
                a["sensory_action_type"] = np.array((t,))
            obs, r, d, tr, info = env.step(a)
        s, orr, od, _, oinfo = rec.step(m)
        if fov is None:
            want = s
        elif kind == "flexible":
            want = fov.step(s, np.asarray(sa), np.array((t,)))
        else:
            want = fov.step(s, np.asarray(sa))
        assert obs.shape == want.shape, (obs.shape, want.shape)
        np.testing.assert_allclose(obs, want, rtol=0, atol=TOL)
        assert r == orr and d == od and tr is False
        assert info["raw_reward"] == oinfo["raw_reward"] and info["reward"] == oinfo["reward"] and info["ep_len"] == oinfo["ep_len"]
        if fov is not None:
            assert np.array_equal(info["fov_loc"], fov.fov_loc) and np.array_equal(env.fov_loc, fov.fov_loc)
        if d:
            done_seen += 1
            obs, info = env.reset()
            s, oinfo = rec.reset()
            want = s if fov is None else fov.reset(s)
            np.testing.assert_allclose(obs, want, rtol=0, atol=TOL)
    assert done_seen >= 1
    assert env.action_space is not None and env.observation_space.shape == obs.shape or kind == "flexible"
    env.close()


def test_device_outputs_and_synthetic_source():
    from active_gym import AtariEnvArgs, AtariVecEnv
    args = AtariEnvArgs(game="boxing", seed=0, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0),
                        sensory_action_mode="absolute", resize_to_full=True, frame_source="synthetic", device="cuda")
    env = AtariVecEnv(args, 16, kind="fixed")
    obs, infos = env.reset()
    assert isinstance(obs, torch.Tensor) and obs.is_cuda and obs.shape == (16, 4, 84, 84)
    # SyncVectorEnv conventions: batched spaces, per-env spaces under single_*
    assert env.observation_space.shape == (16, 4, 84, 84) and env.single_observation_space.shape == (4, 84, 84)
    assert env.action_space["motor_action"].nvec.tolist() == [env.single_action_space["motor_action"].n] * 16
    assert env.action_space["sensory_action"].shape == (16,) and env.single_action_space["sensory_action"].shape == ()
    for _ in range(5):
        a = {"motor_action": np.random.randint(0, 4, 16), "sensory_action": torch.rand(16, 2, device="cuda") * 54}
        obs, r, d, t, infos = env.step(a)
    assert obs.is_cuda and float(obs.max()) <= 1.0 and float(obs.min()) >= 0.0 and float(obs.max()) > 0.0
    assert infos["fov_loc"].shape == (16, 2) and (infos["ep_len"] <= 5).all() and (infos["ep_len"] == 5).any()
    # device outputs: fov_loc stays on the device (int64, no synchronisation inside step); observations are double-buffered
    assert isinstance(infos["fov_loc"], torch.Tensor) and infos["fov_loc"].is_cuda and infos["fov_loc"].dtype == torch.int64
    a = {"motor_action": np.zeros(16, np.int64), "sensory_action": torch.full((16, 2), 11.0, device="cuda")}
    o1, _, d1, _, i1 = env.step(a)
    keep = o1.clone()
    o2, _, _, _, _ = env.step(a)
    assert o2.data_ptr() != o1.data_ptr() and torch.equal(o1, keep)           # the previous observation is still intact
    o3, _, _, _, _ = env.step(a)
    assert o3.data_ptr() == o1.data_ptr()                                      # ... until the step after next
    assert torch.equal(i1["fov_loc"][~torch.from_numpy(d1).cuda()], torch.full((int((~d1).sum()), 2), 11, device="cuda"))
    env.close()
    args.copy_obs = True
    env = AtariVecEnv(args, 4, kind="fixed")
    env.reset()
    a = {"motor_action": np.zeros(4, np.int64), "sensory_action": torch.zeros(4, 2, device="cuda")}
    k1 = env.step(a)[0]
    k2 = env.step(a)[0]
    k3 = env.step(a)[0]
    assert len({k1.data_ptr(), k2.data_ptr(), k3.data_ptr()}) == 3             # copy_obs: a fresh tensor per call
    env.close()


def test_recording_buffer_keys(tmp_path):
    import active_gym
    args = _args(fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute", resize_to_full=True, record=True)
    env = active_gym.AtariFixedFovealEnv(args)
    n = 0
    while n < 3:                                 # an episode of at least 3 steps (a first-step terminal records no `done`)
        env.reset()
        done = False
        n = 0
        while not done and n < 400:
            obs, r, done, _, info = env.step({"motor_action": 1, "sensory_action": np.array((5, 7))})
            n += 1
    env.reset()                                  # moves the finished episode to prev_record_buffer
    rw = env.env
    buf = rw.prev_record_buffer
    assert set(buf) >= {"rgb", "state", "action", "reward", "done", "truncated", "info", "return_reward", "fov_size", "fov_loc"}
    assert buf["rgb"][0].shape == (256, 256, 3) and len(buf["action"]) == n
    # the reference's RecordWrapper sits under the fovea wrapper: it records the FULL state, float64 (fov_env.py:59,74)
    assert buf["state"][0].shape == (4, 84, 84) and buf["state"][0].dtype == np.float64
    assert len(buf["state"]) == len(buf["rgb"]) == len(buf["fov_loc"]) == n          # n - 1 non-terminal steps + reset
    assert len(buf["done"]) == len(buf["info"]) == n and buf["done"][-1] is True and not any(buf["done"][:-1])
    assert np.array_equal(buf["fov_loc"][-1], (5, 7)) and buf["info"][-1]["ep_len"] == n
    path = str(tmp_path / "rec.pt")
    rw.save_record_to_file(path)
    saved = torch.load(path, weights_only=False)
    assert isinstance(saved["rgb"], str) and len(saved["reward"]) == n
    env.close()


@pytest.mark.parametrize("kind,factory", [("base", "AtariBaseEnv"), ("fixed", "AtariFixedFovealEnv"), ("flex", "AtariFlexibleFovealEnv"),
                                          ("per", "AtariFixedFovealPeripheralEnv")])
def test_record_buffers_match_reference_run_control_flow_pinned_resize_unpinned(kind, factory, tmp_path):
    """f2 as a parity row: the drop-in envs with record=True, through the real device path, against the record buffers the
    REFERENCE's own env stack produced over the same scripted emulator, actions and no-op draws.
    WHAT THIS PINS: the reference's control flow, buffer layout and key order, rewards / dones / infos, fov_loc / fov_res - all
    produced by the reference's own code.  WHAT IT DOES NOT: the OpenCV resize arithmetic - make_golden.py had to supply
    `cv2.resize` from the repo's own oracle restatement (cv2 is not in the image), so the `state` / `rgb` CRC checks below are
    "kernel == oracle restatement" through the reference's plumbing, i.e. PARITY UNPINNED (oracle resize) for those two arrays
    (DESIGN.md section 4's table lists the restated arithmetic).
    (tests/golden/record_atari_*.npz from tests/golden/make_golden.py::_record_atari_case: fov_env.py:34-37,51-102,152-154,
    253-256,370-373 on top of atari_env.py:73-169).  Contents, not only keys: every recorded full state (u8 numerators, CRC:
    bit-exact), every 256x256 frame (CRC), actions, cumulative / returned rewards, dones, truncated flags, infos incl.
    fov_loc / fov_res, key order, and the saved .pt."""
    import os
    import zlib
    import active_gym
    from active_gym import AtariEnvArgs
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"record_atari_{kind}.npz"))
    seed = int(g["seed"])
    args = AtariEnvArgs(game="scripted", seed=seed, obs_size=tuple(int(v) for v in g["obs_size"]), frame_stack=int(g["frame_stack"]),
                        action_repeat=int(g["action_repeat"]), record=True, fov_size=tuple(int(v) for v in g["fov_size"]),
                        fov_init_loc=tuple(int(v) for v in g["init_loc"]), sensory_action_mode="absolute", resize_to_full=True,
                        mask_out=False, peripheral_res=tuple(int(v) for v in g["peripheral_res"]), antialias=True,
                        frame_format="gray",               # the reference reads ale.getScreenGrayscale() (atari_env.py:74)
                        frame_source=lambda a, i: ScriptedALE(seed=seed, screen_hw=(210, 160), n_actions=4,
                                                              start_lives=int(g["start_lives"]), p_life=float(g["p_life"]),
                                                              p_over=float(g["p_over"])))
    env = getattr(active_gym, factory)(args)
    noops = iter(g["noops"].tolist())
    env.unwrapped._core.runner.noop_fn = lambda: int(next(noops))
    rw = env
    while not hasattr(rw, "prev_record_buffer"):
        rw = rw.env
    for is_reset, motor, s0, s1, typ in g["drive"].tolist():
        if is_reset:
            env.reset()
        elif kind == "base":
            env.step(motor)
        else:
            a = {"motor_action": motor, "sensory_action": np.array((s0, s1), np.int64 if kind == "flex" else np.float64)}
            if kind == "flex":
                a["sensory_action_type"] = np.array((typ,))
            env.step(a)

    def check(tag, buf):
        assert list(buf.keys()) == g[f"{tag}_keys"].tolist(), (tag, list(buf.keys()))
        for k in ("rgb", "state", "action", "reward", "done", "truncated", "info", "return_reward", "fov_loc", "fov_res"):
            if f"{tag}_len_{k}" in g.files:
                assert len(buf[k]) == int(g[f"{tag}_len_{k}"]), (tag, k, len(buf[k]))
        rgb = np.stack(buf["rgb"])
        assert rgb.dtype == np.uint8 and rgb.shape[1:] == (256, 256, 3)
        assert np.array_equal(rgb[0][::8, ::8], g[f"{tag}_rgb_first"])
        assert [zlib.crc32(np.ascontiguousarray(f).tobytes()) for f in rgb] == g[f"{tag}_rgb_crc"].tolist(), \
            "rgb frames (parity unpinned: the golden's cv2.resize is the oracle's restatement)"
        st = np.stack(buf["state"])
        assert st.dtype == np.float64                         # the reference records its float64 full state (fov_env.py:59,74)
        u8 = np.rint(st * 255.0).astype(np.uint8)
        assert np.array_equal((u8.astype(np.float32) / np.float32(255.0)).astype(np.float64), st)
        assert np.array_equal(u8[-1], g[f"{tag}_state_last_u8"])
        assert [zlib.crc32(np.ascontiguousarray(f).tobytes()) for f in u8] == g[f"{tag}_state_crc"].tolist(), \
            "recorded states (parity unpinned: the golden's cv2.resize is the oracle's restatement)"
        for k in ("action", "reward", "done", "truncated", "return_reward"):
            assert np.array_equal(np.array(buf[k]), g[f"{tag}_{k}"]), (tag, k, buf[k])
        for k in ("fov_loc", "fov_res", "fov_size", "peripheral_res"):
            if f"{tag}_{k}" in g.files:
                assert np.array_equal(np.array(buf[k], dtype=np.int64), g[f"{tag}_{k}"]), (tag, k)
        for k in ("raw_reward", "reward", "ep_len"):
            assert np.array_equal(np.array([i[k] for i in buf["info"]], dtype=np.float64), g[f"{tag}_info_{k}"]), (tag, k)
        assert list(buf["info"][-1].keys()) == g[f"{tag}_info_keys"].tolist()
        for k in ("fov_loc", "fov_res"):
            if f"{tag}_info_{k}" in g.files:
                assert np.array_equal(np.array([i[k] for i in buf["info"]], dtype=np.int64), g[f"{tag}_info_{k}"]), (tag, k)

    check("prev", rw.prev_record_buffer)
    check("cur", rw.record_buffer)
    path = str(tmp_path / f"{kind}.pt")
    rw.save_record_to_file(path)
    saved = torch.load(path, weights_only=False)              # written by the line above
    assert list(saved.keys()) == g["prev_keys"].tolist() and isinstance(saved["rgb"], str)
    assert np.array_equal(np.array(saved["reward"]), g["prev_reward"])
    assert saved["state"] == [0] * len(g["prev_reward"])     # the reference replaces the states by zeros on save (fov_env.py:100)
    env.close()


@pytest.mark.parametrize("kind", ["base", "fixed"])
def test_reset_does_not_overwrite_the_observation_the_caller_holds(kind):
    """Device outputs are double-buffered: an observation returned by step() stays valid until the step after next.  reset()
    and reset_envs() take the next buffer too - the terminal observation a caller still holds must not turn into the reset
    observation under its hands (and the reset observation must not alias it)."""
    from active_gym import AtariVecEnv
    N = 6
    args = _args(fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute", resize_to_full=True, device="cuda:0")
    env = AtariVecEnv(args, N, kind=kind, autoreset=False)
    env.reset()
    act = np.zeros(N, np.int64) if kind == "base" else {"motor_action": np.zeros(N, np.int64), "sensory_action": np.full((N, 2), 11.0)}
    env.step(act)
    held, *_ = env.step(act)
    snap = held.clone()
    robs, _ = env.reset_envs([1, 4])
    assert robs.data_ptr() != held.data_ptr()
    assert torch.equal(held, snap), "reset_envs() rewrote the observation returned by the previous step()"
    for i in (0, 2, 3, 5):                                   # envs that were not reset keep their observation
        assert torch.equal(robs[i], snap[i])
    assert not torch.equal(robs[1], snap[1])
    held2, *_ = env.step(act)
    snap2 = held2.clone()
    robs2, _ = env.reset()
    assert robs2.data_ptr() != held2.data_ptr() and torch.equal(held2, snap2)
    env.close()


def test_bench_contract_json():
    """bench.py prints ONE JSON line with the driver's contract fields (small, quick configuration)."""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--steps", "8", "--warmup", "2", "--envs", "64",
                          "--cpu-seconds", "3"], capture_output=True, text=True, timeout=600, cwd=repo)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert abs(d["value"] - 64 * 8 / (d["ms_per_step"] * 8e-3)) / d["value"] < 1e-6
    # the timed region is K plain steps: no HIP event is created for, handed to or recorded around any of its launches
    # (the per-kernel durations behind `roofline` come from a sampling pass that follows it)
    assert d["events_in_timed_region"] == 0 and d["preroll"] >= 0 and d["rehearsals"] >= 0
    assert r["launches_timed"] >= 16 and "sampling pass" in r["timing"]
    # round 4: the fovea kernel against HBM (its output rotating through more buffers than the Infinity Cache holds), the same step on
    # the runner's compact staging layout, and the e2e leg's placement / staging facts
    assert d["kernels_obs_pool"]["buffers"] >= 3 and d["kernels_obs_pool"]["k_fovea_fixed"]["avg_us"] > 0
    ci = d["compact_input"]
    assert ci["value"] > 0 and ci["k_ingest_avg_us"] > 0 and "configs[1]" in d["config"]["workload"]
    e = d["e2e"]
    assert "error" not in e, e
    assert e["rgb"]["rows_staged_per_screen"] == 168 and e["rgb"]["h2d_bytes_per_step"] == 64 * 2 * 168 * 160 * 3
    assert e["placement"]["workers"] >= 1 and e["step_loop"].startswith("native")


@pytest.mark.parametrize("kind", ["fixed", "flex", "per"])
def test_spaces_and_attributes_match_reference(kind):
    """tests/golden/spaces.json: what the reference's wrappers hand to gymnasium's space constructors and the
    attributes callers read (fov_env.py:110-147,236-251,358-367), incl. non-square, relative and degenerate cases."""
    import json, os
    import active_gym
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spaces.json")))
    geo = {"abs84": ((84, 84), (30, 30), "absolute", None, True, False), "rel": ((84, 84), (30, 30), "relative", (-10.0, 10.0), True, False),
           "rel_degenerate": ((84, 84), (30, 30), "relative", (3, 3), False, False)}        # Atari obs must be square
    factory = {"fixed": active_gym.AtariFixedFovealEnv, "flex": active_gym.AtariFlexibleFovealEnv,
               "per": active_gym.AtariFixedFovealPeripheralEnv}[kind]
    for tag, (obs, fov, mode, sas, rtf, mo) in geo.items():
        g = gold[f"{kind}_{tag}"]
        args = _args(obs_size=obs, fov_size=fov, fov_init_loc=(2.5, 3.5), sensory_action_mode=mode, sensory_action_space=sas,
                     resize_to_full=rtf, mask_out=mo, peripheral_res=(9, 7),
                     frame_source=lambda a, i: ScriptedALE(seed=1, n_actions=6))
        env = factory(args)
        sp = env.action_space
        assert sorted(sp.keys()) == sorted(g["action_space"].keys())
        assert sp["motor_action"].n == g["action_space"]["motor_action"]["n"]
        sa, gs = sp["sensory_action"], g["action_space"]["sensory_action"]
        assert float(np.min(sa.low)) == gs["low"] and float(np.max(sa.high)) == gs["high"] and np.dtype(sa.dtype).name == gs["dtype"]
        if kind == "flex":
            assert sp["sensory_action_type"].n == 2
        go = g["observation_space"]
        assert list(env.observation_space.shape) == go["shape"] and np.dtype(env.observation_space.dtype).name == go["dtype"]
        assert float(np.min(env.observation_space.low)) == go["low"] and float(np.max(env.observation_space.high)) == go["high"]
        assert np.asarray(env.sensory_action_space).tolist() == g["sensory_action_space"]
        env.reset()
        assert np.asarray(env.fov_loc).tolist() == g["fov_loc"] and list(env.fov_size) == g["fov_size"]
        assert bool(env.mask_out) == g["mask_out"]
        if kind == "flex":
            assert np.asarray(env.fov_res).tolist() == g["fov_res"]
        if kind == "per":
            assert list(env.peripheral_res) == g["peripheral_res"] and bool(env.resize_to_full) == g["resize_to_full"]
        env.close()


def test_config1_single_env_10k_steps_against_cpu_reference_path():
    """BASELINE.json configs[0] / SURVEY §8d config 1: one Breakout-shaped AtariFixedFovealEnv, 84x84 / 30x30,
    absolute integer sensory actions uniform in [0, 54]^2, 10 000 steps with resets - the drop-in single env against
    the CPU reference path (the oracle chain), every step: reward / done / counters / fov_loc exact, observation
    within 1e-5.  A soak for ring wrap-around, life-loss and full resets, and state drift."""
    import active_gym
    from lcg_ale import LcgALE
    mk = lambda: LcgALE(123, 4, 5, 6, 1)                      # 4 actions (Breakout's minimal set), 5 lives
    args = _args(game="breakout", seed=123, fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute",
                 resize_to_full=True, frame_source=lambda a, i: mk())
    env = active_gym.AtariFixedFovealEnv(args)
    noops = _Noops(9)
    env.unwrapped._core.runner.noop_fn = lambda: int(next(noops.a))
    ale = mk()
    base = O.AtariEnvOracle(ale, ale.getMinimalActionSet(), obs_size=(84, 84), frame_stack=4, action_repeat=4,
                            clip_reward=False, noop_fn=lambda: int(next(noops.b)), prefer_rgb=True)
    rec = O.RecordOracle(base)
    fov = O.FixedFovealOracle(obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute",
                              resize_to_full=True)
    rng = np.random.default_rng(2024)
    o, info = env.reset()
    s, oi = rec.reset()
    np.testing.assert_allclose(o, fov.reset(s), rtol=0, atol=TOL)
    episodes = life_resets = 0
    worst = 0.0
    for step in range(10000):
        act = {"motor_action": int(rng.integers(0, 4)), "sensory_action": rng.integers(0, 55, size=2)}
        o, r, d, tr, info = env.step(act)
        s, r2, d2, _, oi = rec.step(act["motor_action"])
        want = fov.step(s, act["sensory_action"])
        assert r == r2 and d == d2 and tr is False, step
        assert info["ep_len"] == oi["ep_len"] and info["reward"] == oi["reward"] and np.array_equal(info["fov_loc"], fov.fov_loc), step
        worst = max(worst, float(np.abs(o - want).max()))
        if d:
            episodes += 1
            life_resets += int(base.life_termination)
            o, info = env.reset()
            s, oi = rec.reset()
            worst = max(worst, float(np.abs(o - fov.reset(s)).max()))
            assert np.array_equal(info["fov_loc"], fov.fov_loc)
    assert worst <= TOL, worst
    assert episodes >= 20 and 0 < life_resets < episodes          # both reset paths were exercised
    env.close()


@pytest.mark.parametrize("factory,mode", [("AtariFixedFovealEnv", "absolute"), ("AtariFixedFovealEnv", "relative"),
                                          ("AtariFlexibleFovealEnv", "absolute"), ("AtariFixedFovealPeripheralEnv", "relative")])
def test_readme_quickstart_action_space_sample_steps(factory, mode):
    """`env.step(env.action_space.sample())` (reference README.md:40-50): the sensory_action Box is SCALAR, as the
    reference declares it (fov_env.py:125-129), so a sample is one number, which np.clip(loc, 0, obs - fov) broadcasts to
    (a, a) (fov_env.py:166-167,193-199).  Single envs and the batched env (one number per env)."""
    import active_gym
    from active_gym import AtariVecEnv
    kw = dict(fov_size=(30, 30), fov_init_loc=(10, 20), sensory_action_mode=mode, sensory_action_space=(-6.0, 6.0),
              resize_to_full=True, peripheral_res=(20, 20))
    env = getattr(active_gym, factory)(_args(**kw))
    env.action_space.seed(3)
    env.reset()
    loc = np.array([10, 20])
    for _ in range(8):
        a = env.action_space.sample()
        assert np.asarray(a["sensory_action"]).shape == ()
        if "sensory_action_type" in a:
            a["sensory_action_type"] = 0                                     # FOV_LOC: the rule below
        obs, r, d, t, info = env.step(a)
        s = float(a["sensory_action"])
        if mode == "absolute":
            loc = np.rint(np.clip(np.array([s, s]), 0, 54)).astype(int)
        else:
            loc = np.rint(np.clip(loc + np.rint(np.clip(np.array([s, s]), -6, 6)).astype(int), 0, 54)).astype(int)
        assert np.array_equal(info["fov_loc"], loc) and obs.shape == (4, 84, 84)
        if d:
            env.reset()
            loc = np.array([10, 20])
    env.close()
    if factory == "AtariFixedFovealEnv":
        N = 5
        venv = AtariVecEnv(_args(**kw), N, kind="fixed")
        venv.action_space.seed(5)
        venv.reset()
        a = venv.action_space.sample()
        assert np.asarray(a["sensory_action"]).shape == (N,)
        _, _, done, _, info = venv.step(a)
        s = np.asarray(a["sensory_action"], dtype=np.float64)[:, None].repeat(2, 1)
        want = (np.rint(np.clip(s, 0, 54)) if mode == "absolute" else
                np.rint(np.clip(np.array([10, 20]) + np.rint(np.clip(s, -6, 6)), 0, 54))).astype(int)
        want[done] = (10, 20)                                                   # autoreset re-initialises fov_loc (fov_env.py:156-160)
        assert np.array_equal(info["fov_loc"], want)
        venv.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible GPUs")
def test_vec_env_on_a_device_that_is_not_the_current_one():
    """copy-done events must be recorded on the pipeline's stream, not on the current device's current
    stream.  With kind='base' and device outputs nothing else synchronises the pinned staging buffer."""
    from active_gym import AtariEnvArgs, AtariVecEnv
    torch.cuda.set_device(0)
    args = AtariEnvArgs(game="boxing", seed=0, obs_size=(84, 84), frame_source="native", device="cuda:1")
    ref_args = AtariEnvArgs(game="boxing", seed=0, obs_size=(84, 84), frame_source="native", device="cuda:0")
    a, b = AtariVecEnv(args, 8, kind="base", noop_per_env=True), AtariVecEnv(ref_args, 8, kind="base", noop_per_env=True)
    oa, _ = a.reset()
    ob, _ = b.reset()
    assert oa.device.index == 1 and torch.equal(oa.cpu(), ob.cpu())
    for _ in range(20):
        m = np.random.randint(0, 4, 8)
        oa = a.step(m)[0]
        ob = b.step(m)[0]
        assert torch.equal(oa.cpu(), ob.cpu())
    a.close()
    b.close()


def test_native_loop_gathers_terminal_rows_of_any_size():
    """A raw-crop context with an odd fov size has observation rows that are not a multiple of 16 bytes (3 x 5 x 7 floats here): the
    loop's gather of the terminal observations must not assume float4 rows.  Native loop == Python loop, every terminal observation."""
    from active_gym import AtariEnvArgs, AtariVecEnv
    N = 40
    kw = dict(game="g", seed=3, obs_size=(84, 84), frame_stack=3, fov_size=(5, 7), fov_init_loc=(1, 2), sensory_action_mode="absolute",
              resize_to_full=False, mask_out=False, frame_source="native", device="cuda:0", num_workers=2, scripted_actions=4,
              scripted_lives=1, scripted_p_life=0, scripted_p_over=150)
    a = AtariVecEnv(AtariEnvArgs(native_loop=True, **kw), N, kind="fixed", noop_fn=lambda: 2)
    b = AtariVecEnv(AtariEnvArgs(native_loop=False, **kw), N, kind="fixed", noop_fn=lambda: 2)
    assert a._loop is not None and b._loop is None and tuple(a.reset()[0].shape) == (N, 3, 5, 7)
    b.reset()
    rng = np.random.default_rng(0)
    seen = 0
    for step in range(12):
        act = {"motor_action": rng.integers(0, 4, N), "sensory_action": rng.uniform(-5, 90, (N, 2)).astype(np.float32)}
        ra, rb = a.step(act), b.step(act)
        assert torch.equal(ra[0], rb[0]) and np.array_equal(ra[2], rb[2])
        for i in np.nonzero(ra[2])[0]:
            assert torch.equal(ra[4]["final_observation"][i], rb[4]["final_observation"][i]), (step, i)
            seen += 1
    assert seen >= 20
    a.close()
    b.close()


@pytest.mark.parametrize("kind,fmt,N,STEPS", [("fixed", "rgb", 700, 14), ("flexible", "gray", 700, 14), ("base", "gray", 700, 14),
                                              ("fixed", "rgb", 4096, 4)])
def test_native_loop_equals_python_loop_at_scale(kind, fmt, N, STEPS):
    """The native step loop (agx_loop_step) against the Python loop of vector.py, same native runner, same seeds, at a batch large
    enough for what the N = 5 oracle chains never reach: several hundred envs ending an episode in ONE step (the gathers and the
    scatter run with grid.y = k in the hundreds), more done envs than a 256-env scan block, every step with resets.  Everything the
    step returns must be identical: observations, rewards, terminals, counters, fov state, every terminal observation / info.
    N = 4096: four times the metric's batch - 330 MB of compact screens per step, ~ 2,000 episode ends in one step."""
    import zlib
    from active_gym import AtariEnvArgs, AtariVecEnv
    kw = dict(game="g", seed=11, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(2, 3), sensory_action_mode="absolute",
              resize_to_full=True, frame_source="native", frame_format=fmt, device="cuda:0", num_workers=4 if N < 1000 else 12,
              scripted_actions=4, scripted_lives=2, scripted_p_life=250, scripted_p_over=60)      # ~ half of the envs end per step
    envs = []
    for loop in (True, False):
        import random
        rnd = random.Random(5)
        e = AtariVecEnv(AtariEnvArgs(native_loop=loop, **kw), N, kind=kind, noop_fn=lambda r=rnd: r.randrange(30))
        assert (e._loop is not None) == loop
        envs.append(e)
    a, b = envs

    def crc(t):
        return zlib.crc32(t.detach().cpu().numpy().tobytes())

    oa, _ = a.reset()
    ob, _ = b.reset()
    assert crc(oa) == crc(ob)
    rng = np.random.default_rng(1)
    most = 0
    for step in range(STEPS):
        motor = rng.integers(0, 4, N)
        if kind == "base":
            act = motor
        else:
            types = rng.integers(0, 2, N)
            sens = np.where(types[:, None] == 1, rng.integers(8, 70, (N, 2)), rng.integers(-5, 60, (N, 2))).astype(np.int64)
            act = {"motor_action": motor, "sensory_action": sens}
            if kind == "flexible":
                act["sensory_action_type"] = types
        ra, rb = a.step(act), b.step(act)
        assert crc(ra[0]) == crc(rb[0]), step
        assert np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2]) and np.array_equal(ra[3], rb[3]), step
        ia, ib = ra[4], rb[4]
        assert set(ia) == set(ib), (step, sorted(ia), sorted(ib))
        for key in ("raw_reward", "reward", "ep_len"):
            assert np.array_equal(ia[key], ib[key]), (step, key)
        for key in ("fov_loc", "fov_res"):
            if key in ia:
                assert torch.equal(ia[key], ib[key]), (step, key)
        done = ra[2]
        most = max(most, int(done.sum()))
        if done.any():
            assert np.array_equal(ia["_final_observation"], ib["_final_observation"])
            for i in np.nonzero(done)[0]:
                assert crc(ia["final_observation"][i]) == crc(ib["final_observation"][i]), (step, i)
                fa, fb = ia["final_info"][i], ib["final_info"][i]
                assert set(fa) == set(fb)
                for key in fa:
                    va, vb = fa[key], fb[key]
                    assert (torch.equal(va, vb) if isinstance(va, torch.Tensor) else va == vb), (step, i, key)
    assert most > 256, most
    assert torch.equal(a.pipe.stack_u8(), b.pipe.stack_u8())
    a.close()
    b.close()


def test_native_loop_binding_keeps_no_per_step_objects():
    """A torch tensor made from __cuda_array_interface__ keeps its source object alive for good, so the loop's binding must not make
    one per step (it did: 0.34 KB of host memory per step with episode ends, found by tools/soak.py): after 300 steps with resets in
    every step there is ONE view object per loop-owned buffer, and their number does not move over another 300."""
    import gc
    from active_gym import AtariEnvArgs, AtariVecEnv
    from active_gym.native_loop import _DeviceView
    N = 64
    env = AtariVecEnv(AtariEnvArgs(game="g", seed=2, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute",
                                   resize_to_full=True, frame_source="native", frame_format="gray", device="cuda:0", num_workers=2,
                                   scripted_lives=2, scripted_p_life=100, scripted_p_over=30), N, kind="flexible")
    assert env._loop is not None
    env.reset()
    act = {"motor_action": np.zeros(N, np.int64), "sensory_action": np.full((N, 2), 20, np.int64), "sensory_action_type": np.zeros(N, np.int64)}

    def live():
        gc.collect()
        return sum(type(o) is _DeviceView for o in gc.get_objects())

    ends = 0
    for _ in range(300):
        ends += int(env.step(act)[2].sum())
    first = live()
    for _ in range(300):
        ends += int(env.step(act)[2].sum())
    assert ends > 600
    assert live() == first and 1 <= first <= 3, (first, live())
    env.close()


@pytest.mark.parametrize("native_loop", [True, False])
def test_two_envs_stepped_from_two_threads(native_loop):
    """Two vector envs of one process stepped CONCURRENTLY from two Python threads (actor threads; ctypes releases the GIL inside the C
    calls, the no-op callback takes it back): each must return exactly what it returns when it runs alone - nothing in the libraries
    is shared between contexts but the device.  Alone they run on the default stream, together each on a stream of its own thread:
    everything an env enqueues (copy stream, event edges, kernels, torch's own work) must be ordered against the CALLER's stream."""
    import threading
    import zlib
    from active_gym import AtariEnvArgs, AtariVecEnv
    N, STEPS = 96, 40

    def make(seed, kind, fmt):
        import random
        rnd = random.Random(seed)
        return AtariVecEnv(AtariEnvArgs(game="g", seed=seed, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(1, 2), sensory_action_mode="absolute",
                                        resize_to_full=True, frame_source="native", frame_format=fmt, device="cuda:0", num_workers=3,
                                        native_loop=native_loop, scripted_actions=4, scripted_lives=2, scripted_p_life=80, scripted_p_over=20),
                           N, kind=kind, noop_fn=lambda r=rnd: r.randrange(30))

    def run(env, seed, out, barrier=None):
        if barrier is None:
            return body(env, seed, out, None)
        with torch.cuda.stream(torch.cuda.Stream(device="cuda:0")):      # every actor thread on a stream of its own
            body(env, seed, out, barrier)

    def body(env, seed, out, barrier):
        try:
            rng = np.random.default_rng(seed)
            if barrier is not None:
                barrier.wait()
            o, _ = env.reset()
            log = [zlib.crc32(o.cpu().numpy().tobytes())]
            for _ in range(STEPS):
                act = {"motor_action": rng.integers(0, 4, N), "sensory_action": rng.integers(-5, 60, (N, 2)).astype(np.int64)}
                if env.kind == "flexible":
                    act["sensory_action_type"] = rng.integers(0, 2, N)
                o, r, d, _, info = env.step(act)
                fin = 0
                if d.any():
                    for i in np.nonzero(d)[0]:
                        fin = zlib.crc32(info["final_observation"][i].cpu().numpy().tobytes(), fin)
                log.append((zlib.crc32(o.cpu().numpy().tobytes()), float(r.sum()), int(d.sum()), fin))
            out.append(log)
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            out.append(e)

    specs = [(3, "fixed", "rgb"), (4, "flexible", "gray")]
    alone = []
    for seed, kind, fmt in specs:
        env = make(seed, kind, fmt)
        run(env, seed, alone)
        env.close()
    envs = [make(*sp) for sp in specs]
    outs = [[], []]
    bar = threading.Barrier(2)
    th = [threading.Thread(target=run, args=(envs[i], specs[i][0], outs[i], bar)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    for e in envs:
        e.close()
    for i in range(2):
        assert len(outs[i]) == 1 and not isinstance(outs[i][0], BaseException), outs[i]
        assert outs[i][0] == alone[i], i
    assert sum(x[2] for x in alone[0][1:]) > 50


def test_loop_abi_error_paths():
    """include/agx_loop.h: every misuse is an error code and a message, never a crash - null arguments, a config of the wrong size,
    a host source without its callbacks, a fovea context stepped without its fov_loc buffer, env indices out of range, a callback
    that fails (the step reports it and the loop stays usable)."""
    import ctypes as C
    from active_gym import AtariEnvArgs, AtariVecEnv, _native as nat
    from active_gym import native_loop as NL
    N = 12
    env = AtariVecEnv(AtariEnvArgs(game="g", seed=2, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute",
                                   resize_to_full=True, frame_source="native", frame_format="gray", device="cuda:0", num_workers=2,
                                   scripted_lives=2, scripted_p_life=100, scripted_p_over=30), N, kind="fixed")
    lp = env._loop
    lib = lp._lib
    P = C.c_void_p
    src = NL.AgxHostSource()
    src.self = env.runner._h
    src.step = C.cast(env.runner._lib.agxr_step, P)
    src.reset_packed = C.cast(env.runner._lib.agxr_reset_packed, P)
    good = NL.AgxLoopConfig(C.sizeof(NL.AgxLoopConfig), 1, 1, 1)
    out = P()
    assert lib.agx_loop_create(None, C.byref(src), C.byref(good), C.byref(out)) == nat.E_INVALID and not out.value
    assert lib.agx_loop_create(env.pipe._ctx, None, C.byref(good), C.byref(out)) == nat.E_INVALID
    bad = NL.AgxLoopConfig(8, 1, 1, 1)
    assert lib.agx_loop_create(env.pipe._ctx, C.byref(src), C.byref(bad), C.byref(out)) == nat.E_INVALID and not out.value
    assert b"struct_size" in lib.agx_loop_last_error(None)
    nosrc = NL.AgxHostSource()
    assert lib.agx_loop_create(env.pipe._ctx, C.byref(nosrc), C.byref(good), C.byref(out)) == nat.E_INVALID
    assert b"step and reset_packed" in lib.agx_loop_last_error(None)
    env.reset()
    motor = np.zeros(N, np.int32)
    res = NL.AgxLoopResult()
    obs, loc = env._obs, env._loc
    st = lp._stream()
    assert lib.agx_loop_step(None, motor.ctypes.data, None, 0, None, P(obs.data_ptr()), P(loc.data_ptr()), None, C.byref(res), st) == nat.E_INVALID
    assert lib.agx_loop_step(lp._h, None, None, 0, None, P(obs.data_ptr()), P(loc.data_ptr()), None, C.byref(res), st) == nat.E_INVALID
    assert lib.agx_loop_step(lp._h, motor.ctypes.data, None, 0, None, None, P(loc.data_ptr()), None, C.byref(res), st) == nat.E_INVALID
    assert lib.agx_loop_step(lp._h, motor.ctypes.data, None, 0, None, P(obs.data_ptr()), None, None, C.byref(res), st) == nat.E_INVALID
    assert b"d_fov_loc" in lib.agx_loop_last_error(lp._h)
    idx = np.array([0, N], np.int32)
    assert lib.agx_loop_reset_envs(lp._h, idx.ctypes.data, 2, None, P(obs.data_ptr()), P(loc.data_ptr()), None, st) == nat.E_INVALID
    assert b"out of range" in lib.agx_loop_last_error(lp._h)
    assert lib.agx_loop_reset_envs(lp._h, idx.ctypes.data, 0, None, P(obs.data_ptr()), P(loc.data_ptr()), None, st) == nat.OK
    assert lib.agx_loop_reset(lp._h, None, None, None, None, st) == nat.E_INVALID
    # a motor action outside the action set: the runner's step callback refuses it, the loop reports the host source's reason
    motor[3] = 99
    with pytest.raises(nat.AgxError, match="host source"):
        lp.step(motor, None, 0, None, obs, loc, None)
    motor[3] = 0
    act = {"motor_action": np.zeros(N, np.int64), "sensory_action": np.full((N, 2), 20, np.int64)}
    for _ in range(10):                                              # ... and keeps working
        o, r, d, _, info = env.step(act)
    assert np.isfinite(o.cpu().numpy()).all()
    assert lib.agx_loop_destroy(None) == nat.OK
    env.close()


def test_native_loop_teardown_in_any_order():
    """The step loop holds the raw handles of its context and of its runner.  A child process closes the three in every order,
    abandons an env to the garbage collector (reference cycle through the no-op callback: finalizers in an order of the collector's
    choosing) and leaves another alive at interpreter exit: no crash, and a closed env raises instead of stepping."""
    import os
    import subprocess
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "active-gym_amd")
    code = r"""
import gc, itertools, sys
sys.path[:0] = [%r]
import numpy as np, torch
from active_gym import AtariEnvArgs, AtariVecEnv
N = 24
def make():
    e = AtariVecEnv(AtariEnvArgs(game="g", seed=2, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute",
                                 resize_to_full=True, frame_source="native", frame_format="gray", device="cuda:0", num_workers=2,
                                 scripted_lives=2, scripted_p_life=100, scripted_p_over=30), N, kind="fixed")
    assert e._loop is not None
    e.reset()
    for _ in range(5):
        e.step({"motor_action": np.zeros(N, np.int64), "sensory_action": np.full((N, 2), 20, np.int64)})
    return e
for order in itertools.permutations(("loop", "pipe", "runner")):
    e = make()
    parts = {"loop": e._loop, "pipe": e.pipe, "runner": e.runner}
    for name in order:
        parts[name].close()
    e.close()
    try:
        e.step({"motor_action": np.zeros(N, np.int64), "sensory_action": np.full((N, 2), 20, np.int64)})
        raise SystemExit("a closed env stepped")
    except SystemExit:
        raise
    except Exception:
        pass
for _ in range(3):
    e = make()
    del e
    gc.collect()
keep = make()
print("teardown ok", flush=True)
""" % pkg
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "teardown ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])


def test_c_loop_demo_matches_python_env(tmp_path):
    """examples/c_loop_demo.cpp drives the WHOLE vector step from plain C++ - libagx_runner.so's emulators (compact staging) into
    libagx.so's native step loop, autoreset inside, no Python, no torch; the same envs through AtariVecEnv must give the same
    observation / terminal-observation checksums, reward sums and done counts at every step."""
    import json
    import os
    import shutil
    import subprocess
    from active_gym import AtariEnvArgs, AtariVecEnv
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not on this box")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_loop_demo")
    libdir = os.path.join(repo, "active-gym_amd", "lib")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-I", os.path.join(repo, "include"), os.path.join(repo, "examples", "c_loop_demo.cpp"),
                    "-o", exe, "-L", libdir, "-lagx", "-lagx_runner", f"-Wl,-rpath,{libdir}"], check=True, timeout=600)
    N, STEPS = 48, 40
    out = subprocess.run([exe, str(N), str(STEPS)], check=True, capture_output=True, text=True, timeout=300).stdout
    got = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(got) == STEPS + 1

    def fnv(t):
        h = 1469598103934665603
        for w in t.detach().cpu().contiguous().numpy().view(np.uint32).reshape(-1).tolist():
            h = ((h ^ w) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h

    k = [0]

    def noop():
        v = (k[0] * 7 + 3) % 30
        k[0] += 1
        return v

    args = AtariEnvArgs(game="g", seed=21, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute",
                        resize_to_full=True, frame_source="native", device="cuda:0", num_workers=4, scripted_actions=4,
                        scripted_lives=2, scripted_p_life=60, scripted_p_over=20)
    env = AtariVecEnv(args, N, kind="fixed", noop_fn=noop)
    assert env._loop is not None and env._compact
    obs, _ = env.reset()
    assert fnv(obs) == got[0]["obs"]
    n_done = 0
    for t in range(STEPS):
        motor = np.array([(t + i) % 4 for i in range(N)])
        sens = np.array([[(t * 5 + i * 3) % 60 - 2.5, (t * 11 + i) % 64 - 4.0] for i in range(N)], np.float32)
        obs, rew, term, trunc, infos = env.step({"motor_action": motor, "sensory_action": sens})
        g = got[t + 1]
        assert g["step"] == t and g["n_done"] == int(term.sum()) and g["reward"] == float(rew.sum()), (t, g)
        assert fnv(obs) == g["obs"], t
        if term.any():
            fin = torch.stack([infos["final_observation"][i] for i in np.nonzero(term)[0]])
            assert fnv(fin) == g["final"], t
            n_done += int(term.sum())
    assert n_done >= 10
    env.close()


def test_host_outputs_through_pinned_buffers_equal_fresh_copies():
    """args.copy_obs = False with host (NumPy) outputs: observations come back as views of two pinned host buffers used alternately
    (no device-to-pageable copy per step): the same values as the default fresh arrays, and an observation survives exactly one
    further step."""
    from active_gym import AtariEnvArgs, AtariVecEnv
    N = 6
    kw = dict(game="g", seed=9, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute", resize_to_full=True,
              frame_source="native", scripted_actions=4, scripted_lives=2, scripted_p_life=80, scripted_p_over=20, num_workers=2)
    a = AtariVecEnv(AtariEnvArgs(**kw), N, kind="fixed", noop_fn=lambda: 1)
    b = AtariVecEnv(AtariEnvArgs(copy_obs=False, **kw), N, kind="fixed", noop_fn=lambda: 1)
    oa, ob = a.reset()[0], b.reset()[0]
    assert isinstance(ob, np.ndarray) and np.array_equal(oa, ob)
    rng = np.random.default_rng(2)
    held = None
    for step in range(10):
        act = {"motor_action": rng.integers(0, 4, N), "sensory_action": rng.uniform(-5, 60, (N, 2))}
        ra, rb = a.step(act), b.step(act)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2]), step
        if held is not None:
            assert np.array_equal(held[0], held[1]), "an observation stays valid through the following step"
        held = (rb[0], ra[0].copy())
        for i in np.nonzero(ra[2])[0]:
            assert np.array_equal(ra[4]["final_observation"][i], rb[4]["final_observation"][i])
    a.close()
    b.close()


def test_default_host_outputs_are_fresh_arrays_from_recycled_pinned_buffers():
    """Default host (NumPy) outputs: every call hands out an array of its own, as the reference's envs do (atari_env.py:143) - kept
    arrays never change, however many the caller keeps (beyond `host_obs_buffers` they are ordinary pageable arrays) - and a
    caller that drops its observations gets the pinned buffers back (the same few addresses recur).  Same values as with the pool
    switched off."""
    import gc
    from active_gym import AtariEnvArgs, AtariVecEnv
    N = 12                                      # 12 x 4 x 84 x 84 floats: above the pool's 1 MB threshold (smaller batches are copied pageable)
    kw = dict(game="g", seed=9, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute", resize_to_full=True,
              frame_source="native", scripted_actions=4, scripted_lives=2, scripted_p_life=80, scripted_p_over=20, num_workers=2)
    a = AtariVecEnv(AtariEnvArgs(host_obs_buffers=0, **kw), N, kind="fixed", noop_fn=lambda: 1)       # pageable copy per call
    b = AtariVecEnv(AtariEnvArgs(host_obs_buffers=3, **kw), N, kind="fixed", noop_fn=lambda: 1)
    assert a._host_pool is None and b._host_pool is not None
    oa, ob = a.reset()[0], b.reset()[0]
    assert isinstance(ob, np.ndarray) and ob.dtype == np.float32 and np.array_equal(oa, ob)
    rng = np.random.default_rng(2)
    kept = [(ob, ob.copy())]
    for step in range(9):                       # keep EVERYTHING: three pinned buffers, then pageable arrays
        act = {"motor_action": rng.integers(0, 4, N), "sensory_action": rng.uniform(-5, 60, (N, 2))}
        ra, rb = a.step(act), b.step(act)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2]), step
        kept.append((rb[0], rb[0].copy()))
        for i in np.nonzero(ra[2])[0]:
            assert np.array_equal(ra[4]["final_observation"][i], rb[4]["final_observation"][i])
    assert len({k[0].ctypes.data for k in kept}) == len(kept), "every kept observation has memory of its own"
    for k, (arr, copy) in enumerate(kept):
        assert np.array_equal(arr, copy), f"observation {k} changed while the caller held it"
    assert b._host_pool._made == 3
    # a view keeps its buffer out of the pool; dropping everything returns the three pinned buffers, and they are what recurs
    row = kept[1][0][2]
    del kept, ra, rb, oa, ob, arr, copy
    gc.collect()
    assert len(b._host_pool._free) == 2
    want_row = row.copy()
    seen = set()
    for step in range(8):
        act = {"motor_action": rng.integers(0, 4, N), "sensory_action": rng.uniform(-5, 60, (N, 2))}
        a.step(act)
        o = b.step(act)[0]
        seen.add(o.ctypes.data)
        del o
    assert len(seen) <= 2 and b._host_pool._made == 3
    assert np.array_equal(row, want_row)
    del row
    gc.collect()
    assert len(b._host_pool._free) == 3
    a.close()
    b.close()


@pytest.mark.parametrize("fmt", ["rgb", "gray"])
def test_native_loop_running_ahead_of_the_device_matches_a_synchronous_run(fmt):
    """The step loop never synchronises the device: the host runs up to two steps ahead, emulators overwrite one pinned staging set
    while the copy of the other is in flight, reset uploads slot in between step uploads on the copy stream.  A wrong event edge would
    show as a stale or half-written screen in SOME step of a long asynchronous run.  150 steps at N = 512 with resets in every step, no
    synchronisation inside the loop (observations are cloned on the stream), against the same envs stepped by the Python loop with a
    synchronisation after every step."""
    import zlib
    from active_gym import AtariEnvArgs, AtariVecEnv
    N, STEPS = 512, 150
    kw = dict(game="g", seed=4, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute",
              resize_to_full=False, mask_out=True, frame_source="native", frame_format=fmt, device="cuda:0", num_workers=6,
              scripted_actions=4, scripted_lives=3, scripted_p_life=20, scripted_p_over=4)
    rng = np.random.default_rng(8)
    motors = rng.integers(0, 4, (STEPS, N))
    sens = torch.from_numpy(rng.uniform(-5, 60, (STEPS, N, 2)).astype(np.float32)).to("cuda:0")

    def run(loop, sync):
        import random
        rnd = random.Random(17)
        env = AtariVecEnv(AtariEnvArgs(native_loop=loop, **kw), N, kind="fixed", noop_fn=lambda: rnd.randrange(30))
        assert (env._loop is not None) == loop
        obs0, _ = env.reset()
        outs, dones, fins = [obs0.clone()], [], []
        for t in range(STEPS):
            o, r, d, tr, info = env.step({"motor_action": motors[t], "sensory_action": sens[t]})
            outs.append(o.clone())                  # device-side copy in stream order: no host synchronisation
            dones.append(d.copy())
            fins.append([info["final_observation"][i] for i in np.nonzero(d)[0]] if d.any() else [])
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        crc = [zlib.crc32(o.cpu().numpy().tobytes()) for o in outs]
        fcrc = [[zlib.crc32(f.cpu().numpy().tobytes()) for f in fl] for fl in fins]
        env.close()
        return crc, np.array(dones), fcrc

    a = run(True, False)
    b = run(False, True)
    assert np.array_equal(a[1], b[1])
    bad = [t for t in range(STEPS + 1) if a[0][t] != b[0][t]]
    assert not bad, f"observations differ at steps {bad[:10]}"
    assert a[2] == b[2]
    assert a[1].any(axis=1).mean() > 0.9, "resets should occur in nearly every step"

"""Randomised env-level parity fuzz on the GPU: AtariVecEnv (Python or native runner, RGB or gray screens, chunked or
not, random action_repeat / frame_stack / clip / train-eval / kind / modes) against the oracle's per-env chain
FovealOracle(RecordOracle(AtariEnvOracle(LcgALE))) with autoreset.   python tools/fuzz_env.py [seconds] [seed]"""
import os, sys, time, traceback
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), REPO, os.path.join(REPO, "tests")]
import numpy as np
import torch
from active_gym import AtariEnvArgs, AtariVecEnv
from lcg_ale import LcgALE
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
TOL = 1e-5


def one_case(k):
    kind = ["fixed", "flexible", "peripheral", "base"][k % 4]
    N = int(rng.integers(1, 6))
    ar, fs = int(rng.integers(1, 7)), int(rng.integers(1, 6))
    clip, training = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    native = bool(rng.integers(0, 2))
    gray = bool(rng.integers(0, 2))
    mode = "absolute" if rng.random() < 0.6 else "relative"
    out = ["resize", "mask", "raw"][int(rng.integers(0, 3))]
    n_act = int(rng.integers(2, 7))
    lives, p_life, p_over = int(rng.integers(1, 5)), int(rng.integers(0, 120)), int(rng.integers(0, 40))
    s0 = int(rng.integers(0, 10000))
    noop_seq = rng.integers(0, 30, 4000).tolist()
    it_a, it_b = iter(noop_seq), iter(noop_seq)
    # round 4: device outputs (the native step loop when the native runner feeds them, unless switched off), compact / whole-screen
    # staging
    on_device = bool(rng.integers(0, 2))
    native_loop = bool(rng.integers(0, 4))
    compact = bool(rng.integers(0, 4))
    chunk = int(rng.integers(0, 3))
    cfg = dict(kind=kind, N=N, ar=ar, fs=fs, clip=clip, training=training, native=native, gray=gray, mode=mode, out=out,
               n_act=n_act, lives=lives, p_life=p_life, p_over=p_over, seed=s0, on_device=on_device, native_loop=native_loop,
               compact=compact, chunk=chunk)
    kw = dict(fov_size=(30, 30), fov_init_loc=(3.5, 4.49), sensory_action_mode=mode, sensory_action_space=(-9.0, 11.0),
              resize_to_full=(out == "resize"), mask_out=(out == "mask"), peripheral_res=(20, 20))
    src = "native" if native else (lambda a, i: LcgALE(a.seed + i, n_act, lives, p_life, p_over))
    args = AtariEnvArgs(game="g", seed=s0, obs_size=(84, 84), frame_stack=fs, action_repeat=ar, clip_reward=clip,
                        frame_source=src, frame_format="gray" if gray else "rgb", h2d_chunk_envs=chunk,
                        device="cuda:0" if on_device else None, native_loop=native_loop, compact_rows=compact,
                        scripted_actions=n_act, scripted_lives=lives, scripted_p_life=p_life, scripted_p_over=p_over, **kw)
    env = AtariVecEnv(args, N, kind=kind, noop_fn=lambda: int(next(it_a)))
    if not on_device and rng.random() < 0.5:
        # host outputs through the recycled pinned pool whatever the batch size (the product uses it from 1 MB of observations up)
        env._HOST_POOL_MIN_ELEMS = 0
        cfg["host_pool"] = True
    if not training:
        env.eval()
    chains = []
    for i in range(N):
        ale = LcgALE(s0 + i, n_act, lives, p_life, p_over)
        base = O.AtariEnvOracle(ale, ale.getMinimalActionSet(), obs_size=(84, 84), frame_stack=fs, action_repeat=ar,
                                clip_reward=clip, noop_fn=lambda: int(next(it_b)), prefer_rgb=not gray)
        base.training = training
        okw = dict(obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(3.5, 4.49), sensory_action_mode=mode,
                   sensory_action_space=(-9.0, 11.0))
        fov = {"fixed": lambda: O.FixedFovealOracle(resize_to_full=kw["resize_to_full"], mask_out=kw["mask_out"], **okw),
               "flexible": lambda: O.FlexibleFovealOracle(resize_to_full=kw["resize_to_full"], mask_out=kw["mask_out"], **okw),
               "peripheral": lambda: O.PeripheralOracle(peripheral_res=(20, 20), **okw), "base": lambda: None}[kind]()
        chains.append((O.RecordOracle(base), fov))

    def view(i, s, a=None, t=0, reset=False):
        fov = chains[i][1]
        if fov is None:
            return s
        if reset:
            return fov.reset(s)
        return fov.step(s, a, np.array((t,))) if kind == "flexible" else fov.step(s, a)

    def npy(x):
        return x.cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)

    def cmp(i, got, want, what):
        got = npy(got)
        if kind == "flexible" and out == "raw":
            rh, rw = chains[i][1].fov_res
            got = got[:, :rh, :rw]
        assert got.shape == want.shape, (cfg, what, got.shape, want.shape)
        err = float(np.abs(got - want).max())
        assert err <= TOL, (cfg, what, i, err)

    obs, infos = env.reset()
    for i in range(N):                                          # same env order as the runner draws its no-ops
        s, _ = chains[i][0].reset()
        cmp(i, obs[i], view(i, s, reset=True), "reset")
    for step in range(int(rng.integers(10, 40))):
        motor = rng.integers(0, n_act, N)
        types = rng.integers(0, 2, N)
        sens = rng.uniform(-14, 14, (N, 2)) if mode == "relative" else rng.uniform(-5, 60, (N, 2))
        if kind == "flexible":
            sens = np.where(types[:, None] == 1, rng.integers(8, 84, (N, 2)), np.rint(sens)).astype(np.int64)
        act = motor if kind == "base" else {"motor_action": motor, "sensory_action": sens}
        if kind == "flexible":
            act["sensory_action_type"] = types
        held = (obs, np.array(obs, copy=True)) if cfg.get("host_pool") and isinstance(obs, np.ndarray) else None
        obs, rew, term, trunc, infos = env.step(act)
        if held is not None:                     # an observation the caller still holds is never overwritten by a later step
            assert np.array_equal(held[0], held[1]), (cfg, step, "a held observation changed")
        for key in ("fov_loc", "fov_res"):
            if key in infos:
                infos[key] = npy(infos[key])
        for i in range(N):
            rec, fov = chains[i]
            s, r, d, tr, info = rec.step(int(motor[i]))
            w = view(i, s, sens[i], int(types[i]))
            assert float(rew[i]) == float(r) and bool(term[i]) == bool(d), (cfg, step, i, rew[i], r, term[i], d)
            if d:
                cmp(i, infos["final_observation"][i], w, "final")
                assert infos["final_info"][i]["ep_len"] == info["ep_len"] and infos["final_info"][i]["reward"] == info["reward"], cfg
                s, info = rec.reset()
                w = view(i, s, reset=True)
            cmp(i, obs[i], w, ("step", step))
            assert infos["ep_len"][i] == info["ep_len"] and infos["reward"][i] == info["reward"], (cfg, step, i)
            if fov is not None:
                assert np.array_equal(infos["fov_loc"][i], fov.fov_loc), (cfg, step, i)
                if kind == "flexible":
                    assert np.array_equal(infos["fov_res"][i], fov.fov_res), (cfg, step, i)
    env.close()
    return cfg


t0 = time.time()
n = 0
while time.time() - t0 < budget:
    try:
        one_case(n)
    except Exception:                                           # noqa: BLE001
        traceback.print_exc()
        print("ENV FUZZ FAILURE after", n, "cases, seed", seed, flush=True)
        sys.exit(1)
    n += 1
    if n % 50 == 0:
        print(f"{n} env cases ok, {time.time() - t0:.0f}s", flush=True)
print(f"env fuzz ok: {n} cases, seed {seed}")

"""GPU: the DMC pixel front end (agx_ingest_rgb + DMCVecEnv / DMC*Env drop-ins) against the oracle and against the
goldens the reference's own dmc_env.py produced (tests/golden/dmc_*.npz; control flow pinned, OpenCV's BGR2GRAY
arithmetic "parity unpinned" - see oracle/oracle.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from fake_dmc import ScriptedDMC
from oracle import oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-5


@pytest.mark.parametrize("mode", ["cv15", "cv14"])
@pytest.mark.parametrize("n,obs,fs", [(7, (84, 84), 3), (3, (12, 16), 2), (33, (36, 48), 4)])
def test_ingest_rgb_bit_exact(mode, n, obs, fs):
    from active_gym import ObsPipeline, _native as nat
    dev = torch.device("cuda:0")
    pipe = ObsPipeline(num_envs=n, kind="base", obs_size=obs, frame_stack=fs, device=dev)
    rng = np.random.default_rng(5)
    ring = np.zeros((n, fs) + obs, np.uint8)                     # oldest -> newest
    gm = nat.GRAY_CV15 if mode == "cv15" else nat.GRAY_CV14
    for step in range(7):
        frames = rng.integers(0, 256, (n,) + obs + (3,), dtype=np.uint8)
        if step == 2:
            frames[0] = 255
            frames[1, ..., 1:] = 0                                # pure channel 0
        cmd = np.ones(n, np.uint8)
        if step == 0:
            cmd |= nat.CMD_CLEAR
        if step == 3:
            cmd[0] = nat.CMD_SKIP | 1
            cmd[n - 1] = nat.CMD_CLEAR | 1
            cmd[1] = 0                                            # nvalid 0: zeros are appended
        pipe.ingest_rgb(torch.from_numpy(frames).to(dev), torch.from_numpy(cmd).to(dev), gm)
        gray = O.cv_bgr2gray_u8(frames, mode)
        for i in range(n):
            if cmd[i] & nat.CMD_SKIP:
                continue
            if cmd[i] & nat.CMD_CLEAR:
                ring[i] = 0
            new = gray[i] if (cmd[i] & 3) else np.zeros(obs, np.uint8)
            ring[i] = np.concatenate([ring[i, 1:], new[None]], 0)
        assert np.array_equal(pipe.stack_u8().cpu().numpy(), ring), step
    full = pipe.observe_full().cpu().numpy()
    assert np.array_equal(full, ring.astype(np.float32) / np.float32(255))
    assert pipe.algorithmic_bytes("ingest_rgb") == n * obs[0] * obs[1] * 4
    pipe.close()


def _args(seed, **kw):
    from active_gym import DMCEnvArgs
    base = dict(domain_name="scripted", task_name="t", seed=seed, obs_size=(12, 16),
                frame_source=lambda a, i: ScriptedDMC(a.seed + i, episode_len=getattr(a, "episode_len", 23)))
    base.update(kw)
    return DMCEnvArgs(**base)


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLD, "dmc_*.npz"))))
def test_single_env_replays_reference_golden(name):
    """DMCBaseEnv / DMCFixedFovealEnv of this package == the reference's, step for step, on the same script."""
    import active_gym
    g = np.load(os.path.join(GOLD, f"dmc_{name}.npz"))
    kw = dict(frame_stack=int(g["frame_stack"]), action_repeat=int(g["action_repeat"]), clip_reward=bool(g["clip_reward"]),
              episode_len=int(g["episode_len"]))
    fixed = bool(g["fixed_fov"])
    if fixed:
        kw.update(fov_size=(4, 6), fov_init_loc=(1, 2), sensory_action_mode="absolute", resize_to_full=True)
    args = _args(int(g["seed"]), obs_size=tuple(g["obs_size"]), **kw)
    env = active_gym.DMCFixedFovealEnv(args) if fixed else active_gym.DMCBaseEnv(args)
    base = env.unwrapped
    assert np.array_equal(base._true_action_space.low, g["true_low"]) and np.array_equal(base._true_action_space.high, g["true_high"])
    assert tuple(base.observation_space.shape) == tuple(g["obs_space_shape"]) and tuple(base.state_space.shape) == tuple(g["state_space_shape"])
    assert tuple(base.reward_range) == tuple(g["reward_range"])
    k = 0

    def check(s, r, d, info):
        nonlocal k
        if fixed:
            np.testing.assert_allclose(s, g["states_f64"][k], rtol=0, atol=TOL)
            assert np.array_equal(info["fov_loc"], g["fov_loc"][k])
        else:
            assert np.array_equal(s, g["states_u8"][k].astype(np.float32) / np.float32(255)), k
        assert float(r) == g["rewards"][k] and bool(d) == bool(g["dones"][k]) and float(info["raw_reward"]) == g["raw_rewards"][k]
        disc = np.nan if info["discount"] is None else info["discount"]
        assert (np.isnan(disc) and np.isnan(g["discount"][k])) or disc == g["discount"][k]
        assert np.array_equal(info["internal_state"], g["internal_state"][k])
        assert info["ep_len"] == g["ep_len"][k] and float(info["reward"]) == g["cum_reward"][k]
        assert np.array_equal(base.current_state, g["current_state"][k])
        k += 1

    s, info = env.reset()
    check(s, 0.0, False, info)
    for t in range(len(g["motor"])):
        if fixed:
            s, r, d, tr, info = env.step({"motor_action": g["motor"][t], "sensory_action": g["sens"][t]})
        else:
            s, r, d, tr, info = env.step(g["motor"][t])
        assert tr is False
        check(s, r, d, info)
        if d:
            s, info = env.reset()
            check(s, 0.0, False, info)
    assert k == len(g["rewards"])
    with pytest.raises(AssertionError):
        env.step({"motor_action": np.array([1.5, 0.0], np.float32), "sensory_action": (0, 0)} if fixed
                 else np.array([1.5, 0.0], np.float32))
    env.close()


@pytest.mark.parametrize("kind,gray", [("fixed", "cv15"), ("peripheral", "cv14"), ("flexible", "cv15"), ("base", "cv15")])
def test_dmc_vec_env_matches_oracle_with_autoreset(kind, gray):
    from active_gym import DMCVecEnv
    N, STEPS = 5, 40
    obs = (36, 48)
    kw = dict(obs_size=obs, fov_size=(10, 16), fov_init_loc=(2, 3), sensory_action_mode="absolute", resize_to_full=True,
              peripheral_res=(9, 7), episode_len=9, gray_mode=gray, clip_reward=(kind == "peripheral"))
    args = _args(40, **kw)
    env = DMCVecEnv(args, N, kind=kind)
    chains = []
    for i in range(N):
        e = O.DMCEnvOracle(ScriptedDMC(40 + i, episode_len=9), obs_size=obs, frame_stack=3, action_repeat=4,
                           clip_reward=args.clip_reward, gray_mode=gray)
        fkw = dict(obs_size=obs, fov_size=(10, 16), fov_init_loc=(2, 3), sensory_action_mode="absolute")
        fov = {"fixed": lambda: O.FixedFovealOracle(resize_to_full=True, mask_out=False, **fkw),
               "flexible": lambda: O.FlexibleFovealOracle(resize_to_full=True, mask_out=False, **fkw),
               "peripheral": lambda: O.PeripheralOracle(peripheral_res=(9, 7), **fkw), "base": lambda: None}[kind]()
        chains.append((O.RecordOracle(e), fov))

    def view(i, s, a=None, t=0, reset=False):
        fov = chains[i][1]
        s = s.astype(np.float64)
        if fov is None:
            return s
        if reset:
            return fov.reset(s)
        return fov.step(s, a, np.array((t,))) if kind == "flexible" else fov.step(s, a)

    rng = np.random.default_rng(8)
    obs_, infos = env.reset()
    want = [view(i, chains[i][0].reset()[0], reset=True) for i in range(N)]
    np.testing.assert_allclose(obs_, np.stack(want), rtol=0, atol=TOL)
    assert infos["discount"][0] is None and infos["internal_state"].shape == (N, 3)
    n_done = 0
    for step in range(STEPS):
        motor = rng.uniform(-1, 1, (N, 2)).astype(np.float32)
        types = rng.integers(0, 2, N)
        sens = rng.uniform(-4, 40, (N, 2))
        if kind == "flexible":
            sens = np.where(types[:, None] == 1, rng.integers(6, 36, (N, 2)), np.rint(sens)).astype(np.int64)
        act = motor if kind == "base" else {"motor_action": motor, "sensory_action": sens}
        if kind == "flexible":
            act["sensory_action_type"] = types
        o, rew, term, trunc, infos = env.step(act)
        for i in range(N):
            rec, fov = chains[i]
            s, r, d, tr, info = rec.step(motor[i])
            w = view(i, s, sens[i], int(types[i]))
            assert float(rew[i]) == float(r) and bool(term[i]) == bool(d), (step, i)
            if d:
                n_done += 1
                np.testing.assert_allclose(infos["final_observation"][i], w, rtol=0, atol=TOL)
                fi = infos["final_info"][i]
                assert fi["ep_len"] == info["ep_len"] and fi["raw_reward"] == info["raw_reward"] and fi["discount"] == info["discount"]
                assert np.array_equal(fi["internal_state"], info["internal_state"])
                s, info = rec.reset()
                w = view(i, s, reset=True)
            np.testing.assert_allclose(o[i], w, rtol=0, atol=TOL, err_msg=f"step {step} env {i}")
            assert infos["ep_len"][i] == info["ep_len"] and infos["reward"][i] == info["reward"]
            assert infos["discount"][i] == info["discount"] and np.array_equal(infos["internal_state"][i], info["internal_state"])
            if fov is not None:
                assert np.array_equal(infos["fov_loc"][i], fov.fov_loc)
    assert n_done >= 5
    env.close()


@pytest.mark.parametrize("kind", ["fixed", "flex", "per"])
def test_dmc_spaces_non_square_match_reference(kind):
    import json
    import active_gym
    gold = json.load(open(os.path.join(GOLD, "spaces.json")))
    factory = {"fixed": active_gym.DMCFixedFovealEnv, "flex": active_gym.DMCFlexibleFovealEnv,
               "per": active_gym.DMCFixedFovealPeripheralEnv}[kind]
    for tag, rtf, mo in (("abs_nonsq", False, False), ("abs_mask", False, True)):
        g = gold[f"{kind}_{tag}"]
        env = factory(_args(3, obs_size=(36, 48), frame_stack=4, fov_size=(10, 16), fov_init_loc=(2.5, 3.5),
                            sensory_action_mode="absolute", resize_to_full=rtf, mask_out=mo, peripheral_res=(9, 7)))
        sa, gs = env.action_space["sensory_action"], g["action_space"]["sensory_action"]
        assert float(np.min(sa.low)) == gs["low"] and float(np.max(sa.high)) == gs["high"] and np.dtype(sa.dtype).name == gs["dtype"]
        assert env.action_space["motor_action"].shape == (2,)                       # DMC: Box(-1, 1, action_dim)
        assert list(env.observation_space.shape) == g["observation_space"]["shape"]
        assert np.asarray(env.sensory_action_space).tolist() == g["sensory_action_space"]
        o, info = env.reset()
        assert list(o.shape) == g["observation_space"]["shape"]
        assert np.asarray(env.fov_loc).tolist() == g["fov_loc"] and bool(env.mask_out) == g["mask_out"]
        env.close()

"""CPU: the host runner (active_gym/runner.py) reproduces the reference's
emulator-facing control flow.  Replays the atari goldens (reference AtariEnv +
RecordWrapper over a scripted emulator): the runner's screens + command bytes,
pushed through a plain Python model of the device ring, must give the
reference's states, rewards, dones and counters."""
import collections

import numpy as np
import pytest

from fake_ale import ScriptedALE
from golden_util import atari_case_names, load_atari

from active_gym import _native as nat
from active_gym.runner import AtariHostRunner


class _Args:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _RingModel:
    """What agx_ingest does, with 'mean of RGB' as the gray stage (the goldens use ScriptedALE's
    getScreenGrayscale == channel mean and an identity resize)."""

    def __init__(self, fs, obs):
        self.fs, self.obs = fs, obs
        self.dq = collections.deque([np.zeros(obs, np.uint8)] * fs, maxlen=fs)

    def ingest(self, frames2, cmd):
        if cmd & nat.CMD_SKIP:
            return
        if cmd & nat.CMD_CLEAR:
            for _ in range(self.fs):
                self.dq.append(np.zeros(self.obs, np.uint8))
        o = np.zeros(self.obs, np.uint8)
        for f in range(cmd & 3):
            rgb = frames2[f].astype(np.uint16)
            o = np.maximum(o, ((rgb[..., 0] + rgb[..., 1] + rgb[..., 2]) // 3).astype(np.uint8))
        self.dq.append(o)

    def stack(self):
        return np.stack(list(self.dq))


@pytest.mark.parametrize("name", [n for n in atari_case_names() if not n.startswith("fixedfov")])
def test_runner_matches_reference_control_flow(name, monkeypatch):
    c = load_atari(name)
    obs = tuple(int(v) for v in c["obs_size"])
    noops = list(c["noops"])
    args = _Args(game="scripted", seed=c["seed"], action_repeat=c["action_repeat"], clip_reward=c["clip_reward"],
                 frame_source=lambda a, i: ScriptedALE(seed=c["seed"], screen_hw=obs, n_actions=c["n_actions"]))
    frames = np.zeros((1, 2) + obs + (3,), np.uint8)
    # the runner asserts the Atari screen shape; the goldens use small scripted screens
    r = AtariHostRunner.__new__(AtariHostRunner)
    monkeypatch.setattr("active_gym.runner.RAW_H", obs[0])
    monkeypatch.setattr("active_gym.runner.RAW_W", obs[1])
    AtariHostRunner.__init__(r, args, 1, frames=frames, workers=1, noop_fn=lambda: int(noops.pop(0)))
    if not c["training"]:
        r.eval()
    ring = _RingModel(c["frame_stack"], obs)
    want = c["states_u8"]
    cum, ep = 0.0, 0
    i = 0
    cmd = r.reset()
    ring.ingest(frames[0], int(cmd[0]))
    assert np.array_equal(ring.stack(), want[i]) and c["is_reset"][i]
    for t in range(len(c["motor"])):
        i += 1
        ret, done, cmd, raw = r.step([int(c["motor"][t])])
        ring.ingest(frames[0], int(cmd[0]))
        ep += 1
        cum += raw[0]
        assert np.array_equal(ring.stack(), want[i]), (name, t)
        assert ret[0] == c["rewards"][i] and bool(done[0]) == bool(c["dones"][i])
        assert ep == c["ep_len"][i] and cum == c["cum_reward"][i]
        if done[0]:
            i += 1
            cmd = r.reset([0])
            ring.ingest(frames[0], int(cmd[0]))
            cum, ep = 0.0, 0
            assert np.array_equal(ring.stack(), want[i]) and c["is_reset"][i]
    assert i + 1 == len(c["dones"]) and not noops


def test_runner_threads_equal_serial():
    def mk(workers):
        args = _Args(game="g", seed=5, action_repeat=4, clip_reward=False,
                     frame_source=lambda a, i: ScriptedALE(seed=100 + i, n_actions=6))
        return AtariHostRunner(args, 8, workers=workers, noop_fn=lambda: 3)
    a, b = mk(1), mk(4)
    rng = np.random.default_rng(0)
    assert np.array_equal(a.reset(), b.reset()) and np.array_equal(a.frames[:, 0], b.frames[:, 0])
    for _ in range(12):
        m = rng.integers(0, 6, 8)
        ra, rb = a.step(m), b.step(m)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)
        assert np.array_equal(a.frames, b.frames)
        d = np.nonzero(ra[1])[0]
        if len(d):
            assert np.array_equal(a.reset(d), b.reset(d))
    a.close(); b.close()


def test_args_and_spaces_surface():
    from active_gym import AtariEnvArgs
    from active_gym.spaces import Box, Dict, Discrete
    a = AtariEnvArgs(game="breakout", seed=1, obs_size=(84, 84), fov_size=(30, 30), anything=7)
    assert (a.frame_stack, a.action_repeat, a.mask_out, a.record, a.clip_reward) == (4, 4, False, False, False)
    assert a.max_episode_length == 108e3 and a.anything == 7 and a.device is None and a.env_backend == "atari_py"
    d = Dict({"motor_action": Discrete(4), "sensory_action": Box(low=54, high=54, dtype=int)})
    d["sensory_action_type"] = Discrete(2)
    assert set(d.keys()) == {"motor_action", "sensory_action", "sensory_action_type"}
    assert d["sensory_action"].shape == () and d["motor_action"].n == 4


def test_frame_source_ale_missing_is_loud():
    from active_gym import AtariEnvArgs
    from active_gym.frame_source import make_emulator, SyntheticALE
    try:
        import atari_py  # noqa: F401
        pytest.skip("atari_py present")
    except ImportError:
        pass
    try:
        import ale_py  # noqa: F401
        pytest.skip("ale_py present")
    except ImportError:
        pass
    with pytest.raises(ImportError, match="atari_py"):
        make_emulator(AtariEnvArgs(game="breakout", seed=0, obs_size=(84, 84)))
    e = make_emulator(AtariEnvArgs(game="breakout", seed=0, obs_size=(84, 84), frame_source="synthetic"))
    assert isinstance(e, SyntheticALE) and e.getScreenRGB().shape == (210, 160, 3)


def test_batch_space_fallback_matches_gymnasium_conventions():
    from active_gym import spaces as sp
    if sp.HAVE_GYMNASIUM:
        pytest.skip("gymnasium present: its own batch_space is used")
    single = sp.Dict({"motor_action": sp.Discrete(6), "sensory_action": sp.Box(low=54, high=54, dtype=int),
                      "obs": sp.Box(low=-1.0, high=1.0, shape=(4, 8, 8), dtype=np.float32)})
    b = sp.batch_space(single, 5)
    assert b["motor_action"].nvec.tolist() == [6] * 5 and b["motor_action"].contains(np.array([0, 5, 3, 2, 1]))
    assert b["sensory_action"].shape == (5,) and b["sensory_action"].low.tolist() == [54] * 5
    assert b["obs"].shape == (5, 4, 8, 8) and b["obs"].dtype == np.float32 and b.sample()["obs"].shape == (5, 4, 8, 8)

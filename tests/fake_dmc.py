"""A scripted stand-in for a dm_control Environment (event script + arithmetic renders, no physics): exactly the
surface the reference's DMCEnv touches (dmc_env.py:102-131,175-192,204,223-225).  Used to generate the DMC
control-flow goldens from the reference and to drive the product / the oracle identically."""
import collections

import numpy as np


class Array:                                   # dm_env.specs.Array
    def __init__(self, shape, dtype=np.float64):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)


class BoundedArray(Array):                     # dm_env.specs.BoundedArray
    def __init__(self, shape, minimum, maximum, dtype=np.float64):
        super().__init__(shape, dtype)
        self.minimum = np.asarray(minimum, dtype=self.dtype)
        self.maximum = np.asarray(maximum, dtype=self.dtype)


class TimeStep:
    def __init__(self, step_type, reward, discount, observation):
        self.step_type, self.reward, self.discount, self.observation = step_type, reward, discount, observation

    def last(self):
        return self.step_type == 2


class _Physics:
    def __init__(self, env):
        self.env = env

    def render(self, height, width, camera_id=0):
        e = self.env
        y, x = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
        k = e.seed * 131 + e.episode * 17 + e.t * 7 + camera_id * 3
        base = (y * 5 + x * 11 + k + ((y * x) >> 3)) & 0xFFFF
        # the last commanded action leaves a mark so that the action conversion is visible in the pixels
        a = int(np.rint(np.sum(e.last_action) * 1000)) & 0xFF
        return np.stack([(base + a) & 0xFF, (base * 3 + 40) & 0xFF, (base * 7 + 90 + (k >> 2)) & 0xFF], -1).astype(np.uint8)

    def get_state(self):
        e = self.env
        return np.array([e.t, e.episode, float(np.sum(e.last_action))], dtype=np.float64)


class ScriptedDMC:
    def __init__(self, seed, action_dim=2, low=-2.0, high=3.0, episode_len=23):
        self.seed, self.action_dim, self.low, self.high, self.episode_len = seed, action_dim, low, high, episode_len
        self.rng = np.random.default_rng(seed)
        self.t, self.episode = 0, 0
        self.last_action = np.zeros(action_dim, np.float32)
        self.physics = _Physics(self)

    def action_spec(self):
        return BoundedArray((self.action_dim,), np.full(self.action_dim, self.low), np.full(self.action_dim, self.high))

    def observation_spec(self):
        return collections.OrderedDict(position=Array((2,)), velocity=Array((1,)))

    def _obs(self):
        return collections.OrderedDict(position=np.array([self.t * 0.5, self.episode * 1.0]), velocity=np.array([float(self.t)]))

    def reset(self):
        self.t = 0
        self.episode += 1
        self.last_action = np.zeros(self.action_dim, np.float32)
        return TimeStep(0, None, None, self._obs())

    def step(self, action):
        action = np.asarray(action)
        assert action.dtype == np.float32 and action.shape == (self.action_dim,)
        self.last_action = action
        self.t += 1
        u = self.rng.random()
        reward = None if u < 0.1 else float(np.round(self.rng.normal(), 3))       # `time_step.reward or 0`
        last = self.t >= self.episode_len
        return TimeStep(2 if last else 1, reward, 0.0 if last else 1.0, self._obs())

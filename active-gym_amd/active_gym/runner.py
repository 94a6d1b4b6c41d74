"""AtariHostRunner — the host half of N Atari envs: one emulator per env,
stepped by a pool of worker threads (one per host core), writing raw RGB
screens into a (pinned) staging buffer for the device ingest kernel.

It reproduces, per env, the emulator-facing control flow of the reference's
``AtariEnv._step`` / ``_reset`` (reference atari_env.py:84-148) — everything
except the image arithmetic, which is what the HIP kernels do:

  step : ``action_repeat`` x ``ale.act``; the screens after t==2 and t==3 are
         sampled (whatever action_repeat is, reference atari_env.py:125-128);
         early ``break`` on game over; life-loss terminal in training mode.
  reset: life-loss reset (one no-op, stack kept) or full reset (stack zeroed,
         ``random.randrange(30)`` no-ops from the GLOBAL ``random`` module like
         the reference, fire-reset when the game has >= 3 actions); one screen.

Each call returns the per-env command byte for ``agx_ingest``:
``nvalid | CMD_CLEAR | CMD_SKIP`` (include/agx.h).
"""
from __future__ import annotations

import os
import random
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Optional, Sequence

import numpy as np

from . import _native as nat
from .frame_source import RAW_H, RAW_W, make_emulator, resolve_frame_format


def per_env_noop_seed(seed, global_index: int) -> int:
    """Seed of env `global_index`'s own no-op stream (noop_per_env), shared by the Python and the native runner."""
    return (int(seed) * 1000003 + int(global_index)) & 0x7FFFFFFF


def check_motor_actions(motor, num_actions: int):
    """Index into the minimal action set, validated like the native runner does (the reference's `actions.get(a)`
    yields None for an unknown index and ALE raises, atari_env.py:122-124): no silent wrap-around of -1, no truncation
    of fractional values."""
    m = np.asarray(motor)
    if m.dtype.kind == "f":
        if not np.all(m == np.floor(m)):
            raise ValueError("motor_action must be integral")
    elif m.dtype.kind not in "iub":
        raise TypeError(f"motor_action dtype {m.dtype} not supported")
    mi = m.astype(np.int64)
    if mi.size and (mi.min() < 0 or mi.max() >= num_actions):
        raise ValueError(f"motor_action out of range [0, {num_actions})")
    return mi


class AtariHostRunner:
    def __init__(self, args, num_envs: int, frames: Optional[np.ndarray] = None,
                 workers: Optional[int] = None, noop_fn: Optional[Callable[[], int]] = None,
                 env_offset: int = 0, noop_per_env: bool = False, src_rows: Optional[Sequence[int]] = None):
        """``src_rows``: compact staging - only these screen rows are staged (``ObsPipeline.source_rows()``: the rows
        cv2.resize reads, 168 of 210 for 84 x 84); every screen in ``frames`` then has ``len(src_rows)`` rows."""
        self.args = args
        self.num_envs = int(num_envs)
        self.action_repeat = int(args.action_repeat)
        self.clip_reward = bool(args.clip_reward)
        self.training = True                    # reference atari_env.py:58: args.training is ignored
        self.noop_fn = noop_fn or (lambda: random.randrange(30))     # reference atari_env.py:96
        self.env_offset = int(env_offset)
        # noop_per_env: every env draws its no-op counts from a stream of its own, seeded by (args.seed, GLOBAL env
        # index), instead of the process-global `random` the reference uses - what makes a sharded run reproduce the
        # unsharded one (sharding.py); an explicit noop_fn wins
        self._noop_rngs = ([random.Random(per_env_noop_seed(args.seed, self.env_offset + i)) for i in range(self.num_envs)]
                           if noop_per_env and noop_fn is None else None)
        self.emulators = [make_emulator(args, env_offset + i) for i in range(self.num_envs)]
        acts = [list(e.getMinimalActionSet()) for e in self.emulators]
        self.actions = acts                     # index -> emulator action, reference atari_env.py:51-52
        self.num_actions = len(acts[0])
        self.lives = np.zeros(self.num_envs, np.int64)
        self.life_termination = np.zeros(self.num_envs, bool)
        # frame_format "gray": ALE's own grayscale screens (getScreenGrayscale - what the reference reads,
        # atari_env.py:74) instead of RGB; a third of the bytes, no luminance arithmetic on the device
        self.gray = resolve_frame_format(args) == "gray"
        self.src_rows = None if src_rows is None else np.asarray(src_rows, dtype=np.intp)
        self.rows = RAW_H if self.src_rows is None else len(self.src_rows)
        shape = (self.num_envs, 2, self.rows, RAW_W) + (() if self.gray else (3,))
        if frames is None:
            frames = np.zeros(shape, np.uint8)
        assert frames.shape == shape and frames.dtype == np.uint8
        self.frames = frames
        if workers is None:
            # the CPUs this process may use (affinity, cgroup quota) shared between the ranks of the node: hostplan.py
            from .hostplan import read_topology, usable_cpus
            lws = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
            workers = max(1, min(64, usable_cpus(read_topology()) // lws))
        n_workers = min(self.num_envs, int(workers))
        self._pool = ThreadPoolExecutor(max_workers=n_workers) if n_workers > 1 else None
        self._n_workers = max(1, n_workers)

    # ------------------------------------------------------------------ helpers
    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    def set_frames(self, frames: np.ndarray):
        """Point the runner at another staging buffer of the same shape (double-buffered pinned staging)."""
        assert frames.shape == self.frames.shape and frames.dtype == np.uint8 and frames.flags.c_contiguous
        self.frames = frames

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    def _grab(self, i, slot, buf=None, row=None):
        e = self.emulators[i]
        dst = (self.frames if buf is None else buf)[i if row is None else row, slot]
        scr = np.asarray(e.getScreenGrayscale()).reshape(RAW_H, RAW_W) if self.gray else np.asarray(e.getScreenRGB())
        if self.src_rows is None:
            np.copyto(dst, scr)
        else:
            np.take(scr, self.src_rows, axis=0, out=dst)

    def _map(self, fn, idx: Sequence[int]):
        idx = list(idx)
        if self._pool is None or len(idx) <= 1:
            for i in idx:
                fn(i)
            return
        chunks = np.array_split(np.asarray(idx), min(self._n_workers, len(idx)))
        futs = [self._pool.submit(lambda c=c: [fn(int(i)) for i in c]) for c in chunks if len(c)]
        for f in futs:
            f.result()

    def _draw_noops(self, i: int) -> int:
        return self._noop_rngs[i].randrange(30) if self._noop_rngs is not None else int(self.noop_fn())

    # ------------------------------------------------------------------ step   (atari_env.py:119-148)
    def step(self, motor_actions) -> tuple:
        n = self.num_envs
        motor = check_motor_actions(motor_actions, self.num_actions).reshape(n)
        raw = np.zeros(n, np.float64)
        done = np.zeros(n, bool)
        cmd = np.zeros(n, np.uint8)

        def one(i):
            e = self.emulators[i]
            a = self.actions[i][int(motor[i])]
            reward, d, nvalid = 0, False, 0
            for t in range(self.action_repeat):
                reward += e.act(a)
                if t == 2:
                    self._grab(i, 0)
                    nvalid = 1
                elif t == 3:
                    self._grab(i, 1)
                    nvalid = 2
                d = e.game_over()
                if d:
                    break
            if self.training:
                lives = e.lives()
                if lives < self.lives[i] and lives > 0:
                    self.life_termination[i] = not d
                    d = True
                self.lives[i] = lives
            raw[i] = reward
            done[i] = d
            cmd[i] = nvalid

        self._map(one, range(n))
        ret = np.sign(raw) if self.clip_reward else raw.copy()      # atari_env.py:144
        return ret, done, cmd, raw

    # ------------------------------------------------------------------ reset  (atari_env.py:84-117)
    def reset(self, idx: Optional[Sequence[int]] = None, out: Optional[np.ndarray] = None, packed: bool = False) -> np.ndarray:
        """Reset the envs in `idx` (all by default).  The single reset screen of env i goes to
        ``out[i, 0]`` (default: the step staging buffer), or - ``packed`` - the j-th reset env's to ``out[j, 0]``."""
        n = self.num_envs
        idx = list(range(n)) if idx is None else [int(i) for i in idx]
        row_of = {i: j for j, i in enumerate(idx)} if packed else None
        cmd = np.full(n, nat.CMD_SKIP, np.uint8)
        # no-op counts are drawn on the calling thread, in env order, from the global RNG
        noops = {i: (0 if self.life_termination[i] else self._draw_noops(i)) for i in idx}

        def one(i):
            e = self.emulators[i]
            if self.life_termination[i]:
                self.life_termination[i] = False
                e.act(0)
                clear = 0
            else:
                clear = nat.CMD_CLEAR
                e.reset_game()
                for _ in range(noops[i]):
                    e.act(0)
                    if e.game_over():
                        e.reset_game()
            if len(self.actions[i]) >= 3:
                e.act(1)
                if e.game_over():
                    e.reset_game()
                    e.act(2)
                if e.game_over():
                    e.reset_game()
            self._grab(i, 0, out, None if row_of is None else row_of[i])
            self.lives[i] = e.lives()
            cmd[i] = 1 | clear

        self._map(one, idx)
        return cmd

    def render(self, i=0, size=(256, 256)):
        """Raw RGB screen of env i (the reference resizes it with cv2, atari_env.py:165-169)."""
        return np.array(self.emulators[i].getScreenRGB(), copy=True)

"""CPU: env-index sharding — pure partition logic, plus a
world_size-2 gloo run: each rank drives the host runner + a ring model for its
shard with no data-path collective; the gathered per-env digests must equal the
unsharded run, and the bench's barrier / max-over-ranks timing helper works."""
import os
import subprocess
import sys

import numpy as np
import pytest

from active_gym.sharding import shard_bounds

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,w", [(8192, 8), (1024, 1), (10, 4), (7, 8), (1000, 3)])
def test_shard_bounds_partition(n, w):
    spans = [shard_bounds(n, r, w) for r in range(w)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    for (a, b), (c, d) in zip(spans, spans[1:]):
        assert b == c and a <= b
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(n, w, w)


WORKER = r'''
import os, sys, zlib
import numpy as np, torch, torch.distributed as dist
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), os.path.join(REPO, "tests"), REPO]
from fake_ale import ScriptedALE
from active_gym.runner import AtariHostRunner
from active_gym.sharding import shard_bounds
import bench

class A:
    pass
def run(lo, hi, steps=6):
    a = A(); a.game="g"; a.seed=11; a.action_repeat=4; a.clip_reward=False
    a.frame_source = lambda args, i: ScriptedALE(seed=1000 + i, n_actions=4)     # i is the GLOBAL env index
    r = AtariHostRunner(a, hi - lo, workers=2, noop_fn=lambda: 2, env_offset=lo)
    rng = np.random.default_rng(5)
    motor_all = rng.integers(0, 4, size=(steps, N))
    dig = np.zeros(hi - lo, np.int64)
    def upd(cmd, which):
        for k in which:                      # only the envs the call touched: a digest is per env
            dig[k] = zlib.crc32(r.frames[k].tobytes() + bytes([int(cmd[k])]), int(dig[k]) & 0xFFFFFFFF)
    upd(r.reset(), range(hi - lo))
    for t in range(steps):
        ret, done, cmd, raw = r.step(motor_all[t, lo:hi]); upd(cmd, range(hi - lo))
        d = np.nonzero(done)[0]
        if len(d): upd(r.reset(d), d)
    r.close()
    return dig

N = 6
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{PORT}", rank=RANK, world_size=2)
lo, hi = shard_bounds(N, RANK, 2)
local = run(lo, hi)
parts = [None, None]
dist.all_gather_object(parts, (lo, hi, local.tolist()))
t = bench.max_over_ranks(0.5 + RANK, dist, torch.device("cpu"))
bench.barrier(dist, None)
if RANK == 0:
    full = run(0, N)
    got = np.zeros(N, np.int64)
    for a_, b_, d_ in parts: got[a_:b_] = d_
    assert np.array_equal(got, full), (got, full)
    assert abs(t - 1.5) < 1e-9, t
    print("SHARD_OK")
dist.destroy_process_group()
'''


def test_two_rank_gloo_shards_reproduce_unsharded(tmp_path):
    port = 29650 + (os.getpid() % 200)
    procs = []
    for rank in range(2):
        code = f"REPO={REPO!r}; RANK={rank}; PORT={port}\n" + WORKER
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True, cwd=REPO))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARD_OK" in outs[0]


def test_shard_game_mix():
    from active_gym import shard_game
    games = ["breakout", "boxing", "pong"]
    assert [shard_game(games, r) for r in range(8)] == ["breakout", "boxing", "pong", "breakout", "boxing", "pong", "breakout", "boxing"]
    assert shard_game("seaquest", 5) == "seaquest"
    with pytest.raises(ValueError):
        shard_game([], 0)

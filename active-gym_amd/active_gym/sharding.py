"""Env-index data parallelism across the GPUs of one node.

Each env's state (emulator, frame-stack ring, fov_loc / fov_res, counters) is
private (reference atari_env.py:57, fov_env.py:146-150,245), so the batch
shards into contiguous env-index blocks with NO data-path collective: rank g
of G owns envs [g*N/G, (g+1)*N/G).  One process per GPU (torch.distributed
only carries barriers / timing reductions in bench.py)."""
from __future__ import annotations

import os
from typing import Tuple


def shard_bounds(num_envs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of env indices owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, rem = divmod(int(num_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_game(game, rank: int):
    """BASELINE.json configs[4] ("mixed Atari-57 games", one shard per GPU): ``args.game`` may be a list, rank g then
    plays ``game[g % len(game)]`` - every env of one shard shares a game, hence one minimal action set and one
    action space per vec env, as with the reference's per-game envs."""
    if isinstance(game, (list, tuple)):
        if not game:
            raise ValueError("empty game list")
        return game[rank % len(game)]
    return game


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


class ShardedAtariVecEnv:
    """The local shard of a global batch of `num_envs` envs.  Env i of the global batch keeps its identity whatever
    the world size - emulator seed = args.seed + i, and (``noop_per_env``, on by default here) a no-op reset stream of
    its own seeded by (args.seed, i) instead of the reference's process-global ``random`` - so a sharded run reproduces
    ``AtariVecEnv(args, num_envs, noop_per_env=True)`` env for env (tests/test_gpu_sharding.py).  Pass
    ``noop_per_env=False`` (or a ``noop_fn``) to keep the reference's global stream per process."""

    def __init__(self, args, num_envs: int, kind: str = "fixed", rank=None, world_size=None, local_rank=None, **kw):
        import torch
        from .vector import AtariVecEnv
        r, w, lr = env_rank_world()
        self.rank = r if rank is None else rank
        self.world_size = w if world_size is None else world_size
        self.local_rank = lr if local_rank is None else local_rank
        self.global_num_envs = int(num_envs)
        self.lo, self.hi = shard_bounds(num_envs, self.rank, self.world_size)
        if getattr(args, "device", None) is None or str(args.device) == "cuda":
            torch.cuda.set_device(self.local_rank)
        if isinstance(getattr(args, "game", None), (list, tuple)):
            import copy
            args = copy.copy(args)
            args.game = shard_game(args.game, self.rank)
        self.game = getattr(args, "game", None)
        kw.setdefault("noop_per_env", kw.get("noop_fn") is None)
        self.env = AtariVecEnv(args, self.hi - self.lo, kind=kind, env_offset=self.lo, **kw)

    def __getattr__(self, name):
        return getattr(self.env, name)


def make_vec_env(args, num_envs: int, kind: str = "fixed", **kw):
    """AtariVecEnv on one GPU, or the local shard when launched under torch.distributed.run."""
    _, world, _ = env_rank_world()
    if world > 1:
        return ShardedAtariVecEnv(args, num_envs, kind=kind, **kw)
    from .vector import AtariVecEnv
    return AtariVecEnv(args, num_envs, kind=kind, **kw)

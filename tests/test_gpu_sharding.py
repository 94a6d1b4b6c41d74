"""The DEVICE path under two ranks: two processes, each owning the contiguous env block of its rank
(`ShardedAtariVecEnv`: host runner + pinned staging + H2D + ingest + fovea kernels for its shard, no data-path
collective), must reproduce the unsharded N-env run env for env - observations, rewards, terminals, fov_loc.  Both
ranks share cuda:0 here (the box has one GPU; RCCL refuses two ranks on one device, so the control plane is gloo -
the data path has no collective to carry).  Plus a rehearsal of `bench.py --gpus 2`: one JSON line, n_gpus 2, one
`per_gpu` value per rank.  Ranks are fresh child processes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, zlib
import numpy as np, torch, torch.distributed as dist
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), os.path.join(REPO, "tests"), REPO]
from active_gym import AtariEnvArgs, AtariVecEnv
from active_gym.sharding import ShardedAtariVecEnv

N, STEPS = 10, 20
def mk():
    return AtariEnvArgs(game="breakout", seed=3, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0),
                        sensory_action_mode="relative", sensory_action_space=(-8.0, 8.0), resize_to_full=True,
                        frame_source="native", device="cuda:0", num_workers=2, scripted_p_life=40, scripted_p_over=10,
                        h2d_chunk_envs=KIND_CHUNK)

def run(env, lo, hi):
    rng = np.random.default_rng(17)                      # the GLOBAL action script; a shard takes its slice
    dig = np.zeros(hi - lo, np.int64)
    def upd(obs, extra):
        o = obs.cpu().numpy()
        for k in range(hi - lo):
            dig[k] = zlib.crc32(o[k].tobytes() + extra[k].tobytes(), int(dig[k]) & 0xFFFFFFFF)
    obs, info = env.reset()
    upd(obs, info["fov_loc"].cpu().numpy())
    for t in range(STEPS):
        motor = rng.integers(0, 4, N)
        sens = rng.uniform(-10, 10, (N, 2)).astype(np.float32)
        obs, rew, done, trunc, info = env.step({"motor_action": motor[lo:hi], "sensory_action": sens[lo:hi]})
        upd(obs, np.concatenate([info["fov_loc"].cpu().numpy(), rew[:, None].astype(np.int64), done[:, None].astype(np.int64),
                                 info["ep_len"][:, None].astype(np.int64)], 1))
    return dig

dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{PORT}", rank=RANK, world_size=2)
env = ShardedAtariVecEnv(mk(), N, kind="fixed", rank=RANK, world_size=2, local_rank=0)
assert (env.lo, env.hi) == ((0, 5) if RANK == 0 else (5, 10)) and env.num_envs == 5
local = run(env, env.lo, env.hi)
env.close()
parts = [None, None]
dist.all_gather_object(parts, (env.lo, env.hi, local.tolist()))
if RANK == 0:
    full_env = AtariVecEnv(mk(), N, kind="fixed", noop_per_env=True)
    full = run(full_env, 0, N)
    full_env.close()
    got = np.zeros(N, np.int64)
    for a_, b_, d_ in parts: got[a_:b_] = d_
    assert np.array_equal(got, full), (got, full)
    assert "libagx.so" in open("/proc/self/maps").read()
    print("DEVICE_SHARD_OK")
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("chunk", [0, 2])
def test_two_ranks_on_the_device_path_reproduce_the_unsharded_run(chunk):
    port = 29850 + (os.getpid() % 100) + chunk
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for rank in range(2):
        code = f"REPO={REPO!r}; RANK={rank}; PORT={port}; KIND_CHUNK={chunk}\n" + WORKER
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True, cwd=REPO, env=env))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "DEVICE_SHARD_OK" in outs[0]


def test_bench_two_gpu_rehearsal_prints_one_line_with_per_gpu_values():
    env = dict(os.environ, AGX_BENCH_SHARE_GPU="1", AGX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "24", "--warmup", "4", "--envs", "256",
                        "--no-cpu-baseline"], capture_output=True, text=True, cwd=REPO, env=env, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["total_envs"] == 512
    assert len(d["per_gpu"]) == 2 and all(v > 0 for v in d["per_gpu"])
    assert d["value"] <= sum(d["per_gpu"]) * 1.0001          # whole-job rate uses the slowest rank's time
    assert "cpu_baseline" not in d and "e2e" not in d
    # the PCIe- and emulator-inclusive leg runs on every rank at once; each rank reports its own share of the host
    # (host cores partitioned per rank): workers sized by LOCAL_WORLD_SIZE, disjoint pinned CPUs
    e = d["e2e_per_rank"]
    assert len(e) == 2 and all("error" not in x for x in e), e
    assert all(x["local_world_size"] == 2 and x["rgb"]["env_steps_per_s"] > 0 and x["gray"]["env_steps_per_s"] > 0 for x in e)
    assert all(x["rgb"]["rows_staged_per_screen"] == 168 for x in e)
    cpus = [set(x["placement"]["pinned_cpus"]) for x in e]
    assert cpus[0] and cpus[1] and not (cpus[0] & cpus[1]), cpus
    assert all(x["placement"]["workers"] == x["workers"] for x in e)
    assert [x["game"] for x in e] == ["breakout", "boxing"]                      # configs[4]: one game per shard
    agg = d["e2e_aggregate"]
    assert agg["ranks"] == 2 and abs(agg["rgb"]["env_steps_per_s_median"] - sum(x["rgb"]["env_steps_per_s_median"] for x in e)) < 1e-6


def test_bench_single_rank_over_rccl():
    """The control plane of the multi-GPU bench on the real backend: one rank, process group forced, backend nccl (= RCCL).
    Executes, on the GPU, what the 8-GPU run depends on and a one-GPU lease otherwise never runs: RCCL initialisation with a
    bound device, barrier(device_ids=...), all_reduce(MAX) of the timed region and the all_gather behind `per_gpu`."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, AGX_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("AGX_BENCH_BACKEND", None)
    env.pop("AGX_BENCH_SHARE_GPU", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "24", "--warmup", "4", "--envs", "256",
                        "--no-cpu-baseline", "--no-e2e"], capture_output=True, text=True, cwd=REPO, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and len(d["per_gpu"]) == 1 and d["per_gpu"][0] > 0
    assert abs(d["per_gpu"][0] - d["value"]) / d["value"] < 1e-6          # one rank: its own rate is the whole job's
    assert d["control_plane"] == "nccl"

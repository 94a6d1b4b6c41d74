#!/usr/bin/env python3
"""K1 and K2 alone and alternating, N = 1024, a pool of 8 screen batches (nothing served from the Infinity Cache): per-launch time of
300 back-to-back launches between two syncs.  Separates what a kernel costs by itself from what it costs behind the other one."""
import os, sys, time
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "active-gym_amd")]
import torch
from active_gym import ObsPipeline
dev = torch.device("cuda:0")
N, R = int(os.environ.get("N", 1024)), 300
p = ObsPipeline(num_envs=N, kind="fixed", fov_size=(30, 30), resize_to_full=True, device=dev)
g = torch.Generator(device=dev); g.manual_seed(0)
frames = [torch.randint(0, 256, (N, 2, 210, 160, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(8)]
cmd = torch.full((N,), 2, dtype=torch.uint8, device=dev)
act = [torch.rand((N, 2), device=dev) * 54 for _ in range(8)]
obs = [torch.empty(p.obs_shape, device=dev) for _ in range(2)]
def run(name, fn):
    for k in range(40): fn(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(R): fn(k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R * 1e6
    print(f"{name:34s} {dt:7.2f} us per iteration", flush=True)
for rep in range(2):
    run("K1 alone", lambda k: p.ingest(frames[k % 8], cmd))
    run("K2 alone", lambda k: p.fovea(act[k % 8], out=obs[k % 2]))
    run("K1 + K2", lambda k: (p.ingest(frames[k % 8], cmd), p.fovea(act[k % 8], out=obs[k % 2])))
    run("K1 + K1", lambda k: (p.ingest(frames[k % 8], cmd), p.ingest(frames[(k + 4) % 8], cmd)))
    run("K2 + K2", lambda k: (p.fovea(act[k % 8], out=obs[0]), p.fovea(act[(k + 4) % 8], out=obs[1])))

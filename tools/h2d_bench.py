#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: the raw screens start in PINNED HOST memory every step and are copied
to HBM (hipMemcpyAsync through torch) before ingest + fovea.  Reported in DESIGN.md, never as bench.py's value."""
import os, sys, time, json
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "active-gym_amd")]
import torch
from active_gym import ObsPipeline
dev = torch.device("cuda:0")
N, K = 1024, 60
p = ObsPipeline(num_envs=N, kind="fixed", fov_size=(30, 30), resize_to_full=True, device=dev)
h = [torch.randint(0, 256, (N, 2, 210, 160, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
d = torch.empty((N, 2, 210, 160, 3), dtype=torch.uint8, device=dev)
cmd = torch.full((N,), 2, dtype=torch.uint8, device=dev)
act = torch.rand((N, 2), device=dev) * 54
obs = torch.empty(p.obs_shape, device=dev)
def step(k):
    d.copy_(h[k & 1], non_blocking=True); p.ingest(d, cmd); p.fovea(act, out=obs)
for k in range(5): step(k)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(K): step(k)
torch.cuda.synchronize(); el = time.perf_counter() - t0
nbytes = h[0].numel()
print(json.dumps({"h2d_inclusive_env_steps_per_s": N * K / el, "ms_per_step": el / K * 1e3,
                  "h2d_GBps_effective": nbytes * K / el / 1e9, "bytes_per_step": nbytes}))

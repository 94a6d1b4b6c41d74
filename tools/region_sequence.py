#!/usr/bin/env python3
"""us per step of twelve consecutive 20-step timed regions (synchronize, 20 steps, synchronize) in ONE process, the first right after a
205-step warm-up as bench.py ran it until round 4, with half a second of idle GPU after the sixth: what a short timed region costs at
the start of a process and after an idle spell (power management), against the steady state in between.
    python tools/region_sequence.py [spin]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd")]
import torch
from active_gym import ObsPipeline
K = 20; N, POOL = 1024, 8
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
if len(sys.argv) > 1 and sys.argv[1] == "spin":
    import ctypes
    ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(1)
pipe = ObsPipeline(num_envs=N, kind="fixed", fov_size=(30, 30), resize_to_full=True, device=dev)
g = torch.Generator(device=dev); g.manual_seed(7)
frames = [torch.randint(0, 256, (N, 2, 210, 160, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(POOL)]
cmds = [torch.full((N,), 2, dtype=torch.uint8, device=dev) for _ in range(POOL)]
acts = [(torch.rand((N, 2), device=dev, generator=g) * 54).contiguous() for _ in range(POOL)]
obs = torch.empty(pipe.obs_shape, dtype=torch.float32, device=dev); loc = torch.empty((N, 2), dtype=torch.int32, device=dev)
def step(k):
    i = k % POOL
    pipe.ingest(frames[i], cmds[i]); pipe.fovea(acts[i], out=obs, loc_out=loc)
for k in range(205): step(k)
out = []
for r in range(12):
    torch.cuda.synchronize(dev); t0 = time.perf_counter()
    for k in range(K): step(k)
    torch.cuda.synchronize(dev); out.append((time.perf_counter() - t0) * 1e6 / K)
    if r == 5: time.sleep(0.5)
print("us/step of consecutive 20-step regions (0.5 s idle after the 6th):", " ".join(f"{x:.2f}" for x in out))

#!/usr/bin/env python3
"""Full-size identity check of an experimental ingest form against the shipped one: N = 1024, six steps, ring bytes compared.
    AGX_CHECK_ENV="AGX_INGEST_NO_FULL=1" python tools/check_stream.py      (any knob agx_create reads; round 3: AGX_INGEST_STREAM=1792)"""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "active-gym_amd")]
import torch
from active_gym import ObsPipeline
dev = torch.device("cuda:0")
N = int(os.environ.get("N", 1024))
kw = dict(num_envs=N, kind="fixed", fov_size=(30, 30), resize_to_full=True, device=dev)
a = ObsPipeline(**kw)
k, v = os.environ.get("AGX_CHECK_ENV", "AGX_INGEST_NO_FULL=1").split("=")
os.environ[k] = v
b = ObsPipeline(**kw)
del os.environ[k]
g = torch.Generator(device=dev); g.manual_seed(1)
bad = 0
for step in range(6):
    fr = torch.randint(0, 256, (N, 2, 210, 160, 3), dtype=torch.uint8, device=dev, generator=g)
    cmd = torch.full((N,), 2, dtype=torch.uint8, device=dev)
    if step == 3: cmd[::7] = 1
    a.ingest(fr, cmd); b.ingest(fr, cmd)
    ra, rb = a.stack_u8(), b.stack_u8()
    d = (ra != rb)
    nbad = int(d.sum())
    bad += nbad
    print(f"step {step}: differing ring bytes {nbad}" + (f"  envs {sorted(set(d.nonzero()[:, 0].tolist()))[:10]}" if nbad else ""), flush=True)
print("IDENTICAL" if bad == 0 else "MISMATCH")
sys.exit(0 if bad == 0 else 1)

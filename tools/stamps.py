#!/usr/bin/env python3
"""Diagnostic: per-wave s_memtime stamps of k_ingest (needs a -DAGX_STAMPS build passed via AGX_LIB)."""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "active-gym_amd")]
import numpy as np, torch
dev = torch.device("cuda:0")
N = 1024
st = torch.zeros((N * 7 * 4, 6), dtype=torch.int64, device=dev)
os.environ["AGX_DBG_PTR"] = str(st.data_ptr())
from active_gym import ObsPipeline
p = ObsPipeline(num_envs=N, kind="fixed", fov_size=(30, 30), resize_to_full=True, device=dev)
g = torch.Generator(device=dev); g.manual_seed(0)
frames = [torch.randint(0, 256, (N, 2, 210, 160, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(4)]
cmd = torch.full((N,), 2, dtype=torch.uint8, device=dev)
act = torch.rand((N, 2), device=dev) * 54
obs = torch.empty(p.obs_shape, device=dev)
for k in range(6):
    p.ingest(frames[k % 4], cmd); p.fovea(act, out=obs)
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)
d = np.diff(s[:, :5], axis=1)          # 4 segments
names = ["prologue+issue", "lum+LDS write", "barrier wait", "phase2+store"]
t0 = s[:, 0].min()
print("kernel span (cycles @100MHz memtime?):", s[:, 4].max() - t0)
for i, nme in enumerate(names):
    print(f"{nme:16s} mean {d[:, i].mean():9.0f}  p50 {np.median(d[:, i]):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}")
print("wave life mean", (s[:, 4] - s[:, 0]).mean(), " start spread p50/p99", np.percentile(s[:, 0] - t0, 50), np.percentile(s[:, 0] - t0, 99))
# timeline occupancy: how many waves alive over time
life = np.stack([s[:, 0] - t0, s[:, 4] - t0], 1)
T = life[:, 1].max()
for frac in (0.1, 0.3, 0.5, 0.7, 0.9):
    t = frac * T
    print(f"t={frac:.1f}: waves alive {(np.logical_and(life[:,0] <= t, life[:,1] > t)).sum()}")

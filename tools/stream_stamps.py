#!/usr/bin/env python3
"""Per-item timeline of the experimental k_ingest_stream12 (csrc/experiments/agx_k1_stream.h.txt + SSTAMP build, AGX_LIB):
s_memtime at the top of an item, after the wait for its first pieces, at its last row job, before / after the barrier, at its end."""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "active-gym_amd")]
import numpy as np, torch
dev = torch.device("cuda:0")
N = 1024
G = int(os.environ.get("AGX_INGEST_STREAM", "1792"))
st = torch.zeros((G * 4 * 4, 8), dtype=torch.int64, device=dev)
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
s = st.cpu().numpy().reshape(G * 4, 4, 8).astype(np.float64)
names = ["top -> first pieces there", "row jobs 0-2", "row job 3 + LDS", "barrier", "phase 2 + store"]
print(f"G = {G}; s_memtime ticks, mean over waves (p50 / p90)")
for it in range(4):
    m = (s[:, it, 0] > 0) & (s[:, it, 5] > 0)
    if not m.any(): continue
    d = np.diff(s[m, it, :6], axis=1)
    tot = s[m, it, 5] - s[m, it, 0]
    print(f"item {it}: n = {m.sum():5d}  total {tot.mean():8.0f}   " + "  ".join(f"{nm}: {d[:, i].mean():6.0f} ({np.median(d[:, i]):.0f}/{np.percentile(d[:, i], 90):.0f})" for i, nm in enumerate(names)))
    if it > 0:
        mm = m & (s[:, it - 1, 5] > 0)
        print(f"         end of item {it - 1} -> top of item {it}: {(s[mm, it, 0] - s[mm, it - 1, 5]).mean():6.0f}")
span = s[:, :, 5].max() - s[:, 0, 0][s[:, 0, 0] > 0].min()
print(f"first top -> last end over all waves: {span:.0f} ticks")
# wall-clock picture from s_memrealtime (100 MHz, one base for the chip): when do the workgroups start, when does each item end
r0, r1 = s[:, :, 6], s[:, :, 7]
ok = r0[:, 0] > 0
base = r0[ok, 0].min()
print("top of item 0 (us after the first wave): p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % tuple(np.percentile((r0[ok, 0] - base) / 100.0, q) for q in (10, 50, 90, 100)))
for it in range(4):
    m = ok & (r1[:, it] > 0)
    if m.any():
        print("item %d: top p50 %.2f us  end p50 %.2f  end p90 %.2f  end max %.2f   (duration p50 %.2f us)" % (
            it, np.median((r0[m, it] - base) / 100.0), np.median((r1[m, it] - base) / 100.0), np.percentile((r1[m, it] - base) / 100.0, 90),
            ((r1[m, it] - base) / 100.0).max(), np.median((r1[m, it] - r0[m, it]) / 100.0)))

#!/usr/bin/env python3
"""Diagnostic: per-wave s_memtime / s_memrealtime stamps + placement of k_ingest (needs a -DAGX_STAMPS build, AGX_LIB)."""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "active-gym_amd")]
import numpy as np, torch
dev = torch.device("cuda:0")
N = 1024
KERNEL = sys.argv[1] if len(sys.argv) > 1 else "ingest"
WGS = N * 7 if KERNEL == "ingest" else N * 4
st = torch.zeros((WGS * 4, 8), dtype=torch.int64, device=dev)
os.environ["AGX_DBG_PTR" if KERNEL == "ingest" else "AGX_DBG_PTR2"] = str(st.data_ptr())
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
s = st.cpu().numpy()
cyc = s[:, :5].astype(np.float64)
d = np.diff(cyc, axis=1)
for i, nme in enumerate(["issue+cmd wait", "lum+LDS write", "barrier wait", "phase2+store"] if KERNEL == "ingest" else ["loads->LDS image", "barrier wait", "H pass+barrier", "lerp+stores"]):
    print(f"{nme:16s} mean {d[:, i].mean():9.0f}  p50 {np.median(d[:, i]):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f} cycles")
life_c = cyc[:, 4] - cyc[:, 0]
rt0, rt1 = s[:, 6].astype(np.float64), s[:, 7].astype(np.float64)          # 100 MHz constant clock
life_us = (rt1 - rt0) / 100.0
print("wave life: %.0f cycles = %.2f us  -> shader clock %.2f GHz" % (life_c.mean(), life_us.mean(), life_c.mean() / life_us.mean() / 1e3))
t0 = rt0.min(); span = (rt1.max() - t0) / 100.0
print("kernel span by memrealtime: %.1f us" % span)
xcc = s[:, 5] & 0xFF; hw = s[:, 5] >> 8
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
cuid = xcc * 1000 + se * 100 + sh * 50 + cu
ids = np.unique(cuid)
print("distinct CUs seen:", len(ids), " waves per CU: min %d max %d" % (np.bincount(np.searchsorted(ids, cuid)).min(), np.bincount(np.searchsorted(ids, cuid)).max()))
ts = np.linspace(0, span, 41)[1:-1]
occ = []
for c in ids[:: max(1, len(ids) // 32)]:
    m = cuid == c
    a, b = (rt0[m] - t0) / 100.0, (rt1[m] - t0) / 100.0
    occ.append([(np.logical_and(a <= t, b > t)).sum() for t in ts])
occ = np.array(occ)
print("mean resident waves per CU over time (us : waves):")
print("  " + "  ".join(f"{t:.0f}:{o:.0f}" for t, o in zip(ts[::3], occ.mean(0)[::3])))
last_start = np.array([((rt0[cuid == c] - t0) / 100.0).max() for c in ids])
print("last wave start per CU: mean %.1f us  max %.1f us ; kernel span %.1f" % (last_start.mean(), last_start.max(), span))
# turnaround of a wave slot: from the last stamp of one wave to the first stamp of the next wave in the same (CU, SIMD, slot).
# That interval holds the end of the old wave (its last store, s_endpgm), the dispatcher's hand-over - a workgroup is admitted
# when all of its waves' slots and its LDS are free - and the new wave's start (SGPR / VGPR init, kernel-argument s_loads).
simd = (hw >> 4) & 3; wslot = hw & 0xF
slot = cuid * 64 + simd * 16 + wslot
order = np.lexsort((rt0, slot))
so, a, b = slot[order], rt0[order], rt1[order]
same = so[1:] == so[:-1]
gaps = ((a[1:] - b[:-1]) / 100.0)[same]
print("wave-slot turnaround (end stamp -> next wave's first stamp, same CU / SIMD / slot): mean %.2f us  p50 %.2f  p90 %.2f  (n = %d); "
      "distinct slots used %d (%.1f per CU)" % (gaps.mean(), np.median(gaps), np.percentile(gaps, 90), len(gaps), len(np.unique(slot)), len(np.unique(slot)) / len(ids)))
busy = (b - a).sum() / 100.0
print("slot time inside the stamps: %.0f wave-us = %.1f %% of (distinct slots x kernel span)" % (busy, 100.0 * busy / (len(np.unique(slot)) * span)))

"""H2D copy rate of the step screens (pinned -> HBM) on the launch stream and on a side stream, alone and while the ingest /
fovea kernels run: what a dedicated copy stream is worth on this box.  usage: python tools/h2d_streams.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd")]
import torch
from active_gym import ObsPipeline
dev = torch.device("cuda:0")
N = 1024
h = torch.empty((N, 2, 210, 160, 3), dtype=torch.uint8).pin_memory()
d = [torch.empty_like(h, device=dev) for _ in range(2)]
side = torch.cuda.Stream(device=dev)
p = ObsPipeline(num_envs=N, kind="fixed", fov_size=(30, 30), resize_to_full=True, device=dev)
cmd = torch.full((N,), 2, dtype=torch.uint8, device=dev)
act = torch.rand((N, 2), device=dev) * 54
obs = torch.empty(p.obs_shape, device=dev)


def t(fn, reps=6):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def copy_cur():
    d[0].copy_(h, non_blocking=True)


def copy_side():
    with torch.cuda.stream(side):
        d[0].copy_(h, non_blocking=True)


def copy_side_kernels():
    with torch.cuda.stream(side):
        d[0].copy_(h, non_blocking=True)
    for _ in range(20):
        p.ingest(d[1], cmd); p.fovea(act, out=obs)


def copy_cur_kernels():
    d[0].copy_(h, non_blocking=True)
    for _ in range(20):
        p.ingest(d[1], cmd); p.fovea(act, out=obs)


gb = h.numel() / 1e9
for name, fn in (("launch stream, alone", copy_cur), ("side stream, alone", copy_side),
                 ("launch stream + 20 steps of kernels behind it", copy_cur_kernels),
                 ("side stream + 20 steps of kernels beside it", copy_side_kernels)):
    dt = t(fn)
    print(f"{name:48s} {dt * 1e3:7.2f} ms  ({gb / dt:5.1f} GB/s if it were the copy alone)", flush=True)

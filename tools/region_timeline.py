#!/usr/bin/env python3
"""Where the driver's 20-step timed region (bench.py --steps 20 --warmup 5: ~1.1 ms) spends what it spends on top of 20 x (K1 + K2).

The region is rebuilt here exactly as bench.py runs it (200 pre-roll steps, 5 warm-up steps, synchronize, K plain steps, synchronize) and
repeated R times in one process, in three forms:
  plain    the region as bench.py times it: host elapsed only
  stamped  every launch carries its own start / stop events (agx_profile_next): the device-side timeline of the same region -
           first kernel start -> last kernel end, the gaps between consecutive kernels, each step's K1 / K2 durations
  onecall  the step as ONE C call (agx_step_fixed: both launches behind one ctypes call) instead of two pipeline calls
host elapsed - device span = what the region pays for getting the first launch onto an idle queue plus the wake-up after the last kernel.

    python tools/region_timeline.py [K=20] [R=15]
"""
import os
import statistics as st
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd")]
import torch  # noqa: E402

from active_gym import ObsPipeline  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 15
N, POOL = 1024, 8
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
pipe = ObsPipeline(num_envs=N, kind="fixed", fov_size=(30, 30), resize_to_full=True, device=dev)
g = torch.Generator(device=dev)
g.manual_seed(7)
frames = [torch.randint(0, 256, (N, 2, 210, 160, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(POOL)]
cmds = [torch.full((N,), 2, dtype=torch.uint8, device=dev) for _ in range(POOL)]
acts = [(torch.rand((N, 2), device=dev, generator=g) * 54).contiguous() for _ in range(POOL)]
obs = torch.empty(pipe.obs_shape, dtype=torch.float32, device=dev)
loc = torch.empty((N, 2), dtype=torch.int32, device=dev)


def step(k, e=None):
    i = k % POOL
    if e is not None:
        pipe.profile_next("ingest", e[0], e[1])
        pipe.profile_next("fovea", e[2], e[3])
    pipe.ingest(frames[i], cmds[i])
    pipe.fovea(acts[i], out=obs, loc_out=loc)


def step_onecall(k, e=None):
    i = k % POOL
    pipe.step_fixed(frames[i], cmds[i], acts[i], out=obs, loc_out=loc)


def region(fn, ev=None):
    for k in range(200):
        fn(k)
    for k in range(5):
        fn(k)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(K):
        fn(k, None if ev is None else ev[k])
    t1 = time.perf_counter()                    # every launch is enqueued
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    return (t2 - t0) * 1e6, (t1 - t0) * 1e6


def med(x):
    return st.median(x)


ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(K)]
for e in ev:
    for x in e:
        x.record()
torch.cuda.synchronize(dev)
region(step)                                     # first use of everything
rows = {}
for name, fn, stamped in (("plain", step, False), ("onecall", step_onecall, False), ("stamped", step, True), ("plain again", step, False)):
    H, Q, span, gaps, k1s, k2s, first = [], [], [], [], [], [], []
    for r in range(R):
        h, q = region(fn, ev if stamped else None)
        H.append(h)
        Q.append(q)
        if stamped:
            span.append(ev[0][0].elapsed_time(ev[K - 1][3]) * 1e3)
            gs = []
            for k in range(K):
                gs.append(ev[k][1].elapsed_time(ev[k][2]) * 1e3)                    # K1 end -> K2 start
                if k + 1 < K:
                    gs.append(ev[k][3].elapsed_time(ev[k + 1][0]) * 1e3)            # K2 end -> next K1 start
            gaps.append(sum(gs))
            k1s.append([ev[k][0].elapsed_time(ev[k][1]) * 1e3 for k in range(K)])
            k2s.append([ev[k][2].elapsed_time(ev[k][3]) * 1e3 for k in range(K)])
    line = f"{name:12s} host elapsed: median {med(H):7.1f} us  min {min(H):7.1f}  max {max(H):7.1f}  ({med(H) / K:.2f} us / step); all {K} steps enqueued after {med(Q):6.1f} us"
    if stamped:
        kern = [sum(a) + sum(b) for a, b in zip(k1s, k2s)]
        line += (f"\n{'':12s} device span (first K1 start -> last K2 end): median {med(span):7.1f} us = kernels {med(kern):7.1f} + gaps {med(gaps):5.1f};"
                 f" host elapsed - device span = {med([h - s for h, s in zip(H, span)]):5.1f} us")
        k1m = [med([r_[k] for r_ in k1s]) for k in range(K)]
        k2m = [med([r_[k] for r_ in k2s]) for k in range(K)]
        line += "\n" + " " * 12 + " K1 per step (us): " + " ".join(f"{x:.1f}" for x in k1m)
        line += "\n" + " " * 12 + " K2 per step (us): " + " ".join(f"{x:.1f}" for x in k2m)
    print(line, flush=True)

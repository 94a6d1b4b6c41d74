import os, sys, time
sys.path[:0] = ["/root/repo/active-gym_amd", "/root/repo"]
import torch, bench
dev = torch.device("cuda:0")
n = 1024
pipe = bench.make_pipeline("fixed", n, dev)
frames, cmds, acts = bench.synth_inputs(torch, dev, n, 8, 1234)
obs = torch.empty(pipe.obs_shape, dtype=torch.float32, device=dev); loc = torch.empty((n, 2), dtype=torch.int32, device=dev)
def run(K, events):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)] if events else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        i = k % 8
        if events: ev[k][0].record()
        pipe.ingest(frames[i], cmds[i])
        if events: ev[k][1].record()
        pipe.fovea(acts[i], out=obs, loc_out=loc)
        if events: ev[k][2].record()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6
for ev in (False, True, False, True):
    run(50, ev)
    print("events" if ev else "plain ", "enqueue us/step %.1f   total us/step %.1f" % run(300, ev))

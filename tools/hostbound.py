"""Host enqueue cost per step against GPU time per step (N = 1024, device-resident inputs): the two-call form bench.py times
(pipe.ingest + pipe.fovea), the one-call form (pipe.step_fixed = agx_step_fixed), and the same on compact input screens."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), REPO]
import torch, bench
dev = torch.device("cuda:0")
n = 1024
pipe = bench.make_pipeline("fixed", n, dev)
frames, cmds, acts = bench.synth_inputs(torch, dev, n, 8, 1234)
rows = torch.from_numpy(pipe.source_rows()).to(dev).long()
cframes = [f.index_select(2, rows).contiguous() for f in frames]
obs = torch.empty(pipe.obs_shape, dtype=torch.float32, device=dev); loc = torch.empty((n, 2), dtype=torch.int32, device=dev)
def run(K, form):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        i = k % 8
        if form == "two calls":
            pipe.ingest(frames[i], cmds[i]); pipe.fovea(acts[i], out=obs, loc_out=loc)
        elif form == "one call":
            pipe.step_fixed(frames[i], cmds[i], acts[i], out=obs, loc_out=loc)
        elif form == "two calls, compact":
            pipe.ingest_compact(cframes[i], cmds[i]); pipe.fovea(acts[i], out=obs, loc_out=loc)
        else:
            pipe.ingest_compact(cframes[i], cmds[i])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6
for rep in range(2):
    for form in ("two calls", "one call", "two calls, compact", "ingest_compact only"):
        run(50, form)
        print("%-22s enqueue us/step %.1f   total us/step %.1f" % ((form,) + run(400, form)), flush=True)

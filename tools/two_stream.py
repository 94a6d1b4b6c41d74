"""Experiment: the 1024-env step as two 512-env halves on two HIP streams (own contexts), so that one half's K2 and
drain overlap the other half's K1.  Prints env steps/s for 1 stream x 1024 and 2 streams x 512."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), REPO]
import torch, bench
dev = torch.device("cuda:0")
def run(parts, K=400, prio=False):
    n = 1024 // parts
    pipes = [bench.make_pipeline("fixed", n, dev) for _ in range(parts)]
    data = [bench.synth_inputs(torch, dev, n, 8, 1234 + i) for i in range(parts)]
    obs = [torch.empty(p.obs_shape, dtype=torch.float32, device=dev) for p in pipes]
    loc = [torch.empty((n, 2), dtype=torch.int32, device=dev) for _ in pipes]
    streams = [torch.cuda.Stream(priority=(-1 if (prio and i == 0) else 0)) for i in range(parts)] if parts > 1 else [torch.cuda.current_stream()]
    def go(K):
        for k in range(K):
            i = k % 8
            for j in range(parts):
                with torch.cuda.stream(streams[j]):
                    f, c, a = data[j]
                    pipes[j].ingest(f[i], c[i]); pipes[j].fovea(a[i], out=obs[j], loc_out=loc[j])
    go(40); torch.cuda.synchronize()
    t = time.perf_counter(); go(K); torch.cuda.synchronize(); dt = time.perf_counter() - t
    return 1024 * K / dt
for rep in range(2):
    print("1 stream  x1024: %.2f M env steps/s" % (run(1) / 1e6))
    print("2 streams x512 : %.2f M" % (run(2) / 1e6))
    print("2 streams x512 (one high priority): %.2f M" % (run(2, prio=True) / 1e6))
    print("4 streams x256 : %.2f M" % (run(4) / 1e6), flush=True)

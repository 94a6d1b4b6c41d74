"""Experiment: the device-resident step (K1 + fovea kernel) captured into a hipGraph and replayed, against the same launches
issued one by one.  The C ABI's step entry points only enqueue kernels on the caller's stream, so a caller may capture them
(hipStreamBeginCapture / torch.cuda.graph); the context flips its double-buffered head / fov state on the host at every call,
so a captured sequence must hold an EVEN number of steps to leave that parity where the replay expects it.

Prints, per batch size: us per step for eager launches from Python, for the replay of an 8-step graph (one pass over the input
pool) and of a 2-step graph, and checks that 16 replayed steps leave the ring / fov_loc / observations bit-identical to 16 eager
steps on a second context."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), REPO]
import torch, bench
dev = torch.device("cuda:0")
POOL = 8


def build(n, kind="fixed"):
    pipe = bench.make_pipeline(kind, n, dev)
    frames, cmds, acts = bench.synth_inputs(torch, dev, n, POOL, 1234)
    obs = torch.empty(pipe.obs_shape, dtype=torch.float32, device=dev)
    loc = torch.empty((n, 2), dtype=torch.int32, device=dev)

    def step(k):
        i = k % POOL
        pipe.ingest(frames[i], cmds[i])
        pipe.fovea(acts[i], out=obs, loc_out=loc)
    return pipe, step, obs, loc


def capture(step, steps):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for k in range(2):                  # warm the side stream (the steps count: two of them keep the parity)
            step(k)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for k in range(steps):
            step(2 + k)
    return g


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


for n in (64, 256, 1024, 4096):
    # parity: 2 warm steps + 16 replayed == 18 eager
    pa, sa, oa, la = build(n)
    pb, sb, ob, lb = build(n)
    g2 = capture(sa, 2)
    # the 2-step graph holds pool entries 2, 3: replay it 8 times; eager runs the same sequence
    for _ in range(8):
        g2.replay()
    for k in range(2):
        sb(k)
    for _ in range(8):
        sb(2); sb(3)
    torch.cuda.synchronize()
    same = bool(torch.equal(pa.stack_u8(), pb.stack_u8()) and torch.equal(la, lb) and torch.equal(oa.view(torch.int32), ob.view(torch.int32)))
    # timing
    pe, se, _, _ = build(n)
    K = 400
    for k in range(100):
        se(k)
    t_eager = timed(lambda: [se(k) for k in range(K)], 3) / K
    pg, sg, _, _ = build(n)
    g8 = capture(sg, POOL)
    for _ in range(20):
        g8.replay()
    t_g8 = timed(lambda: [g8.replay() for _ in range(K // POOL)], 3) / K
    t_g2 = timed(lambda: [g2.replay() for _ in range(K // 2)], 3) / K
    print("N = %5d: eager %7.2f us/step (%6.2f M env steps/s) | 8-step graph %7.2f us/step (%6.2f M) | 2-step graph %7.2f us/step | replay == eager: %s"
          % (n, t_eager * 1e6, n / t_eager / 1e6, t_g8 * 1e6, n / t_g8 / 1e6, t_g2 * 1e6, same), flush=True)
    del pa, pb, pe, pg

"""Re-measure the experiments build's step forms on today's kernels (libagx_exp.so): the two stand-alone launches against the fused
heterogeneous launch (AGX_STEP_FUSED=1: ingest bands + fovea of the three untouched ring slots in one grid, the written slot's fovea
after), the split step and one workgroup per env - pipe.step_fixed in a loop, N = 1024, 400 steps, each form in a child process.
    AGX_LIB=active-gym_amd/lib/libagx_exp.so python tools/fused_retest.py"""
import os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), REPO]

def child(obs_bufs):
    import torch, bench
    dev = torch.device("cuda:0")
    n = 1024
    pipe = bench.make_pipeline("fixed", n, dev)
    frames, cmds, acts = bench.synth_inputs(torch, dev, n, 8, 1234)
    obs = [torch.empty(pipe.obs_shape, dtype=torch.float32, device=dev) for _ in range(obs_bufs)]
    loc = torch.empty((n, 2), dtype=torch.int32, device=dev)
    def run(K):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K):
            i = k % 8
            pipe.step_fixed(frames[i], cmds[i], acts[i], out=obs[k % obs_bufs], loc_out=loc)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e6
    run(300)
    print("%.2f us/step" % min(run(400), run(400)), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2])); sys.exit(0)
    assert "libagx_exp" in os.environ.get("AGX_LIB", ""), "run with AGX_LIB=<...>/libagx_exp.so"
    for rep in range(2):
        for bufs in (1, 3):
            for name, env in (("two launches", {}), ("AGX_STEP_FUSED=1", {"AGX_STEP_FUSED": "1"}), ("AGX_STEP_FUSED=2", {"AGX_STEP_FUSED": "2"}),
                              ("AGX_STEP_FUSED=3", {"AGX_STEP_FUSED": "3"})):
                e = dict(os.environ); e.update(env)
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(bufs)], env=e, capture_output=True, text=True).stdout.strip().split("\n")[-1]
                print(f"obs buffers {bufs}  {name:18s} {out}", flush=True)

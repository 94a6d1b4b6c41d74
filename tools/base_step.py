"""AtariBaseEnv's device step (no fovea): K1 ingest + K0 `k_full` (ring -> stack order, f32 k/255: atari_env.py:143 np.stack) at
N = 1024, and `k_full` alone back to back.  k_full moves 4 x 7,056 B in and 4 x 28,224 B out per env: 144.5 MB per launch at N = 1024."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), REPO]
import torch, bench
from active_gym import ObsPipeline
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pipe = ObsPipeline(num_envs=n, kind="base", obs_size=(84, 84), frame_stack=4, device=dev)
frames, cmds, _ = bench.synth_inputs(torch, dev, n, 8, 1234)
obs = [torch.empty(pipe.full_shape, dtype=torch.float32, device=dev) for _ in range(3)]
by = n * 4 * 7056 * 5


def timed(fn, K=300, reps=3):
    best = 1e9
    for _ in range(reps):
        for k in range(100):
            fn(k)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for k in range(K):
            fn(k)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / K)
    return best


def step(k, P=1):
    pipe.ingest(frames[k % 8], cmds[k % 8])
    pipe.observe_full(obs[k % P])


t_step = timed(step)
t_step3 = timed(lambda k: step(k, 3))
t_full = timed(lambda k: pipe.observe_full(obs[0]))
t_full3 = timed(lambda k: pipe.observe_full(obs[k % 3]))
t_ing = timed(lambda k: pipe.ingest(frames[k % 8], cmds[k % 8]))
print("N = %d: K1 + k_full %.2f us per step (%.2f M env steps/s); output through 3 buffers %.2f us" % (n, t_step * 1e6, n / t_step / 1e6, t_step3 * 1e6))
print("k_full alone, back to back: %.2f us = %.2f TB/s = %.3f of 8 TB/s (one output buffer); %.2f us = %.3f through 3 buffers (347 MB)"
      % (t_full * 1e6, by / t_full / 1e12, by / t_full / 8e12, t_full3 * 1e6, by / t_full3 / 8e12))
print("K1 alone, back to back: %.2f us" % (t_ing * 1e6))

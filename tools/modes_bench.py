"""The observation modes the headline sub-runs do not time, at N = 1024 (device-resident, 300 steps, best of 3): fixed and flexible
fovea in mask-out and raw-crop mode (fov_env.py:176-185, 289-298), and the DMC pixel front end (K1b + K2, frame_stack 3, dmc_env.py:175-186).
Per mode: us per step of ingest + fovea, and of the fovea kernel alone back to back, with its algorithmic bytes."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), REPO]
import torch, bench
from active_gym import ObsPipeline
dev = torch.device("cuda:0")
n = 1024
frames, cmds, acts = bench.synth_inputs(torch, dev, n, 8, 1234)


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


g = torch.Generator(device=dev)
g.manual_seed(5)
types = [torch.randint(0, 2, (n,), dtype=torch.int32, device=dev, generator=g) for _ in range(8)]
facts = [torch.where(types[i][:, None] == 1, torch.randint(10, 61, (n, 2), device=dev, generator=g).float(), acts[i]).contiguous() for i in range(8)]
for kind in ("fixed", "flexible"):
    for mode in ("mask_out", "raw"):
        pipe = ObsPipeline(num_envs=n, kind=kind, obs_size=(84, 84), frame_stack=4, fov_size=(30, 30), fov_init_loc=(0, 0),
                           sensory_action_mode="absolute", mask_out=(mode == "mask_out"), resize_to_full=False, device=dev)
        obs = torch.empty(pipe.obs_shape, dtype=torch.float32, device=dev)
        loc = torch.empty((n, 2), dtype=torch.int32, device=dev)
        res = torch.empty((n, 2), dtype=torch.int32, device=dev)
        if kind == "fixed":
            fov = lambda k: pipe.fovea(acts[k % 8], out=obs, loc_out=loc)
        else:
            fov = lambda k: pipe.fovea(facts[k % 8], action_type=types[k % 8], out=obs, loc_out=loc, res_out=res)

        def step(k):
            pipe.ingest(frames[k % 8], cmds[k % 8])
            fov(k)
        t_step, t_fov = timed(step), timed(fov)
        by = pipe.algorithmic_bytes("fovea")
        print("%-8s %-8s obs %s: step %.2f us (%.2f M env steps/s) | fovea kernel back to back %.2f us, %.1f MB algorithmic = %.3f of 8 TB/s"
              % (kind, mode, tuple(pipe.obs_shape), t_step * 1e6, n / t_step / 1e6, t_fov * 1e6, by / 1e6, by / t_fov / 8e12), flush=True)
        pipe.close()
# DMC pixel front end: obs-sized RGB renders in, frame_stack 3, fixed fovea resized to full
pipe = ObsPipeline(num_envs=n, kind="fixed", obs_size=(84, 84), frame_stack=3, fov_size=(30, 30), fov_init_loc=(0, 0),
                   sensory_action_mode="absolute", resize_to_full=True, device=dev)
rgb = [torch.randint(0, 256, (n, 84, 84, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(8)]
one = torch.ones((n,), dtype=torch.uint8, device=dev)
obs = torch.empty(pipe.obs_shape, dtype=torch.float32, device=dev)
loc = torch.empty((n, 2), dtype=torch.int32, device=dev)


def dstep(k):
    pipe.ingest_rgb(rgb[k % 8], one)
    pipe.fovea(acts[k % 8], out=obs, loc_out=loc)


t_step = timed(dstep)
t_ing = timed(lambda k: pipe.ingest_rgb(rgb[k % 8], one))
t_fov = timed(lambda k: pipe.fovea(acts[k % 8], out=obs, loc_out=loc))
print("DMC      resize   obs %s: step %.2f us (%.2f M env steps/s) | k_ingest_rgb back to back %.2f us (%.1f MB), fovea %.2f us (%.1f MB = %.3f of 8 TB/s)"
      % (tuple(pipe.obs_shape), t_step * 1e6, n / t_step / 1e6, t_ing * 1e6, pipe.algorithmic_bytes("ingest_rgb") / 1e6, t_fov * 1e6,
         pipe.algorithmic_bytes("fovea") / 1e6, pipe.algorithmic_bytes("fovea") / t_fov / 8e12))

"""End to end through the drop-in API: AtariVecEnv.step = host runner (emulators on the CPU cores) -> pinned
staging -> H2D -> ingest + fovea kernels -> observations (device tensors), plus the parts on their own
(runner only, H2D only).  Host-bound by construction: this is the PCIe/emulator-inclusive rate DESIGN.md
quotes beside the device-resident headline.  Best of `reps` repeats (the box's host cores are shared)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd")]
import numpy as np
import torch
from active_gym import AtariEnvArgs, AtariVecEnv
from active_gym.native_runner import NativeHostRunner

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = 3


def best(fn):
    out = []
    for _ in range(reps):
        t = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / steps)
    return min(out)


def mk(src, chunk, workers, fmt="rgb"):
    return AtariEnvArgs(frame_format=fmt, game="breakout", seed=1, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0),
                        sensory_action_mode="absolute", resize_to_full=True, frame_source=src, device="cuda",
                        num_workers=workers, h2d_chunk_envs=chunk)


print(f"host cpus {os.cpu_count()}", flush=True)
m = np.zeros(N, np.int64)
for w in (8, 16, 32, 64, 128, 256):
    r = NativeHostRunner(mk("native", 0, w), N, workers=w, backend="scripted")
    r.reset()
    dt = best(lambda: r.step(m))
    print(f"runner only  workers={w:3d}: {dt * 1e3:6.2f} ms/step  {N / dt:10,.0f} env steps/s", flush=True)
    r.close()
h = torch.empty((N, 2, 210, 160, 3), dtype=torch.uint8).pin_memory()
d = torch.empty_like(h, device="cuda")
dt = best(lambda: d.copy_(h, non_blocking=True))
print(f"H2D only ({h.numel() / 1e6:.0f} MB): {dt * 1e3:.2f} ms/step = {h.numel() / dt / 1e9:.1f} GB/s", flush=True)
act = {"motor_action": m, "sensory_action": np.full((N, 2), 20.0, np.float32)}
for src, chunk, w, fmt in (("native", 0, 64, "rgb"), ("native", 128, 64, "rgb"), ("native", 256, 64, "rgb"), ("native", 0, 64, "gray"),
                           ("native", 128, 64, "gray"), ("native", 256, 64, "gray"), ("native", 128, 128, "gray"),
                           ("synthetic", 0, 16, "rgb")):
    env = AtariVecEnv(mk(src, chunk, w, fmt), N, kind="fixed")
    env.reset()
    if src == "synthetic":
        steps, reps = 3, 1
    env.step(act)
    dt = best(lambda: env.step(act))
    print(f"e2e {src:9s} {fmt:4s} workers={w:3d} chunk={chunk:4d} N={N}: {dt * 1e3:6.2f} ms/step {N / dt:10,.0f} env steps/s", flush=True)
    env.close()

"""What host (NumPy) outputs cost at N = 1024: AtariVecEnv.step with fresh arrays (device -> pageable), with pinned host buffers
(args.copy_obs = False) and with device outputs; the D2H copy alone, pageable and pinned."""
import os, sys, time
sys.path[:0] = ["/root/repo/active-gym_amd", "/root/repo"]
import numpy as np, torch
from active_gym import AtariEnvArgs, AtariVecEnv
N = 1024
act = {"motor_action": np.zeros(N, np.int64), "sensory_action": np.full((N, 2), 20.0, np.float32)}
for dev, copy, nbuf in ((None, True, 0), (None, True, 4), (None, False, 0), ("cuda:0", False, 0)):
    args = AtariEnvArgs(copy_obs=copy, host_obs_buffers=nbuf, frame_format="gray", game="breakout", seed=1, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0),
                        sensory_action_mode="absolute", resize_to_full=True, frame_source="native", device=dev,
                        scripted_lives=3, scripted_p_life=6, scripted_p_over=1)
    env = AtariVecEnv(args, N, kind="fixed")
    env.reset()
    K = 30 if (dev is None and copy and not nbuf) else 200
    for _ in range(20): env.step(act)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(K): o = env.step(act)[0]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
    print(f"device={dev} copy_obs={copy} host_obs_buffers={nbuf}: {dt*1e3:.2f} ms/step, {N/dt/1e6:.3f} M env steps/s, obs {type(o).__name__}", flush=True)
    env.close()
# the D2H copy alone: pageable .cpu() vs pinned non_blocking
x = torch.empty((N, 4, 84, 84), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): y = x.cpu()
print(f".cpu() (pageable): {(time.perf_counter()-t)/5*1e3:.2f} ms")
h = torch.empty(x.shape, dtype=x.dtype).pin_memory()
h.copy_(x, non_blocking=True); torch.cuda.synchronize()          # (the first copy into a fresh pinned buffer maps it: not timed)
t = time.perf_counter()
for _ in range(5): h.copy_(x, non_blocking=True); torch.cuda.synchronize()
print(f"pinned copy: {(time.perf_counter()-t)/5*1e3:.2f} ms")
t = time.perf_counter()
for _ in range(5): z = h.numpy().copy()
print(f"numpy copy of the pinned buffer: {(time.perf_counter()-t)/5*1e3:.2f} ms")

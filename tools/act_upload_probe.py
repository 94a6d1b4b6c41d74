"""What host-resident (NumPy) sensory actions cost the e2e step: AtariVecEnv (native runner + loop, gray compact screens, device
observations, N = 1024) stepped with the sensory action as a device tensor and as a NumPy array, windows of 300 steps."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), REPO]
import numpy as np, torch
from active_gym import AtariEnvArgs, AtariVecEnv
N = 1024
dev = "cuda:0"
fmt = sys.argv[1] if len(sys.argv) > 1 else "gray"
args = AtariEnvArgs(frame_format=fmt, game="breakout", seed=1, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0),
                    sensory_action_mode="absolute", resize_to_full=True, frame_source="native", device=dev,
                    scripted_lives=3, scripted_p_life=6, scripted_p_over=1)
env = AtariVecEnv(args, N, kind="fixed")
env.reset()
motor = np.zeros(N, np.int64)
acts = {"device f32 tensor": torch.full((N, 2), 20.0, dtype=torch.float32, device=dev),
        "numpy f32": np.full((N, 2), 20.0, np.float32),
        "numpy int64 (action_space.sample() style)": np.full((N, 2), 20, np.int64)}
for rep in range(2):
    for name, a in acts.items():
        act = {"motor_action": motor, "sensory_action": a}
        for _ in range(20):
            env.step(act)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            for _ in range(300):
                env.step(act)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) / 300)
        print("%-45s %.3f ms per step best, %.3f median = %.3f M env steps/s" % (name, min(ts) * 1e3, sorted(ts)[1] * 1e3, N / sorted(ts)[1] / 1e6), flush=True)
env.close()

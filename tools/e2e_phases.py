"""Where a gray-screen e2e step's host time goes (N = 1024, native runner, compact staging, resets on): the emulators alone
(agxr_step), the whole AtariVecEnv.step on the host (enqueue only), the step with a synchronisation - per worker count.
    python tools/e2e_phases.py [gray|rgb]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), REPO]
import numpy as np
import torch
from active_gym import AtariEnvArgs, AtariVecEnv
fmt = sys.argv[1] if len(sys.argv) > 1 else "gray"
N = 1024
act = {"motor_action": np.zeros(N, np.int64), "sensory_action": np.full((N, 2), 20.0, np.float32)}
for workers in (12, 16, 24, 32):
    for compact, loop in ((True, True), (True, False), (False, True)):
        args = AtariEnvArgs(frame_format=fmt, game="breakout", seed=1, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0),
                            sensory_action_mode="absolute", resize_to_full=True, frame_source="native", device="cuda:0",
                            num_workers=workers, h2d_chunk_envs=0, scripted_lives=3, scripted_p_life=6, scripted_p_over=1,
                            compact_rows=compact, native_loop=loop)
        env = AtariVecEnv(args, N, kind="fixed")
        env.reset()
        for _ in range(4):
            env.step(act)
        torch.cuda.synchronize()
        # (a) the emulators alone
        m = np.zeros(N, np.int64)
        t_run = float("nan")
        if env.runner.frames is not None:                 # (the native loop owns the staging: no stand-alone runner step there)
            t = time.perf_counter()
            for _ in range(20):
                env.runner.step(m)
            t_run = (time.perf_counter() - t) / 20
        # (b) whole steps, best / median of 5 repeats of 24
        reps = []
        for _ in range(5):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(24):
                env.step(act)
            t_host = (time.perf_counter() - t) / 24
            torch.cuda.synchronize()
            reps.append(((time.perf_counter() - t) / 24, t_host))
        reps.sort()
        print(f"{fmt} workers={workers:3d} compact={int(compact)} native_loop={int(env._loop is not None)} pinned={sorted(set(env.runner.worker_cpus))[:4]}.. rows={env.runner.rows} "
              f"runner alone {t_run * 1e3:5.2f} ms | step best {reps[0][0] * 1e3:5.2f} ms (host enqueue {reps[0][1] * 1e3:5.2f}) "
              f"median {reps[2][0] * 1e3:5.2f} ms -> {N / reps[0][0] / 1e6:.3f} / {N / reps[2][0] / 1e6:.3f} M env steps/s", flush=True)
        env.close()

"""Per-step cost of the single-env drop-in API (N = 1 core): AtariFixedFovealEnv / AtariBaseEnv over the scripted emulator,
NumPy outputs like the reference.  Prints env steps/s and a coarse breakdown (cProfile top entries with --profile)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), os.path.join(REPO, "tests")]
import numpy as np
import torch
import active_gym
from lcg_ale import LcgALE

def mk(kind, **kw):
    args = active_gym.AtariEnvArgs(game="breakout", seed=1, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0),
                                   sensory_action_mode="absolute", resize_to_full=True, peripheral_res=(20, 20),
                                   frame_source=kw.pop("src", "native"), num_workers=1, **kw)
    return {"base": active_gym.AtariBaseEnv, "fixed": active_gym.AtariFixedFovealEnv,
            "flexible": active_gym.AtariFlexibleFovealEnv, "peripheral": active_gym.AtariFixedFovealPeripheralEnv}[kind](args)

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for kind in ("base", "fixed", "flexible", "peripheral"):
    env = mk(kind)
    env.reset()
    act = 0 if kind == "base" else {"motor_action": 0, "sensory_action": np.array([10, 20])}
    if kind == "flexible":
        act["sensory_action_type"] = 0
    for _ in range(50):
        o, r, d, t, i = env.step(act)
        if d: env.reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        o, r, d, t, i = env.step(act)
        if d: env.reset()
    dt = (time.perf_counter() - t0) / steps
    print(f"{kind:10s} single env: {dt * 1e6:7.1f} us/step  {1 / dt:8.0f} env steps/s", flush=True)
    env.close()
if "--profile" in sys.argv:
    import cProfile, pstats
    env = mk("fixed"); env.reset()
    act = {"motor_action": 0, "sensory_action": np.array([10, 20])}
    pr = cProfile.Profile(); pr.enable()
    for _ in range(1000):
        o, r, d, t, i = env.step(act)
        if d: env.reset()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)

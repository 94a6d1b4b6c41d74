#!/usr/bin/env python3
"""Quick start: the reference's usage pattern (README.md:38-56 of elicassion/active-gym) on the MI355X-native
package, single env and batched.  Runs with the procedural stand-in emulator when ALE is not installed:

    python examples/quickstart.py [--ale] [--envs 256]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "active-gym_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from active_gym import AtariEnvArgs, AtariFixedFovealEnv, AtariVecEnv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ale", action="store_true", help="use real ALE through atari_py / ale_py instead of the stand-in")
    ap.add_argument("--native", action="store_true",
                    help="batched part: the C++ host runner (libagx_runner.so; scripted emulator, or real ALE with --ale) with "
                         "grayscale screens under the native step loop (one C call per vector step, autoreset inside)")
    ap.add_argument("--envs", type=int, default=256)
    a = ap.parse_args()
    src = "ale" if a.ale else "synthetic"

    # --- one env, exactly the reference's call pattern
    args = AtariEnvArgs(game="breakout", seed=42, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(27, 27),
                        sensory_action_mode="absolute", resize_to_full=True, frame_source=src)
    env = AtariFixedFovealEnv(args)
    obs, info = env.reset()
    print("single env:", obs.shape, obs.dtype, "fov_loc", info["fov_loc"])
    for _ in range(5):
        action = {"motor_action": env.action_space["motor_action"].sample(),
                  "sensory_action": np.random.uniform(0, 54, size=2)}
        obs, reward, done, truncated, info = env.step(action)
        if done:
            obs, info = env.reset()
    print("  after 5 steps: ep_len", info["ep_len"], "cumulative reward", info["reward"], "fov_loc", info["fov_loc"])
    env.close()

    # --- N envs on one GPU, observations stay in HBM
    extra = {}
    if a.native:
        src = "native:ale" if a.ale else "native"
        extra = dict(frame_format="gray")        # compact staging and the native step loop are the defaults of this path
    args = AtariEnvArgs(game="boxing", seed=0, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0),
                        sensory_action_mode="relative", sensory_action_space=(-10.0, 10.0), resize_to_full=True,
                        frame_source=src, device="cuda", **extra)
    venv = AtariVecEnv(args, num_envs=a.envs, kind="fixed")
    obs, infos = venv.reset()
    t0 = time.perf_counter()
    steps = 20
    for _ in range(steps):
        act = {"motor_action": np.random.randint(0, venv.single_motor_space.n, a.envs),
               "sensory_action": torch.randn(a.envs, 2, device="cuda") * 5}
        obs, rew, term, trunc, infos = venv.step(act)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loop = "native step loop" if venv._loop is not None else "Python step loop"
    print(f"vec env: {a.envs} envs, obs {tuple(obs.shape)} on {obs.device}; {a.envs * steps / dt:.0f} env steps/s end to end, {loop} "
          f"(host emulators + PCIe + kernels; frame source: {src}"
          + ("" if a.native else "; these emulators run in Python threads, try --native") + ")")
    venv.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Soak: AtariVecEnv on the native runner for a time budget per configuration, with episode ends (autoreset) in every step - host RSS, device
memory in use (hipMemGetInfo through torch) and steps/s sampled along the way.  A step call allocates nothing on the native loop and only
its outputs' Python objects on the Python loop, so all three must be flat.  The baseline is taken after 1,500 warm-up steps: the first
few hundred steps of the first env of a process grow its RSS by ~190 MB once or twice (the second increment has been seen as late as 20 k steps in: profiles/r04_soak_final_tree.txt) under either step loop - not in the malloc arena (mallinfo2
stays flat, profiles/r04_soak_heap_stats.txt: a mapping of the HIP runtime's own), not repeated by a second env in the same process -
printed as "warm-up", not counted as growth.

    python tools/soak.py [seconds per configuration = 60]
"""
import os
import resource
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from active_gym import AtariEnvArgs, AtariVecEnv  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device("cuda:0")
N = 256


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * resource.getpagesize() / 1e6


def used_mb():
    free, total = torch.cuda.mem_get_info(dev)
    return (total - free) / 1e6


worst = 0.0
n_jumps_total = 0
for kind, fmt, native_loop, out in (("fixed", "gray", True, "device"), ("flexible", "rgb", True, "device"), ("peripheral", "gray", False, "device"),
                                    ("fixed", "gray", False, "numpy")):
    kw = {}
    if out == "device":
        kw["device"] = str(dev)
    args = AtariEnvArgs(game="boxing", seed=3, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute",
                        resize_to_full=True, frame_source="native", frame_format=fmt, scripted_lives=3, scripted_p_life=20, scripted_p_over=5,
                        native_loop=native_loop, **({"peripheral_res": (20, 20)} if kind == "peripheral" else {}), **kw)
    env = AtariVecEnv(args, N, kind=kind)
    env.reset()
    act = {"motor_action": np.zeros(N, np.int64), "sensory_action": np.full((N, 2), 20.0, np.float32)}
    if kind == "flexible":
        act["sensory_action_type"] = np.zeros(N, np.int64)
    r0 = rss_mb()
    for _ in range(1500):                   # incl. the process's one-time heap growth (see the docstring)
        env.step(act)
    torch.cuda.synchronize(dev)
    base = (rss_mb(), used_mb())
    warm = base[0] - r0
    t0 = t_last = time.perf_counter()
    steps = dones = 0
    samples = []
    while time.perf_counter() - t0 < budget:
        for _ in range(500):
            dones += int(env.step(act)[2].sum())
        steps += 500
        torch.cuda.synchronize(dev)
        now = time.perf_counter()
        samples.append((steps, rss_mb() - base[0], used_mb() - base[1], 500 * N / (now - t_last)))
        t_last = now
    env.close()
    name = f"{kind:10s} {fmt:4s} {'native loop' if native_loop else 'python loop'} {out:6s}"
    # the runtime's one-time mappings (~190 MB each, at most two per process, see the docstring) show as JUMPS between two consecutive
    # samples; a leak shows as drift.  Jumps are reported with the step they came at and taken out of the growth figure; more than two
    # per process (warm-up included) count as growth after all.
    jumps = [(samples[k][0], samples[k][1] - samples[k - 1][1]) for k in range(1, len(samples)) if samples[k][1] - samples[k - 1][1] > 100.0]
    n_jumps_total += len(jumps) + int(round(max(warm, 0.0) / 190.0))
    q = len(samples) // 4
    d_rss = samples[-1][1] - samples[q][1] - sum(j for st, j in jumps if st > samples[q][0])
    if n_jumps_total > 2:
        d_rss = samples[-1][1] - samples[q][1]
    d_dev = samples[-1][2] - samples[q][2]
    worst = max(worst, d_rss, d_dev)
    if jumps:
        print(f"{name}: one-time RSS jumps after the baseline: " + ", ".join(f"{j:+.1f} MB at step {st}" for st, j in jumps) + f" ({n_jumps_total} in this process so far)")
    print(f"{name}: warm-up {warm:+.1f} MB; {steps} steps, {dones} episode ends; host RSS vs start {samples[len(samples) // 4][1]:+.1f} MB at a quarter -> {samples[-1][1]:+.1f} MB at the end;"
          f" device memory {samples[len(samples) // 4][2]:+.1f} -> {samples[-1][2]:+.1f} MB; env steps/s first / last sample {samples[0][3] / 1e3:.0f} k / {samples[-1][3] / 1e3:.0f} k",
          flush=True)
print("growth over the last three quarters of any run, one-time jumps (at most two per process) taken out: %.1f MB" % worst)
sys.exit(1 if worst > 64 else 0)

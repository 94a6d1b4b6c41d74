"""Host-side rate of the frame producers (no GPU): Python runner (thread pool over Python emulators) vs the
native C++ runner (libagx_runner.so, scripted emulator).  env steps/s = N * steps / wall."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), os.path.join(REPO, "tests")]
import numpy as np
from active_gym.native_runner import NativeHostRunner
from active_gym.runner import AtariHostRunner
from lcg_ale import LcgALE


class A:
    def __init__(self, **k): self.__dict__.update(k)


N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
common = dict(game="g", seed=1, action_repeat=4, clip_reward=False, max_episode_length=108e3)
m = np.zeros(N, np.int64)
for name, r in (("native", NativeHostRunner(A(**common), N, backend="scripted")),
                ("python", AtariHostRunner(A(frame_source=lambda a, i: LcgALE(1 + i), **common), N))):
    r.reset()
    t = time.perf_counter()
    for _ in range(steps):
        _, d, _, _ = r.step(m)
        if d.any():
            r.reset(np.nonzero(d)[0])
    dt = time.perf_counter() - t
    print(f"{name}: N={N} {N * steps / dt:,.0f} env steps/s ({dt / steps * 1e3:.1f} ms/step) on {os.cpu_count()} cpus")
    r.close()

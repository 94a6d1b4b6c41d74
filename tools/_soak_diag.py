import os, sys, time, resource, gc
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd")]
import numpy as np, torch, ctypes as C
from active_gym import AtariEnvArgs, AtariVecEnv
from active_gym import native_loop as NL
from active_gym.pipeline import _DT
dev = torch.device("cuda:0"); N = 256
mode = sys.argv[1]
def rss():
    with open("/proc/self/statm") as f: return int(f.read().split()[1]) * resource.getpagesize() / 1e6
args = AtariEnvArgs(game="boxing", seed=3, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute",
                    resize_to_full=True, frame_source="native", frame_format="gray", scripted_lives=3, scripted_p_life=20, scripted_p_over=5,
                    device=str(dev))
env = AtariVecEnv(args, N, kind="fixed"); env.reset()
lp = env._loop
motor = np.zeros(N, np.int32); sens = torch.full((N, 2), 20.0, device=dev)
act = {"motor_action": np.zeros(N, np.int64), "sensory_action": sens}
res = NL.AgxLoopResult(); st = lp._stream()
def raw():
    rc = lp._lib.agx_loop_step(lp._h, motor.ctypes.data, C.c_void_p(sens.data_ptr()), int(_DT[sens.dtype]), None, C.c_void_p(env._obs.data_ptr()),
                               C.c_void_p(env._loc.data_ptr()), None, C.byref(res), st)
    assert rc == 0
def nsl():
    lp.step(motor, sens, _DT[sens.dtype], None, env._obs, env._loc, None)
def nsl_clone():
    r = lp.step(motor, sens, _DT[sens.dtype], None, env._obs, env._loc, None)
    if r[4] is not None: r[4].clone()
def full():
    env.step(act)
fn = {"raw": raw, "nsl": nsl, "nsl_clone": nsl_clone, "full": full}[mode]
for _ in range(100): fn()
torch.cuda.synchronize(); base = rss(); series = []
for blk in range(24):
    for _ in range(500): fn()
    torch.cuda.synchronize(); series.append(round(rss() - base, 1))
print(f"{mode:10s} RSS growth MB per 500 steps:", series, flush=True)

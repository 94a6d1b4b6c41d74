import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), REPO]
import numpy as np, torch
from active_gym import ObsPipeline
dev = torch.device("cuda:0")
N, fs = int(sys.argv[1]) if len(sys.argv) > 1 else 6, 4
kw = dict(num_envs=N, kind="fixed", obs_size=(84, 84), fov_size=(30, 30), frame_stack=fs, resize_to_full=True,
          fov_init_loc=(0, 0), sensory_action_mode="absolute", device=dev)
d = ObsPipeline(**kw)
os.environ["AGX_STEP_ENV"] = "1"
v = ObsPipeline(**kw)
rng = np.random.default_rng(1)
for step in range(7):
    fr = torch.from_numpy(rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8)).to(dev)
    nvalid = rng.integers(0, 3, N); clear = (rng.random(N) < 0.2).astype(np.uint8); skip = (rng.random(N) < 0.15).astype(np.uint8)
    if len(sys.argv) > 2: clear[:] = 0; skip[:] = 0; nvalid[:] = 2
    nvalid[clear == 1] = 1
    cmd = torch.from_numpy((nvalid | clear * 4 | skip * 8).astype(np.uint8)).to(dev)
    act = torch.from_numpy(rng.uniform(-5, 60, (N, 2)).astype(np.float32)).to(dev)
    d.ingest(fr, cmd); od, ld = d.fovea(act)
    ov, lv = v.step_fixed(fr, cmd, act)
    torch.cuda.synchronize()
    sd, sv = d.stack_u8(), v.stack_u8()
    bad_s = (sd != sv).flatten(1).any(1).cpu().numpy()
    bad_o = (od != ov).flatten(2).any(2).cpu().numpy()
    print(step, "cmd", cmd.cpu().numpy().tolist(), "stack bad envs", np.nonzero(bad_s)[0].tolist(), "loc eq", torch.equal(ld, lv),
          "obs bad (env,stackpos)", np.argwhere(bad_o).tolist()[:12])
    if bad_o.any():
        e, j = np.argwhere(bad_o)[0]
        diff = (od[e, j] != ov[e, j]).cpu().numpy()
        rows = np.nonzero(diff.any(1))[0]
        print("   first bad", e, j, "rows", rows[:10], "n", diff.sum(), "max", float((od[e, j] - ov[e, j]).abs().max()))

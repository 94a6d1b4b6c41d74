import sys, numpy as np, torch
sys.path.insert(0, "active-gym_amd"); sys.path.insert(0, ".")
from active_gym import ObsPipeline
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for mode in ("mask", "raw", "resize"):
    p = ObsPipeline(num_envs=1, kind="fixed", obs_size=(84, 84), frame_stack=2, fov_size=(30, 30), fov_init_loc=(10, 20),
                    sensory_action_mode="absolute", resize_to_full=mode == "resize", mask_out=mode == "mask", device=dev)
    st = rng.integers(1, 256, (1, 2, 84, 84), dtype=np.uint8)
    p.set_stack_u8(torch.from_numpy(st).to(dev))
    p.fovea_reset()
    for act in (None, np.array([[1.0, 38.0]]), np.array([[31.0, 5.0]]), np.array([[54.0, 54.0]]), np.array([[0.0, 0.0]])):
        r = p.fovea(None if act is None else torch.from_numpy(act).to(dev))
        obs, loc = r[0].cpu().numpy()[0], r[1].cpu().numpy()[0]
        rr, cc = int(loc[0]), int(loc[1])
        want = st[0, :, rr:rr + 30, cc:cc + 30].astype(np.float32) / np.float32(255)
        if mode == "mask":
            got = obs[:, rr:rr + 30, cc:cc + 30]
        elif mode == "raw":
            got = obs
        else:
            print(mode, loc, "finite", np.isfinite(obs).all(), "corner", obs[0, 0, 0], want[0, 0, 0], obs[1, 0, 0], want[1, 0, 0], obs[0, -1, -1], want[0, -1, -1]); continue
        bad = np.argwhere(got != want)
        print(mode, loc, "bad", len(bad), "first", bad[:5].tolist(), "rows", sorted(set(bad[:, 1]))[:8], "cols", sorted(set(bad[:, 2]))[:12])
    p.close()

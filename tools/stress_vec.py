import sys, os
sys.path[:0]=[os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"active-gym_amd")]
import numpy as np, torch
from active_gym import AtariEnvArgs, AtariVecEnv
for N, chunk, fmt, kind in ((1,0,"rgb","fixed"),(1,1,"gray","flexible"),(2049,300,"rgb","peripheral"),(777,64,"gray","fixed"),(4096,512,"gray","base")):
    args = AtariEnvArgs(game="pong", seed=3, obs_size=(84,84), fov_size=(30,30), fov_init_loc=(0,0), sensory_action_mode="absolute",
                        resize_to_full=True, peripheral_res=(20,20), frame_source="native", frame_format=fmt, h2d_chunk_envs=chunk, device="cuda",
                        scripted_p_life=30, scripted_p_over=10)
    env = AtariVecEnv(args, N, kind=kind)
    obs, info = env.reset()
    tot=0
    for t in range(12):
        m = np.random.randint(0, 4, N)
        a = m if kind=="base" else {"motor_action": m, "sensory_action": np.random.randint(0,55,(N,2))}
        if kind=="flexible": a["sensory_action_type"]=np.random.randint(0,2,N)
        obs, r, d, tr, info = env.step(a); tot+=int(d.sum())
    torch.cuda.synchronize()
    assert obs.shape[0]==N and torch.isfinite(obs).all() and float(obs.max())<=1.0
    print("ok", N, chunk, fmt, kind, tuple(obs.shape), "dones", tot, flush=True)
    env.close()

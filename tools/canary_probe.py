"""tools/canary_probe.py - which state does the asm-store bug of commit 327a14a need?

lib/libagx_canary.so (python active-gym_amd/build.py --canary) is today's library with that commit's inline-asm observation
store.  Round 3's env-chain tests saw it corrupt a few hundred values per launch while every kernel-level parity test stayed
green.  This probe runs K2 (headline geometry, resize_to_full) through a matrix of launch conditions under the canary and
under the shipped library, compares every output with a torch reference computed on the device from the ring itself, and
prints where the wrong values sit (stacked frame, store pass, wave, float4 component).

    python tools/canary_probe.py [lib ...]  # parent: one child per library (default: libagx_canary.so libagx.so; libagx_r3bug.so =
                                            # the library built from a checkout of commit 327a14a itself, copied into lib/)
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), REPO]


def child():
    import numpy as np
    import torch
    import torch.nn.functional as F
    from active_gym import _native as nat
    import ctypes
    probe = ctypes.CDLL(os.environ["AGX_LIB"])
    for name in list(nat.SIGNATURES):                  # a library built from an older commit lacks the newer entry points
        if not hasattr(probe, name):
            del nat.SIGNATURES[name]
    nat.ABI_VERSION = probe.agx_abi_version()
    from active_gym import ObsPipeline
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(5)

    def reference(p, loc):
        st = p.stack_u8().float() / 255.0                                   # [N, fs, 84, 84]
        N, fs = st.shape[:2]
        out = torch.empty_like(st)
        for i in range(N):
            r, c = int(loc[i, 0]), int(loc[i, 1])
            out[i] = F.interpolate(st[i:i + 1, :, r:r + 30, c:c + 30], size=(84, 84), mode="bilinear", align_corners=False)[0]
        return out

    def where(bad):
        """bad: bool [N, fs, 84, 84] -> histogram over (store pass, wave, component) of the float4 index q = tid + 256 * pass"""
        idx = bad.nonzero().cpu().numpy()
        h = {}
        for n, j, y, x in idx:
            q = (y * 84 + x) // 4
            key = f"pass{q // 256}/wave{(q % 256) // 64}/comp{x % 4}"
            h[key] = h.get(key, 0) + 1
        frames = sorted({(int(n), int(j)) for n, j, _, _ in idx})
        return h, frames[:12]

    def run(name, N, fs, prep, action, sync_before=False, reps=1):
        kw = dict(num_envs=N, kind="fixed", obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(3.5, 4.49), frame_stack=fs,
                  resize_to_full=True, device=dev)
        tot, first = 0, None
        for rep in range(reps):
            p = ObsPipeline(**kw)
            prep(p, N, fs)
            if sync_before:
                torch.cuda.synchronize()
            a = None
            if action:
                a = (torch.rand((N, 2), device=dev, generator=g) * 65 - 5).contiguous()
            obs, loc = p.fovea(a)
            torch.cuda.synchronize()
            ref = reference(p, loc.cpu().numpy())
            bad = (obs - ref).abs() > 1e-4
            nbad = int(bad.sum())
            tot += nbad
            if nbad and first is None:
                first = where(bad)
            p.close()
        print(json.dumps({"case": name, "N": N, "fs": fs, "reps": reps, "bad_values": tot,
                          "where": first[0] if first else {}, "frames": first[1] if first else []}), flush=True)

    def prep_set_random(p, N, fs):
        p.set_stack_u8(torch.randint(0, 256, (N, fs, 84, 84), dtype=torch.uint8, device=dev, generator=g))

    def prep_set_reset_like(p, N, fs):
        st = torch.zeros((N, fs, 84, 84), dtype=torch.uint8, device=dev)
        st[:, -1] = torch.randint(0, 256, (N, 84, 84), dtype=torch.uint8, device=dev, generator=g)
        p.set_stack_u8(st)
        p.fovea_reset()

    def prep_ingest_clear(p, N, fs):                      # AtariVecEnv.reset(): K1 with CLEAR | nvalid 1, then fovea_reset
        fr = torch.randint(0, 256, (N, 2, 210, 160, 3), dtype=torch.uint8, device=dev, generator=g)
        p.ingest(fr, torch.full((N,), 1 | 4, dtype=torch.uint8, device=dev))
        p.fovea_reset()

    def prep_ingest_steps(p, N, fs):                      # a live ring: fs + 1 plain steps
        for _ in range(fs + 1):
            fr = torch.randint(0, 256, (N, 2, 210, 160, 3), dtype=torch.uint8, device=dev, generator=g)
            p.ingest(fr, torch.full((N,), 2, dtype=torch.uint8, device=dev))

    for N in (1, 5, 33, 256, 1024):
        reps = 3 if N <= 33 else 1
        run("set_stack(random) -> fovea(action)", N, 4, prep_set_random, True, reps=reps)
        run("set_stack(random) -> sync -> fovea(action)", N, 4, prep_set_random, True, sync_before=True, reps=reps)
        run("set_stack(random) -> fovea(None)", N, 4, prep_set_random, False, reps=reps)
        run("set_stack([0,0,0,frame]) -> fovea_reset -> fovea(None)", N, 4, prep_set_reset_like, False, reps=reps)
        run("ingest(CLEAR) -> fovea_reset -> fovea(None)   [AtariVecEnv.reset]", N, 4, prep_ingest_clear, False, reps=reps)
        run("ingest(CLEAR) -> fovea_reset -> sync -> fovea(None)", N, 4, prep_ingest_clear, False, sync_before=True, reps=reps)
        run("ingest x5 -> fovea(action)   [AtariVecEnv.step]", N, 4, prep_ingest_steps, True, reps=reps)
        run("ingest x5 -> sync -> fovea(action)", N, 4, prep_ingest_steps, True, sync_before=True, reps=reps)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    libdir = os.path.join(REPO, "active-gym_amd", "lib")
    names = sys.argv[1:] or ["libagx_canary.so", "libagx.so"]
    for name in names:
        if not os.path.exists(os.path.join(libdir, name)):
            print(f"== {name}: not built", flush=True)
            continue
        print(f"== {name}", flush=True)
        env = dict(os.environ, AGX_LIB=os.path.join(libdir, name))
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)

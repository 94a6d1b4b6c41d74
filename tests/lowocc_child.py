"""Child of tests/test_gpu_lowocc.py: the fovea kernels of every kind at LOW OCCUPANCY, in the launch sequences the env chain
produces, against the CPU oracle.  Runs under whatever library AGX_LIB names and prints one JSON line per case
({"case", "bad"}: observation values off by more than 1e-5, plus fov_loc / fov_res mismatches); exits 0 whatever it finds -
the parent decides what the counts must be for that library."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), REPO, os.path.join(REPO, "tests")]


def main():
    import torch
    from active_gym import ObsPipeline
    from golden_util import unit64
    from oracle import oracle as O
    dev = torch.device("cuda:0")
    TOL = 1e-5
    kinds = [
        ("fixed-resize", "fixed", O.FixedFovealOracle, dict(resize_to_full=True)),
        ("fixed-mask", "fixed", O.FixedFovealOracle, dict(resize_to_full=False, mask_out=True)),
        ("fixed-raw", "fixed", O.FixedFovealOracle, dict(resize_to_full=False)),
        ("peripheral", "peripheral", O.PeripheralOracle, dict(peripheral_res=(20, 20), resize_to_full=True)),
        ("flexible-resize", "flexible", O.FlexibleFovealOracle, dict(resize_to_full=True)),
        ("flexible-mask", "flexible", O.FlexibleFovealOracle, dict(resize_to_full=False, mask_out=True)),
        ("flexible-raw", "flexible", O.FlexibleFovealOracle, dict(resize_to_full=False)),
        ("flexible-packed", "flexible", O.FlexibleFovealOracle, dict(resize_to_full=False)),
    ]
    only = os.environ.get("LOWOCC_ONLY")
    for name, kind, orc, extra in kinds:
        if only and only not in name:
            continue
        for N in (1, 5):
            rng = np.random.default_rng(1000 + N)
            fs = 4
            kw = dict(obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(3.5, 4.49), sensory_action_mode="absolute")
            kw.update(extra)
            okw = dict(kw)
            if kind != "peripheral":
                okw.pop("peripheral_res", None)
            p = ObsPipeline(num_envs=N, kind=kind, frame_stack=fs, device=dev, **kw)
            orcs = [orc(**okw) for _ in range(N)]
            ring = O.RingOracle(N, fs, (84, 84))
            packed = name == "flexible-packed"
            bad = 0

            def observe(action=None, types=None):
                if packed:
                    flat, off, loc, res = p.fovea_packed(action, action_type=types)
                    flat, off, loc, res = flat.cpu().numpy(), off.cpu().numpy(), loc.cpu().numpy(), res.cpu().numpy()
                    obs = [flat[int(off[i]):int(off[i + 1])].reshape(fs, int(res[i, 0]), int(res[i, 1])) for i in range(N)]
                    return obs, loc, res
                if kind == "flexible":
                    obs, loc, res = p.fovea(action, action_type=types)
                    return obs.cpu().numpy(), loc.cpu().numpy(), res.cpu().numpy()
                obs, loc = p.fovea(action)
                return obs.cpu().numpy(), loc.cpu().numpy(), None

            def check(obs, loc, res, wants):
                n_bad = 0
                for i in range(N):
                    n_bad += int(not np.array_equal(loc[i], orcs[i].fov_loc))
                    got = obs[i]
                    if kind == "flexible":
                        n_bad += int(not np.array_equal(res[i], orcs[i].fov_res))
                        if name == "flexible-raw":
                            rh, rw = (int(v) for v in orcs[i].fov_res)
                            n_bad += int(np.count_nonzero(got[:, rh:, :]) + np.count_nonzero(got[:, :rh, rw:]))
                            got = got[:, :rh, :rw]
                    if got.shape != wants[i].shape:
                        n_bad += got.size
                        continue
                    n_bad += int((np.abs(got.astype(np.float64) - wants[i]) > TOL).sum())
                return n_bad

            # (1) AtariVecEnv.reset(): K1 with CLEAR | one screen, fovea_reset, fovea(None) - the ring is [0, .., 0, frame]
            frames = rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8)
            cmd = np.full(N, 1 | 4, np.uint8)
            p.ingest(torch.from_numpy(frames).to(dev), torch.from_numpy(cmd).to(dev))
            p.fovea_reset()
            obs, loc, res = observe()
            ring.ingest(frames, cmd & 3, clear=np.ones(N, np.uint8))
            full = ring.full_state()
            bad_reset = check(obs, loc, res, [orcs[i].reset(full[i]) for i in range(N)])
            print(json.dumps({"case": f"{name} N={N} reset chain (ingest CLEAR -> fovea_reset -> fovea(None))", "bad": bad_reset}), flush=True)
            # (2) AtariVecEnv.step() x 5: K1, then the fovea kernel with an action, a live ring from the fourth step on
            for step in range(5):
                frames = rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8)
                cmd = np.full(N, 2, np.uint8)
                types = None
                if kind == "flexible":
                    types = rng.integers(0, 2, N).astype(np.int32)
                    a = np.where(types[:, None] == 1, rng.integers(10, 61, (N, 2)), rng.integers(-5, 80, (N, 2))).astype(np.float64)
                else:
                    a = rng.uniform(-5, 60, (N, 2))
                p.ingest(torch.from_numpy(frames).to(dev), torch.from_numpy(cmd).to(dev))
                obs, loc, res = observe(torch.from_numpy(a).to(dev), None if types is None else torch.from_numpy(types).to(dev))
                ring.ingest(frames, cmd)
                full = ring.full_state()
                if kind == "flexible":
                    wants = [orcs[i].step(full[i], a[i].astype(np.int64), np.array((types[i],))) for i in range(N)]
                else:
                    wants = [orcs[i].step(full[i], a[i]) for i in range(N)]
                bad += check(obs, loc, res, wants)
            print(json.dumps({"case": f"{name} N={N} 5 steps (ingest -> fovea(action))", "bad": bad}), flush=True)
            p.close()


if __name__ == "__main__":
    main()

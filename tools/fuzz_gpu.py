"""Randomised-geometry parity fuzz on the GPU: random obs / fov / peripheral sizes, frame stacks, modes, antialias and
actions (incl. out-of-range and .5 ties) through libagx against the oracle, for a time budget.  Exits non-zero on the
first mismatch and prints the configuration that produced it.

    python tools/fuzz_gpu.py [seconds] [seed]
"""
import os, sys, time, traceback
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "active-gym_amd"), REPO]
import numpy as np
import torch
from active_gym import ObsPipeline
from active_gym._native import AgxError
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
TOL = 1e-5


def unit64(u8):
    return (u8.astype(np.float32) / np.float32(255)).astype(np.float64)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def one_case(k):
    kind = ["fixed", "flexible", "peripheral", "ingest", "ingest_gray", "ingest_rgb"][k % 6]
    oh = int(rng.integers(3, 33)) * 4
    ow = oh if kind in ("ingest", "ingest_gray") or rng.random() < 0.5 else int(rng.integers(3, 33)) * 4
    fs = int(rng.integers(1, 6))
    N = int(rng.integers(1, 9))
    cfg = dict(kind=kind, obs=(oh, ow), fs=fs, N=N)
    if kind in ("ingest_gray", "ingest_rgb"):
        from active_gym import _native as nat
        p = ObsPipeline(num_envs=N, kind="base", obs_size=(oh, ow), frame_stack=fs, device=dev)
        ring = np.zeros((N, fs, oh, ow), np.uint8)
        mode = "cv15" if rng.random() < 0.5 else "cv14"
        for step in range(3):
            nvalid = rng.integers(0, 3, N)
            clear = (rng.random(N) < 0.3).astype(np.uint8)
            skip = (rng.random(N) < 0.2).astype(np.uint8)
            nvalid[clear == 1] = 1
            cmd = (nvalid | clear * 4 | skip * 8).astype(np.uint8)
            if kind == "ingest_gray":
                src = rng.integers(0, 256, (N, 2, 210, 160), dtype=np.uint8)
                p.ingest_gray_raw(t(src), t(cmd))
            else:
                src = rng.integers(0, 256, (N, oh, ow, 3), dtype=np.uint8)
                p.ingest_rgb(t(src), t(cmd), nat.GRAY_CV15 if mode == "cv15" else nat.GRAY_CV14)
            for i in range(N):
                if skip[i]:
                    continue
                if clear[i]:
                    ring[i] = 0
                new = np.zeros((oh, ow), np.uint8)
                if kind == "ingest_gray":
                    for f in range(int(nvalid[i])):
                        new = np.maximum(new, O.cv_resize_linear_u8(src[i, f], (ow, oh)))
                elif nvalid[i]:
                    new = O.cv_bgr2gray_u8(src[i], mode)
                ring[i] = np.concatenate([ring[i, 1:], new[None]], 0)
            assert np.array_equal(p.stack_u8().cpu().numpy(), ring), cfg
        p.close()
        return cfg
    if kind == "ingest":
        p = ObsPipeline(num_envs=N, kind="base", obs_size=(oh, ow), frame_stack=fs, device=dev)
        ring = O.RingOracle(N, fs, (oh, ow))
        for step in range(3):
            frames = rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8)
            nvalid = rng.integers(0, 3, N)
            clear = (rng.random(N) < 0.3).astype(np.uint8)
            skip = (rng.random(N) < 0.2).astype(np.uint8)
            nvalid[clear == 1] = 1
            p.ingest(t(frames), t((nvalid | clear * 4 | skip * 8).astype(np.uint8)))
            ring.ingest(frames, nvalid, clear=clear, skip=skip)
            assert np.array_equal(p.stack_u8().cpu().numpy(), ring.stack_u8()), cfg
        assert np.array_equal(p.observe_full().cpu().numpy(), ring.full_state().astype(np.float32)), cfg
        p.close()
        return cfg
    fh, fw = int(rng.integers(1, oh)), int(rng.integers(1, ow))
    mode = "absolute" if rng.random() < 0.6 else "relative"
    out = ["resize", "mask", "raw"][int(rng.integers(0, 3))]
    aa = bool(rng.integers(0, 2))
    init = (float(rng.uniform(0, oh - fh)), float(rng.uniform(0, ow - fw)))
    kw = dict(obs_size=(oh, ow), fov_size=(fh, fw), fov_init_loc=init, sensory_action_mode=mode,
              sensory_action_space=(-7.0, 9.0), antialias=aa)
    cfg.update(fov=(fh, fw), mode=mode, out=out, aa=aa, init=init)
    if kind == "peripheral":
        per = (int(rng.integers(1, oh + 1)), int(rng.integers(1, ow + 1)))
        cfg["per"] = per
        okw = dict(peripheral_res=per, **kw)
        pkw = dict(peripheral_res=per, resize_to_full=True, **kw)
        orc = lambda: O.PeripheralOracle(**okw)
    else:
        pkw = dict(resize_to_full=(out == "resize"), mask_out=(out == "mask"), **kw)
        orc = (lambda: O.FixedFovealOracle(**pkw)) if kind == "fixed" else (lambda: O.FlexibleFovealOracle(**pkw))
    try:
        p = ObsPipeline(num_envs=N, kind=kind, frame_stack=fs, device=dev, **pkw)
    except (AgxError, ValueError) as e:                      # geometry rejected by agx_create (LDS budget etc.)
        cfg["rejected"] = str(e)[:80]
        return cfg
    orcs = [orc() for _ in range(N)]
    for o in orcs:
        o.reset(np.zeros((fs, oh, ow)))
    # flexible raw crops: every other case goes through the packed ragged entry point (agx_fovea_flexible_packed)
    packed_case = kind == "flexible" and out == "raw" and rng.random() < 0.5
    cfg["packed"] = packed_case
    for step in range(4):
        st = rng.integers(0, 256, (N, fs, oh, ow), dtype=np.uint8)
        p.set_stack_u8(t(st))
        types = rng.integers(0, 2, N).astype(np.int32)
        if kind == "flexible":
            a = np.where(types[:, None] == 1, np.stack([rng.integers(1, oh + 1, N), rng.integers(1, ow + 1, N)], 1),
                         np.stack([rng.integers(-5, oh + 5, N), rng.integers(-5, ow + 5, N)], 1)).astype(np.int64)
            if packed_case:
                if oh == ow and rng.random() < 0.5:
                    # the whole step as one call (agx_step_flexible_packed: the state / scan blocks ride in the ingest launch where the
                    # geometry has the band12 plan, three launches inside the call otherwise) with every env's command = SKIP, so
                    # that the ring set above stands and the crops must be what agx_fovea_flexible_packed gives on it
                    cfg["one_call"] = True
                    gray = rng.random() < 0.5
                    scr = rng.integers(0, 256, (N, 2, 210, 160) + (() if gray else (3,)), dtype=np.uint8)
                    flat, off, loc_t, res_t = p.step_flexible_packed(t(scr), t(np.full(N, 8, np.uint8)), t(a), action_type=t(types))
                else:
                    flat, off, loc_t, res_t = p.fovea_packed(t(a), action_type=t(types))
                flat, off, res_n = flat.cpu().numpy(), off.cpu().numpy(), res_t.cpu().numpy()
                assert off[0] == 0 and np.array_equal(np.diff(off), fs * res_n[:, 0].astype(np.int64) * res_n[:, 1]), (cfg, step)
                pad = np.zeros((N, fs, oh, ow), np.float32)
                for i in range(N):
                    pad[i, :, :res_n[i, 0], :res_n[i, 1]] = flat[off[i]:off[i + 1]].reshape(fs, res_n[i, 0], res_n[i, 1])
                r = (torch.from_numpy(pad), loc_t, res_t)
            else:
                r = p.fovea(t(a), action_type=t(types))
        else:
            lo, hi = (-6.0, max(oh, ow) + 6.0) if mode == "absolute" else (-12.0, 12.0)
            a = rng.uniform(lo, hi, (N, 2))
            a[::3] = np.floor(a[::3]) + 0.5
            a = a.astype(np.float32 if rng.random() < 0.5 else np.float64)
            r = p.fovea(t(a))
        obs, loc = r[0].cpu().numpy(), r[1].cpu().numpy()
        for i in range(N):
            if kind == "flexible":
                want = orcs[i].step(unit64(st[i]), a[i], np.array((types[i],)))
                assert np.array_equal(r[2].cpu().numpy()[i], orcs[i].fov_res), (cfg, step, i)
            else:
                want = orcs[i].step(unit64(st[i]), a[i])
            assert np.array_equal(loc[i], orcs[i].fov_loc), (cfg, step, i, a[i], loc[i], orcs[i].fov_loc)
            got = obs[i]
            if kind == "flexible" and out == "raw":
                rh, rw = orcs[i].fov_res
                assert not got[:, rh:, :].any() and not got[:, :, rw:].any(), (cfg, step, i)
                got = got[:, :rh, :rw]
            assert got.shape == want.shape, (cfg, got.shape, want.shape)
            err = float(np.abs(got - want).max()) if got.size else 0.0
            assert err <= TOL, (cfg, step, i, err)
    p.close()
    return cfg


t0 = time.time()
n = rejected = 0
kinds = {}
reasons = {}
while time.time() - t0 < budget:
    try:
        c = one_case(n)
    except Exception:                                        # noqa: BLE001 - report and stop
        traceback.print_exc()
        print("FUZZ FAILURE after", n, "cases, seed", seed, flush=True)
        sys.exit(1)
    n += 1
    rejected += "rejected" in c
    if "rejected" in c:
        key = "".join(ch for ch in c["rejected"].split("(")[0] if not ch.isdigit())[:60]
        reasons[key] = reasons.get(key, 0) + 1
    kinds[c["kind"]] = kinds.get(c["kind"], 0) + 1
    if n % 2000 == 0:
        print(f"{n} cases ok ({rejected} rejected by agx_create) {kinds} {time.time() - t0:.0f}s", flush=True)
print(f"fuzz ok: {n} cases, {rejected} rejected by agx_create, {kinds}, seed {seed}")
print("rejections:", reasons)

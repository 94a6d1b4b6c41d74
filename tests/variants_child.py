"""Child of tests/test_gpu_variants.py: runs inside a process whose AGX_LIB points at libagx_exp.so (the experiments
build).  Every experimental kernel form, selected by its environment knob (read per context in agx_create), is held
against the default kernels of the same library bit for bit; the CRCs of the default kernels' outputs are printed so that
the parent can compare them with what libagx.so itself produces."""
import json
import os
import sys
import zlib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "active-gym_amd"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch

KNOBS = ("AGX_PACKED_WAVE", "AGX_INGEST_NO_FULL", "AGX_INGEST_T", "AGX_INGEST_BAND_ROWS", "AGX_INGEST_PIPE", "AGX_INGEST_WAVE", "AGX_FOVEA_PAIR",
         "AGX_STEP_FUSED", "AGX_STEP_SPLIT", "AGX_STEP_AUX_PRIO", "AGX_INGEST_PAIR12", "AGX_STEP_ENV")
VARIANTS = [{"AGX_INGEST_T": "128"}, {"AGX_INGEST_BAND_ROWS": "7"}, {"AGX_INGEST_BAND_ROWS": "11"}, {"AGX_INGEST_PIPE": "2"},
            {"AGX_INGEST_PIPE": "7"}, {"AGX_INGEST_WAVE": "1"}, {"AGX_INGEST_NO_FULL": "1"}, {"AGX_INGEST_PAIR12": "1"},
            {"AGX_FOVEA_PAIR": "1"}, {"AGX_STEP_FUSED": "1"}, {"AGX_STEP_FUSED": "2"}, {"AGX_STEP_FUSED": "3"}, {"AGX_STEP_SPLIT": "2"}, {"AGX_STEP_SPLIT": "3"}, {"AGX_STEP_ENV": "1"},
            {"AGX_STEP_SPLIT": "4", "AGX_STEP_AUX_PRIO": "-1"}, {"AGX_INGEST_T": "128", "AGX_FOVEA_PAIR": "1"}]


def tie_pixels():
    out = []
    for r in range(0, 256, 5):
        for g in range(256):
            for b in range(0, 256, 2):
                if (2989 * r + 5870 * g + 1140 * b) % 10000 == 5000:
                    out.append((r, g, b))
    return np.array(out, dtype=np.uint8)


def inputs(seed, N, steps, ties):
    rng = np.random.default_rng(seed)
    for step in range(steps):
        fr = rng.integers(0, 256, (N, 2, 210, 160, 3), dtype=np.uint8)
        fr[step % N].reshape(-1, 3)[: len(ties)] = ties                       # exact .5 luminance ties in one env
        nvalid = rng.integers(0, 3, N)
        clear = (rng.random(N) < 0.2).astype(np.uint8)
        skip = (rng.random(N) < 0.15).astype(np.uint8)
        nvalid[clear == 1] = 1
        yield fr, (nvalid | clear * 4 | skip * 8).astype(np.uint8), rng.uniform(-5, 60, (N, 2)).astype(np.float32)


def default_crcs(dev, geom="headline"):
    """CRC32 of (u8 stack, fov_loc, observations) after each of 7 steps of the DEFAULT kernels on seeded inputs."""
    from active_gym import ObsPipeline
    N, fs = 45, 4
    fov = (30, 30) if geom == "headline" else (26, 34)
    d = ObsPipeline(num_envs=N, kind="fixed", obs_size=(84, 84), fov_size=fov, frame_stack=fs, resize_to_full=True,
                    fov_init_loc=(0, 0), sensory_action_mode="absolute", device=dev)
    out = []
    for fr, cmd, act in inputs(11, N, 7, tie_pixels()):
        d.ingest(torch.from_numpy(fr).to(dev), torch.from_numpy(cmd).to(dev))
        o, l = d.fovea(torch.from_numpy(act).to(dev))
        out.append([zlib.crc32(d.stack_u8().cpu().numpy().tobytes()), zlib.crc32(l.cpu().numpy().tobytes()),
                    zlib.crc32(o.cpu().numpy().tobytes())])
    d.close()
    return out


def main():
    from active_gym import ObsPipeline, _native as nat
    assert "libagx_exp.so" in os.environ.get("AGX_LIB", ""), "run with AGX_LIB=<...>/libagx_exp.so"
    dev = torch.device("cuda:0")
    ties = tie_pixels()
    report = {"build": nat.build_info(), "variants": {}}
    for geom in ("headline", "generic"):
        fov = (30, 30) if geom == "headline" else (26, 34)
        N, fs = 45, 4
        kw = dict(num_envs=N, kind="fixed", obs_size=(84, 84), fov_size=fov, frame_stack=fs, resize_to_full=True,
                  fov_init_loc=(0, 0), sensory_action_mode="absolute", device=dev)
        for knob in VARIANTS:
            for k in KNOBS:
                os.environ.pop(k, None)
            d = ObsPipeline(**kw)
            os.environ.update(knob)
            v = ObsPipeline(**kw)
            name = geom + ":" + ",".join(f"{a}={b}" for a, b in knob.items())
            ok = True
            for step, (fr, cmd, act) in enumerate(inputs(11, N, 7, ties)):
                frames, cmd_t, act_t = torch.from_numpy(fr).to(dev), torch.from_numpy(cmd).to(dev), torch.from_numpy(act).to(dev)
                d.ingest(frames, cmd_t)
                od, ld = d.fovea(act_t)
                if "AGX_STEP_FUSED" in knob or "AGX_STEP_SPLIT" in knob or "AGX_STEP_ENV" in knob:
                    ov, lv = v.step_fixed(frames, cmd_t, act_t)
                else:
                    v.ingest(frames, cmd_t)
                    ov, lv = v.fovea(act_t)
                if not (torch.equal(d.stack_u8(), v.stack_u8()) and torch.equal(ld, lv) and torch.equal(od, ov)):
                    ok = False
                    report["variants"][name] = f"differs from the default kernels at step {step}"
                    break
            if ok:
                report["variants"][name] = "ok"
            d.close()
            v.close()
    # the packed ragged crops with one wave per (slot, env) item (round 4, experiments/agx_packed_wave.h) against the shipped form
    for geom in ("headline", "generic"):
        fov = (30, 30) if geom == "headline" else (26, 34)
        N, fs = 45, 3
        kw = dict(num_envs=N, kind="flexible", obs_size=(84, 84), fov_size=fov, frame_stack=fs, resize_to_full=False, mask_out=False,
                  fov_init_loc=(0, 0), sensory_action_mode="absolute", device=dev)
        for aa in (True, False):
            for k in KNOBS:
                os.environ.pop(k, None)
            d = ObsPipeline(antialias=aa, **kw)
            os.environ["AGX_PACKED_WAVE"] = "1"
            v = ObsPipeline(antialias=aa, **kw)
            name = f"{geom}:flexible packed aa={int(aa)}:AGX_PACKED_WAVE=1"
            report["variants"][name] = "ok"
            rng = np.random.default_rng(5 + aa)
            for step in range(6):
                st = torch.from_numpy(rng.integers(0, 256, (N, fs, 84, 84), dtype=np.uint8)).to(dev)
                d.set_stack_u8(st)
                v.set_stack_u8(st)
                types = rng.integers(0, 2, N).astype(np.int32)
                a = np.where(types[:, None] == 1, rng.integers(1, 85, (N, 2)), rng.integers(-5, 80, (N, 2))).astype(np.int64)
                at, tt = torch.from_numpy(a).to(dev), torch.from_numpy(types).to(dev)
                pd, od, ld, rd = d.fovea_packed(at, action_type=tt)
                pv, ov, lv, rv = v.fovea_packed(at, action_type=tt)
                tot = int(od[-1])
                if not (torch.equal(od, ov) and torch.equal(ld, lv) and torch.equal(rd, rv) and torch.equal(pd[:tot], pv[:tot])):
                    report["variants"][name] = f"differs from the shipped packed kernel at step {step}"
                    break
            d.close()
            v.close()
    for k in KNOBS:
        os.environ.pop(k, None)
    report["default_crcs"] = {g: default_crcs(dev, g) for g in ("headline", "generic")}
    print("VARIANTS_JSON " + json.dumps(report))


if __name__ == "__main__":
    main()

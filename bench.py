#!/usr/bin/env python3
"""bench.py — env steps/sec of the batched active-vision observation path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM: K1 ingest (2 raw RGB frames per env -> gray -> OpenCV
fixed-point resize -> 2-frame max -> frame-stack ring) followed by the fovea
kernel (sensory action -> crop -> bilinear resize to 84x84, f32 out).
Workload = BASELINE.json configs[1]: 1024x AtariFixedFovealEnv per GPU, 84x84 obs,
30x30 fovea, frame_stack=4, action_repeat=4 (=> two sampled frames per step).

For N>1 the driver launches one rank per GPU via torch.distributed.run; envs shard
by contiguous index blocks, there is no data-path collective (weak scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "active-gym_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=1024, help="envs per GPU")
    ap.add_argument("--kind", default="fixed", choices=["fixed", "peripheral", "flexible"])
    ap.add_argument("--pool", type=int, default=8, help="distinct synthetic input batches cycled through")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive AtariVecEnv leg (the `e2e` key)")
    ap.add_argument("--frame-format", default="rgb", choices=("rgb", "gray"),
                    help="rgb = the metric's workload (raw RGB screens, luminance on the device); gray = ALE grayscale "
                         "screens in (agx_ingest_gray_raw) - a different, lighter workload, reported for DESIGN.md only")
    ap.add_argument("--no-events", action="store_true", help="skip the per-kernel HIP-event sampling pass (no `roofline`)")
    ap.add_argument("--antialias", type=int, default=1, choices=(0, 1),
                    help="peripheral / flexible kinds: torchvision Resize's antialias setting (SURVEY.md 8d: both are sub-runs)")
    ap.add_argument("--out", default="full", choices=("full", "packed"),
                    help="flexible kind: full = resize_to_full (the config-4 workload); packed = the ragged raw-crop sub-run "
                         "(agx_fovea_flexible_packed: [fs, rh, rw] crops packed back to back + offsets)")
    ap.add_argument("--packed-calls", type=int, default=1, choices=(1, 2),
                    help="--out packed: 1 = the whole step as agx_step_flexible_packed (two launches: the state update + scan ride in "
                         "the ingest launch); 2 = agx_ingest + agx_fovea_flexible_packed (three launches), for A/B runs")
    ap.add_argument("--preroll", type=int, default=1000,
                    help="untimed steps run BEFORE the warmup so that a short run (the driver's --steps 20) is not timed on the "
                         "first milliseconds of an idle device (clock ramp); not part of `warmup` or `steps`")
    ap.add_argument("--rehearsals", type=int, default=4,
                    help="untimed dress rehearsals of the timed region (synchronize, `steps` steps, synchronize) between the pre-roll and "
                         "the warmup: the first synchronize -> launch -> synchronize cycle of a process runs 1-2.5 us per step slower "
                         "than the following ones (profiles/r04_region_sequence.txt)")
    ap.add_argument("--samples", type=int, default=24,
                    help="launch pairs carrying their own HIP events in the sampling pass that FOLLOWS the timed region")
    ap.add_argument("--obs-pool", type=int, default=3,
                    help="second sampling pass of the fovea kernel with its output rotating through this many observation "
                         "buffers (3 x 115.6 MB no longer fits the 256 MB Infinity Cache: `kernels_obs_pool` is the kernel "
                         "against HBM, `kernels` the product's double-buffered form); 0 = skip")
    ap.add_argument("--host-wait", default="spin", choices=("spin", "yield", "auto"),
                    help="how the host thread waits in synchronize() at the end of the timed region (hipSetDeviceFlags)")
    ap.add_argument("--compact", action="store_true",
                    help="time the step on COMPACT input screens (only the 168 of 210 rows K1 reads, agx_ingest_compact: what the "
                         "host runner stages) instead of whole screens; the default run reports it beside `value` as `compact_input`")
    ap.add_argument("--event-mode", default="kernel", choices=("kernel", "stream"),
                    help="kernel: start/stop HIP events stamped by the launch itself (hipExtLaunchKernelGGL through "
                         "agx_profile_next: the kernel's own begin/end, what rocprofv3 reports); stream: events recorded "
                         "on the stream before/after each launch (adds the ~2-3 us dispatch gap and slows the stream)")
    ap.add_argument("--event-every", type=int, default=4,
                    help="sampling pass: every M-th step carries the events, the steps between run plain so that a sampled "
                         "launch sees the same neighbours as in the timed region")
    return ap.parse_args()


def synth_inputs(torch, dev, n, pool, seed, gray=False):
    """Synthetic Atari-shaped inputs: i.i.d. uniform u8 RGB frames, 0.1% of env-steps
    with fewer than two sampled frames, 0.1% full resets, absolute actions uniform in [-5, 60)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    frames, cmds, acts = [], [], []
    for _ in range(pool):
        frames.append(torch.randint(0, 256, (n, 2, 210, 160) + (() if gray else (3,)), dtype=torch.uint8, device=dev, generator=g))
        u = torch.rand((n,), device=dev, generator=g)
        nvalid = torch.full((n,), 2, dtype=torch.uint8, device=dev)
        nvalid[u < 0.001] = 1
        nvalid[u < 0.0005] = 0
        reset = torch.rand((n,), device=dev, generator=g) < 0.001
        cmd = torch.where(reset, torch.tensor(1 | 0x04, dtype=torch.uint8, device=dev), nvalid)
        cmds.append(cmd.contiguous())
        acts.append((torch.rand((n, 2), device=dev, generator=g) * 65.0 - 5.0).contiguous())
    return frames, cmds, acts


def workload_name(args, n, packed_mode, gray, compact):
    """config.workload: which BASELINE.json config this run is (configs[1] = the metric's; [2] / [3] = the peripheral / flexible
    sub-runs of SURVEY.md 8d)."""
    which = {"fixed": ("AtariFixedFovealEnv", "configs[1]"), "peripheral": ("AtariFixedFovealPeripheralEnv, peripheral_res 20x20", "configs[2]"),
             "flexible": ("AtariFlexibleFovealEnv, per-env fov_res in [10,60]^2, 50% FOV_RES actions", "configs[3]")}[args.kind]
    s = f"{n}x {which[0]} per GPU, 84x84 obs, 30x30 fov, frame_stack=4, action_repeat=4, "
    s += (("ragged raw crops packed (agx_step_flexible_packed: ingest + state / scan in one launch, crops in the second)" if args.packed_calls == 1 else
           "ragged raw crops packed (agx_ingest + agx_fovea_flexible_packed: three launches)") if packed_mode else "resize_to_full")
    if args.kind != "fixed":
        s += f", antialias={args.antialias}"
    s += ", absolute sensory actions; "
    if gray:
        s += "device-resident synthetic GRAY screens (getScreenGrayscale format) - NOT the metric's workload"
    else:
        s += f"device-resident synthetic RGB frames (BASELINE.json {which[1]}" + (", the ragged-raw sub-run)" if packed_mode else ")")
    if compact:
        s += "; compact input screens (--compact)"
    return s


def make_pipeline(kind, n, dev, antialias=True, packed=False):
    from active_gym import ObsPipeline
    kw = dict(num_envs=n, obs_size=(84, 84), frame_stack=4, fov_size=(30, 30), fov_init_loc=(0, 0),
              sensory_action_mode="absolute", device=dev)
    if kind == "fixed":
        return ObsPipeline(kind="fixed", resize_to_full=True, **kw)
    if kind == "peripheral":
        return ObsPipeline(kind="peripheral", peripheral_res=(20, 20), antialias=antialias, **kw)
    if packed:          # ragged raw crops (fov_env.py:283-298): neither mask_out nor resize_to_full
        return ObsPipeline(kind="flexible", resize_to_full=False, mask_out=False, antialias=antialias, **kw)
    return ObsPipeline(kind="flexible", resize_to_full=True, antialias=antialias, **kw)


_CPU_CHILD = r"""
import json, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from oracle import cport
seed, n, budget = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
rng = np.random.default_rng(seed)
frames = rng.integers(0, 256, (n, 2, 210, 160, 3), dtype=np.uint8)
acts = rng.uniform(-5, 60, (n, 2))
eb = cport.EnvBatch(n, frame_stack=4)
eb.step_fixed(frames, acts)
t0 = time.perf_counter(); reps = 0
while time.perf_counter() - t0 < budget:
    eb.step_fixed(frames, acts); reps += 1
print(json.dumps({"steps": n * reps, "wall": time.perf_counter() - t0}))
"""


def _build_info():
    from active_gym import _native as nat
    return nat.build_info()


def host_cores():
    """Host cores of this box: present, and usable by this job (scheduler affinity and the cgroup CPU quota)."""
    present = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = present
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    usable = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return {"present": present, "affinity": aff, "cgroup_quota": quota, "usable": usable}


def cpu_reference_shaped(budget_s, seed):
    """The reference's own shape of the work: one env after the other like
    SyncVectorEnv (atari_env.py:241), each step in NumPy float64 + torch's CPU `interpolate` on one thread:
    gray + cv2-style resize of the two sampled screens (oracle restatement, atari_env.py:73-75), max, deque append,
    np.stack (atari_env.py:121-143), clip/rint, crop, Resize(obs_size) (fov_env.py:166-183)."""
    import collections
    import numpy as np
    import torch
    import torch.nn.functional as F
    from oracle import oracle as O
    n = 8
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, (n, 2, 210, 160, 3), dtype=np.uint8)
    acts = rng.uniform(-5, 60, (n, 2))
    dq = [collections.deque([np.zeros((84, 84)) for _ in range(4)], maxlen=4) for _ in range(n)]
    nthr = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < budget_s:
            for i in range(n):
                fb = np.zeros((2, 84, 84))
                for f in range(2):
                    fb[f] = O.get_state_u8(frames[i, f], (84, 84)).astype(np.float32) / 255.
                dq[i].append(fb.max(0))
                state = np.stack(dq[i], 0)
                loc = np.rint(np.clip(acts[i], 0, 54)).astype(int)
                fov = state[..., loc[0]:loc[0] + 30, loc[1]:loc[1] + 30]
                F.interpolate(torch.from_numpy(fov)[None], size=(84, 84), mode="bilinear", align_corners=False, antialias=True)[0].numpy()
            reps += 1
        el = time.perf_counter() - t0
    finally:
        torch.set_num_threads(nthr)
    return {"value": n * reps / el, "unit": "env steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} envs x {reps} steps of the same config, one env after the other (SyncVectorEnv shape), NumPy float64 + "
                      f"oracle gray/cv-resize restatement + torch CPU interpolate on 1 thread, {el:.1f} s"}


def cpu_baseline(budget_s, seed):
    """The oracle (a per-env CPU port of the reference's NumPy/OpenCV/torchvision arithmetic: oracle/cport.c,
    falling back to oracle/oracle.py) timed on this box's host cores over a bounded sample of the same workload:
    one process per core usable by this job, each stepping its own 16 envs serially (one SyncVectorEnv per core); the
    children are plain subprocesses that import neither torch nor HIP, each under a hard timeout.  Beside it,
    `reference_shaped`: the reference's own serial NumPy/torch shape of the work on one core."""
    import numpy as np
    hc = host_cores()
    try:
        from oracle import cport
        have_c = cport.available()
    except Exception:
        have_c = False
    n = 16
    out = None
    if have_c:
        cores = max(1, min(hc["usable"], 512))
        cmd = lambda i: [sys.executable, "-c", _CPU_CHILD, REPO, str(seed + i), str(n), str(budget_s)]
        procs = [subprocess.Popen(cmd(i), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(cores)]
        res = []
        for pr in procs:
            try:
                o, _ = pr.communicate(timeout=budget_s + 90)
                res.append(json.loads(o.strip().splitlines()[-1]))
            except Exception:  # noqa: BLE001 - a stuck or failed child is dropped, never waited for
                pr.kill()
        if res:
            steps = sum(r["steps"] for r in res)
            wall = max(r["wall"] for r in res)
            out = {"value": steps / wall, "unit": "env steps/s", "cores": len(res), "kind": "port",
                   "cores_present": hc["present"], "cores_usable": hc["usable"], "per_core": steps / wall / len(res),
                   "sample": f"{len(res)} processes (every core this job may use: {hc['usable']} of {hc['present']} present; "
                             f"affinity {hc['affinity']}, cgroup quota {hc['cgroup_quota']}) x {n} envs of the same config, each stepped "
                             f"serially through oracle/cport.c (C, -O2) for {wall:.1f} s"}
    if out is None:
        from oracle import oracle as O
        rng = np.random.default_rng(seed)
        frames = rng.integers(0, 256, (n, 2, 210, 160, 3), dtype=np.uint8)
        acts = rng.uniform(-5, 60, (n, 2))
        kw = dict(obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0), sensory_action_mode="absolute",
                  resize_to_full=True)
        ring = O.RingOracle(n, 4, (84, 84))
        fov = [O.FixedFovealOracle(**kw) for _ in range(n)]
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < budget_s:
            ring.ingest(frames, np.full(n, 2))
            full = ring.full_state()
            for i in range(n):
                fov[i].step(full[i], acts[i])
            reps += 1
        el = time.perf_counter() - t0
        out = {"value": n * reps / el, "unit": "env steps/s", "cores": 1, "kind": "port", "cores_present": hc["present"],
               "cores_usable": hc["usable"],
               "sample": f"{n} envs x {reps} steps of the same config through oracle/oracle.py (NumPy, single thread), {el:.1f} s"}
    try:
        out["reference_shaped"] = cpu_reference_shaped(min(6.0, budget_s), seed)
    except Exception as e:  # noqa: BLE001
        out["reference_shaped"] = {"error": repr(e)}
    return out


GAMES = ("breakout", "boxing", "pong", "seaquest", "qbert", "alien", "asterix", "freeway")      # configs[4]: one game per shard / GPU


def run_e2e(dev, n, hc, rank=0):
    """PCIe- and emulator-inclusive rate through the drop-in API (never `value`): AtariVecEnv.step = native C++ host
    runner (scripted emulator, one thread per core) -> pinned staging (two sets, alternating) -> hipMemcpyAsync ->
    ingest + fovea -> device observations; RGB screens (the metric's input format, north_star) and ALE grayscale screens (what the
    reference reads, atari_env.py:74).  Bounded: a few dozen steps each."""
    import numpy as np
    import torch
    from active_gym import AtariEnvArgs, AtariVecEnv
    # emulator threads: the product's default - the cores this job may use (affinity and cgroup quota) divided between the ranks of
    # the host, one pinned thread per CPU.  Measured on a 16-core quota (tools/e2e_phases.py, N = 1024, profiles/r04_e2e_phases_*.txt):
    # 12 / 16 threads 0.338 / 0.338 M env steps/s median with RGB screens, 0.98 / 0.98 M with gray screens (the PCIe copy either way);
    # 24 / 32 threads (two per CPU) reach the same best step but fall into 3-6 ms medians every few runs.  (Rounds 1-3 ran 2 x the
    # cores: the scripted emulator was four times as expensive then and its threads mostly waited on their screen writes.)
    lws = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))            # one process per GPU shares the host with its siblings
    workers = max(1, min(64, hc["usable"] // lws))
    out = {"envs": n, "runner": "libagx_runner.so (C++ threads, scripted emulator)", "workers": workers, "h2d_chunk_envs": 0,
           "local_world_size": lws,
           "overlap": "double-buffered pinned staging: the emulators of step t+1 run under the H2D copy and kernels of step t",
           "host_cores_usable": hc["usable"], "host_cores_present": hc["present"],
           "resets": "scripted life-loss / game-over events at 6 / 1 per mille per emulator frame: done envs are reset inside the "
                     "timed steps (autoreset: packed reset screens, one H2D copy + one scatter per step)"}
    h = torch.empty((n, 2, 210, 160, 3), dtype=torch.uint8).pin_memory()
    d = torch.empty_like(h, device=dev)
    for _ in range(2):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(5):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(dev)
    out["h2d_GBps"] = h.numel() * 5 / (time.perf_counter() - t0) / 1e9
    del h, d
    # motor actions on the host (the emulators need them), sensory actions as a device tensor (a policy's output lives on the GPU:
    # no per-step upload)
    act = {"motor_action": np.zeros(n, np.int64), "sensory_action": torch.full((n, 2), 20.0, dtype=torch.float32, device=dev)}
    # Every repeat is a window of >= 0.35 s: the job's CPU quota is enforced per 100 ms period (CFS bandwidth control), and 2 x quota
    # busy emulator threads run unthrottled for only half of a period - a 40 ms window (what rounds 1-3 timed) catches either
    # the burst or the stall; three periods and more give the rate the box sustains.
    window_s = 0.35
    for fmt in ("rgb", "gray"):
        from active_gym.sharding import shard_game
        game = shard_game(list(GAMES), rank)              # rank g plays GAMES[g % 8] (the game only selects the emulator, never a shape)
        args = AtariEnvArgs(frame_format=fmt, game=game, seed=1 + 100003 * rank, obs_size=(84, 84), fov_size=(30, 30), fov_init_loc=(0, 0),
                            sensory_action_mode="absolute", resize_to_full=True, frame_source="native", device=str(dev),
                            num_workers=workers, h2d_chunk_envs=0, scripted_lives=3, scripted_p_life=6, scripted_p_over=1)
        env = AtariVecEnv(args, n, kind="fixed")
        plan = env.host_plan
        out["game"] = game
        out["placement"] = {"numa_node": plan["numa_node"], "workers": env.runner.num_workers, "pinned_cpus": sorted(set(env.runner.worker_cpus)),
                            "cpus_usable": plan["usable"], "per_rank_default": plan["per_rank"],
                            "staging": "pinned buffers allocated while bound to the rank's CPUs (first touch on the GPU's NUMA node)"}
        out["step_loop"] = "native (agx_loop_step: one C call per step, autoreset inside)" if env._loop is not None else "python (active_gym/vector.py)"
        env.reset()
        for _ in range(8):                       # warm-up: both staging sets, both device sets, both reset sets have been through
            env.step(act)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(8):
            env.step(act)
        torch.cuda.synchronize(dev)
        steps = max(16, int(window_s / max((time.perf_counter() - t0) / 8, 1e-4)) + 1)
        times, dones = [], 0
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                dones += int(env.step(act)[2].sum())
            torch.cuda.synchronize(dev)
            times.append((time.perf_counter() - t0) / steps)
        shp = env.runner.frames_shape            # compact staging: only the rows K1 reads cross PCIe
        bytes_step, rows_staged = int(np.prod(shp)), int(shp[2])
        env.close()
        best, med, worst = min(times), sorted(times)[1], max(times)
        out[fmt] = {"ms_per_step": best * 1e3, "env_steps_per_s": n / best, "env_steps_per_s_median": n / med, "env_steps_per_s_min": n / worst,
                    "h2d_bytes_per_step": bytes_step, "rows_staged_per_screen": rows_staged,
                    "pcie_GBps_effective": bytes_step / best / 1e9, "steps_timed": steps, "window_s": steps * med,
                    "repeats": 3, "reset_fraction": dones / (3.0 * steps * n)}
    return out


def pmc_traffic(kernel, n_envs, kind, compact=False):
    """HBM bytes per launch of `kernel` from the committed PMC profile (profiles/r*_traffic*.json: FETCH_SIZE and
    WRITE_SIZE collected in separate rocprofv3 --pmc passes of this same command by tools/traffic.sh, FETCH_SIZE
    x2 as MI355X_MICROARCH.md prescribes for gfx950 - the factor re-measured on 12 B/lane and 16 B/lane reads of a
    known byte count).  Counters cannot be read from inside the process, so this is the profile's number, quoted
    only for the configuration it was collected on AND only when the profile was collected on the very build that is
    running (agx_build_info(): a hash of the kernel sources); otherwise None, with the reason."""
    import glob
    from active_gym import _native as nat
    if n_envs != 1024:
        return None, "profile exists for 1024 envs per GPU only"
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic*.json")))
    files = [f for f in files if not f.endswith(f"_{kind}.json")] + [f for f in files if f.endswith(f"_{kind}.json")]
    why = None
    for f in reversed(files):                   # newest round first, the file collected under this --kind before the others
        try:
            prof = json.load(open(f))
            e = prof["kernels"][kernel]
            if ("--compact" in str(prof.get("bench_args", ""))) != bool(compact):
                continue                        # a profile of the other input layout: a different K1 kernel
            if prof.get("build") != nat.build_info():
                if why is None:                 # name the NEWEST profile that does not match, not the oldest
                    why = f"{os.path.relpath(f, REPO)} was collected on '{prof.get('build')}', this library is '{nat.build_info()}': stale, not quoted"
                continue
            return {"traffic": float(e["traffic"]),
                    "note": f"{os.path.relpath(f, REPO)} ({e.get('kernel_name', kernel)}; build {prof['build']}): FETCH_SIZE x "
                            f"{e['fetch_factor_used']:.3f} + WRITE_SIZE, separate --pmc passes"}, None
        except (KeyError, ValueError, OSError, TypeError):
            continue
    return None, why or "no profiles/r*_traffic*.json"


def barrier(dist, local_rank):
    """All ranks rendezvous (no-op for a single process)."""
    if dist is not None and dist.is_initialized():
        if local_rank is None or dist.get_backend() != "nccl":
            dist.barrier()
        else:
            dist.barrier(device_ids=[local_rank])


def max_over_ranks(value, dist, device):
    """MAX of a per-rank scalar (the timed region's duration) over all ranks."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(value, dist, device):
    """Every rank's scalar, in rank order (a one-element list for a single process)."""
    if dist is None or not dist.is_initialized():
        return [float(value)]
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start the ranks as a child job (nothing here has
    touched the GPU yet) and exit with its code."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        relaunch_under_torchrun(args)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the observation path has no CPU implementation)")
    # Rehearsal hooks for a one-GPU box (never set by the driver): AGX_BENCH_SHARE_GPU=1 lets several ranks share
    # cuda:0, AGX_BENCH_BACKEND=gloo swaps the control-plane backend (RCCL refuses two ranks on one device),
    # AGX_BENCH_FORCE_DIST=1 initialises the process group even for a single rank (exercises RCCL itself).
    if os.environ.get("AGX_BENCH_SHARE_GPU") == "1":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # How the host waits in synchronize(): HIP's default on a host with more logical CPUs than contexts is to YIELD the core
    # (hipDeviceScheduleAuto), which adds a wake-up to the end of the timed region - 1-2 us per step of the driver's 20-step,
    # 1.1 ms region (profiles/r04_sched_ab.txt: 18.46-18.91 M yielding, 18.65-19.57 M spinning, pairwise + 1-3.5 %).  The
    # bench spins (hipDeviceScheduleSpin: the runtime polls for completion on the submitting thread); --host-wait auto restores
    # the runtime's default.  Nothing the GPU does changes.
    host_wait = os.environ.get("AGX_BENCH_SCHED", args.host_wait)
    if host_wait in ("spin", "yield"):
        import ctypes
        try:
            rc = ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(1 if host_wait == "spin" else 2)
        except OSError:
            rc = -1
        if rc != 0:
            host_wait = f"auto (hipSetDeviceFlags returned {rc})"
    dist = None
    if world > 1 or os.environ.get("AGX_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        backend = os.environ.get("AGX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)          # RCCL; used for barrier + MAX of the timing only
        else:
            dist.init_process_group(backend)

    n = args.envs
    packed_mode = args.kind == "flexible" and args.out == "packed"
    pipe = make_pipeline(args.kind, n, dev, antialias=bool(args.antialias), packed=packed_mode)
    gray = args.frame_format == "gray"
    frames, cmds, acts = synth_inputs(torch, dev, n, args.pool, 1234 + rank, gray)
    ingest_full = pipe.ingest_gray_raw if gray else pipe.ingest
    ingest_compact = pipe.ingest_gray_raw_compact if gray else pipe.ingest_compact
    src_rows = torch.from_numpy(pipe.source_rows()).to(dev).long()

    def compact_pool():
        """The same synthetic screens in the runner's compact staging layout: only the rows K1 reads, packed."""
        return [f.index_select(2, src_rows).contiguous() for f in frames]

    use_compact = bool(args.compact)
    if use_compact:
        frames = compact_pool()
    ingest = ingest_compact if use_compact else ingest_full
    types = None
    if args.kind == "flexible":
        g = torch.Generator(device=dev)
        g.manual_seed(99 + rank)
        types = [torch.randint(0, 2, (n,), dtype=torch.int32, device=dev, generator=g) for _ in range(args.pool)]
        for i in range(args.pool):       # FOV_RES actions carry integer resolutions in [10, 60]
            res = torch.randint(10, 61, (n, 2), device=dev, generator=g).float()
            acts[i] = torch.where(types[i][:, None] == 1, res, acts[i]).contiguous()
    obs = None if packed_mode else torch.empty(pipe.obs_shape, dtype=torch.float32, device=dev)
    loc = torch.empty((n, 2), dtype=torch.int32, device=dev)
    res_out = torch.empty((n, 2), dtype=torch.int32, device=dev)
    packed_buf = offsets = None
    if packed_mode:      # worst case: every env at the largest window the actions can ask for (60 x 60)
        packed_buf = torch.empty((n * pipe.frame_stack * 60 * 60,), dtype=torch.float32, device=dev)
        offsets = torch.empty((n + 1,), dtype=torch.int64, device=dev)

    kernel_events = args.event_mode == "kernel"
    issued = {"events": 0}          # HIP events handed to launches / recorded on the stream, counted per phase below
    state = {"ingest": ingest, "frames": frames, "obs": [obs]}      # what step() runs on (the later legs swap these)

    def step(k, e=None):
        """One pass of the hot path over batch k of the pool; `e` = the HIP events of a sampled step: 4 stamped by
        its two launches themselves (kernel mode) or 3 recorded on the stream around them (stream mode)."""
        i = k % args.pool
        out = state["obs"][k % len(state["obs"])]
        if e is not None:
            issued["events"] += len(e)
        if e is not None and kernel_events:
            pipe.profile_next("ingest", e[0], e[1])
            pipe.profile_next("fovea", e[2], e[3])
            e = None
        if e is not None:
            e[0].record()
        if packed_mode and args.packed_calls == 1:
            # the whole step in one ABI call and two launches (the layout is read off the screens' shape)
            pipe.step_flexible_packed(state["frames"][i], cmds[i], acts[i], action_type=types[i], packed=packed_buf, offsets=offsets,
                                      loc_out=loc, res_out=res_out)
            if e is not None:
                e[1].record()
                e[2].record()
            return
        state["ingest"](state["frames"][i], cmds[i])
        if e is not None:
            e[1].record()
        if packed_mode:
            pipe.fovea_packed(acts[i], action_type=types[i], packed=packed_buf, offsets=offsets, loc_out=loc, res_out=res_out)
        elif types is None:
            pipe.fovea(acts[i], out=out, loc_out=loc)
        else:
            pipe.fovea(acts[i], action_type=types[i], out=out, loc_out=loc, res_out=res_out)
        if e is not None:
            e[2].record()

    # the first collective of a process group builds its communicator (RCCL: hundreds of milliseconds): here, not in the barrier
    # in front of the timed region, where it would leave the device idle for that long right before t0
    barrier(dist, local_rank)
    # untimed pre-roll: the device leaves its idle clocks before anything is timed (a 20-step run is 1.3 ms long)
    for k in range(args.preroll):
        step(k)
    K = args.steps
    for _ in range(max(0, args.rehearsals)):
        torch.cuda.synchronize(dev)
        for k in range(K):
            step(k)
        torch.cuda.synchronize(dev)
    for k in range(args.warmup):
        step(k)
    barrier(dist, local_rank)
    torch.cuda.synchronize(dev)
    issued["events"] = 0
    t0 = time.perf_counter()
    # the timed region: exactly K plain steps - no event, no profiling hook, nothing but the launches
    for k in range(K):
        step(k)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    events_in_timed_region = issued["events"]
    barrier(dist, local_rank)
    per_rank = all_ranks(elapsed, dist, dev)                 # every rank's own duration of the timed region
    elapsed = max_over_ranks(elapsed, dist, dev)

    # the sampling pass (untimed, rank 0 only; the ranks share nothing): the same step loop continues, and every M-th step's two
    # launches carry their own start / stop events, on the launch stream (torch's current stream is the one handed to the C ABI)
    use_ev = not args.no_events and rank == 0
    M = max(1, args.event_every)
    S = max(16, args.samples)

    def sample(S_, k0):
        """S_ sampled launch pairs (every M-th step of S_ * M steps that continue the loop at step k0): mean (ingest, fovea) seconds"""
        ev_ = [[torch.cuda.Event(enable_timing=True) for _ in range(4 if kernel_events else 3)] for _ in range(S_)]
        for e in ev_:                  # create the HIP event handles
            for x in e:
                x.record()
        torch.cuda.synchronize(dev)
        for j in range(100):           # run-up: the synchronisation above let the device idle (see --rehearsals)
            step(k0 + j)
        k0 += 100
        for j in range(S_ * M):
            step(k0 + j, ev_[j // M] if j % M == M - 1 else None)
        torch.cuda.synchronize(dev)
        t_i = sum(e[0].elapsed_time(e[1]) for e in ev_) / len(ev_) * 1e-3
        t_f = sum((e[2].elapsed_time(e[3]) if kernel_events else e[1].elapsed_time(e[2])) for e in ev_) / len(ev_) * 1e-3
        return t_i, t_f

    ev = []
    t_ing = t_fov = None
    extra = {}
    if use_ev:
        t_ing, t_fov = sample(S, K)
        ev = [None] * S
        # (a) the fovea kernel against HBM: its output rotates through P buffers that together exceed the 256 MB Infinity Cache
        #     (tools/storebench `sizes`, profiles/r04_storebench_sizes.txt: the bare 115.6 MB store stream takes 17.1 us while
        #     the buffers it cycles through fit the cache - one or two of them - and 20.2 us from 3 buffers up)
        P = int(args.obs_pool)
        if P > 1 and not packed_mode and world == 1:
            state["obs"] = [obs] + [torch.empty_like(obs) for _ in range(P - 1)]
            for k in range(2 * P):
                step(k)
            ti_p, tf_p = sample(16, K + S * M)
            extra["obs_pool"] = (P, ti_p, tf_p)
            state["obs"] = [obs]
        # (b) the same step on compact input screens (what the host runner stages and agx_ingest_compact reads)
        if not use_compact and world == 1:
            state["frames"], state["ingest"] = compact_pool(), ingest_compact
            for k in range(200):             # (the device has idled through the sampling passes' synchronisations)
                step(k)
            torch.cuda.synchronize(dev)
            Kc = max(K, 100)
            tc0 = time.perf_counter()
            for k in range(Kc):
                step(k)
            torch.cuda.synchronize(dev)
            tc = (time.perf_counter() - tc0) / Kc
            ti_c, tf_c = sample(16, Kc)
            extra["compact"] = (tc, ti_c, tf_c, Kc)
            state["frames"], state["ingest"] = frames, ingest

    # the PCIe- and emulator-inclusive leg runs on EVERY rank at the same time (eight ranks share one host's cores, memory
    # system and PCIe root complexes: that is what it is there to show); rank 0 prints them all
    e2e_mine = None
    if not args.no_e2e:
        barrier(dist, local_rank)
        try:
            e2e_mine = run_e2e(dev, n, host_cores(), rank)
        except Exception as ex:  # noqa: BLE001 - a reported extra, never the metric
            e2e_mine = {"error": repr(ex)}
    e2e_all = [e2e_mine]
    if dist is not None and dist.is_initialized() and world > 1 and not args.no_e2e:
        e2e_all = [None] * world
        dist.all_gather_object(e2e_all, e2e_mine)

    out = None
    if rank == 0:
        total_envs = n * world
        ms_per_step = elapsed / K * 1e3
        kernels = {}
        roof = None
        if use_ev:
            b_ing, b_fov = pipe.algorithmic_bytes("ingest_gray_raw" if gray else "ingest"), pipe.algorithmic_bytes("fovea")
            if args.kind == "flexible":
                # 4*rh*rw (u8 windows of the 4 stacked frames) + 112,896 B per env, with rh*rw the mean over
                # the seeded resolution distribution as the envs hold it at the end of the run (the ABI's figure uses the
                # nominal 30x30 window)
                win_mean = float((res_out[:, 0].double() * res_out[:, 1].double()).mean().item())
                b_fov = int(n * pipe.frame_stack * (win_mean + 84 * 84 * 4))
                if packed_mode:      # ragged raw crops: rh*rw u8 read + rh*rw f32 written per stacked frame
                    b_fov = int(n * pipe.frame_stack * win_mean * 5)
            plan = (("k_ingest_grayraw" if gray else "k_ingest", b_ing, t_ing), ("k_fovea_" + args.kind, b_fov, t_fov))
            for name, b, t in plan:
                kernels[name] = {"avg_us": t * 1e6, "algorithmic_bytes": b, "achieved_GBps": b / t / 1e9,
                                 "frac": b / t / 1e9 / HBM_PEAK_GBS}
            dom = max(kernels, key=lambda k_: kernels[k_]["avg_us"])
            roof = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["achieved_GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": kernels[dom]["frac"], "traffic": None,
                    "avg_launch_us": kernels[dom]["avg_us"], "algorithmic_bytes_per_launch": kernels[dom]["algorithmic_bytes"],
                    "launches_timed": len(ev), "timing": (f"HIP start/stop events stamped by the launch itself (hipExtLaunchKernelGGL), on the launch stream, on every "
                               f"{M}th step of a {S * M}-step sampling pass that continues the step loop right after the timed region "
                               "(the timed region itself carries no events)" if kernel_events else
                               f"HIP events recorded on the launch stream around every {M}th step of a {S * M}-step sampling pass after the timed region")}
            tr, why = pmc_traffic(dom, n, args.kind, use_compact)
            if tr is not None:
                roof["traffic"] = tr["traffic"]
                roof["traffic_unit"] = "bytes per launch"
                roof["traffic_source"] = tr["note"]
                # the same launch priced on the bytes it actually moved (PMC) instead of the algorithmic ones: how close the kernel
                # runs to what the memory system streams at all; `frac` above stays the algorithmic figure
                roof["traffic_frac"] = tr["traffic"] / (kernels[dom]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
                roof["traffic_over_algorithmic"] = tr["traffic"] / kernels[dom]["algorithmic_bytes"]
            else:
                roof["traffic_source"] = why
        out = {
            "metric": "env steps/sec at N=1024 AtariFixedFovealEnv; 1/2/4/8-GPU scaling",
            "value": total_envs * K / elapsed, "unit": "env steps/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "preroll": args.preroll, "rehearsals": args.rehearsals, "events_in_timed_region": events_in_timed_region, "host_wait": host_wait,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 ingest / f32 resize", "data": "synthetic",
            "config": {"workload": workload_name(args, n, packed_mode, gray, use_compact),
                       "envs_per_gpu": n, "total_envs": total_envs, "input_pool": args.pool,
                       "input_layout": ("compact screens u8 [N,2,%d,160,%s]: the rows K1 reads, as the host runner stages them (agx_ingest_compact)"
                                        % (int(src_rows.numel()), "1" if gray else "3")) if use_compact else
                                       "whole screens u8 [N,2,210,160,%s]" % ("1" if gray else "3"),
                       "step_form": ("two stand-alone launches per step" if not packed_mode else
                                     "two launches per step (ingest + state / scan | crops)" if args.packed_calls == 1 else
                                     "three launches per step (ingest | state / scan | crops)"),
                       "parallelism": f"env-shard x{world}, no collective"},
            "roofline": roof, "kernels": kernels, "build": _build_info(),
            "per_gpu": [n * K / t_ for t_ in per_rank], "per_gpu_unit": "env steps/s of each rank over its own timed region",
            "control_plane": (dist.get_backend() if dist is not None and dist.is_initialized() else "none"),
        }
        if use_ev and "obs_pool" in extra:
            P, ti_p, tf_p = extra["obs_pool"]
            name = "k_fovea_" + args.kind
            out["kernels_obs_pool"] = {
                "buffers": P, "bytes": P * int(obs.numel()) * 4,
                name: {"avg_us": tf_p * 1e6, "frac": kernels[name]["algorithmic_bytes"] / tf_p / 1e9 / HBM_PEAK_GBS},
                "k_ingest": {"avg_us": ti_p * 1e6},
                "note": "the fovea kernel's output rotating through buffers that exceed the 256 MB Infinity Cache: its rate against "
                        "HBM; `kernels` is the product's form (two observation buffers, 231 MB, cache-resident)"}
        if use_ev and "compact" in extra:
            tc, ti_c, tf_c, Kc = extra["compact"]
            out["compact_input"] = {
                "value": n / tc, "unit": "env steps/s", "ms_per_step": tc * 1e3, "steps": Kc,
                "k_ingest_avg_us": ti_c * 1e6, "k_ingest_frac": kernels["k_ingest_grayraw" if gray else "k_ingest"]["algorithmic_bytes"] / ti_c / 1e9 / HBM_PEAK_GBS,
                "k_fovea_avg_us": tf_c * 1e6,
                "note": "the same step with the screens in the host runner's compact staging layout (only the %d of 210 rows K1 "
                        "reads, agx_ingest_compact): same results, same algorithmic bytes; never `value`" % int(src_rows.numel())}
        if world > 1:
            out["config"]["games"] = "one synthetic frame source per GPU (the game mix only changes the emulator, never the shapes)"
        if not args.no_e2e:
            if world == 1:
                out["e2e"] = e2e_all[0]
            else:
                out["e2e_per_rank"] = e2e_all
                ok = [x for x in e2e_all if isinstance(x, dict) and "error" not in x]
                try:
                    out["e2e_aggregate"] = {fmt: {"env_steps_per_s_median": sum(x[fmt]["env_steps_per_s_median"] for x in ok),
                                                  "env_steps_per_s_best": sum(x[fmt]["env_steps_per_s"] for x in ok)} for fmt in ("rgb", "gray")}
                    out["e2e_aggregate"]["ranks"] = len(ok)
                except (KeyError, TypeError) as ex:      # a rank's leg came back incomplete: the line still goes out
                    out["e2e_aggregate"] = {"error": repr(ex), "ranks": len(ok)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, 1234)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

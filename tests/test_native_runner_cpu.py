"""CPU: the C++ host runner (libagx_runner.so) against the Python runner (itself pinned to the reference's
control flow by tests/test_runner_cpu.py), both over the same scripted emulator: screens, command bytes,
rewards, dones, lives and life-termination flags must agree bit for bit, with resets interleaved."""
import importlib.util
import os

import numpy as np
import pytest

from lcg_ale import LcgALE

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the screen rows cv2.resize reads for 210 -> 84 rows (agx_source_rows of an 84 x 84 context): y0 = floor(2.5 dy + 0.75), y0 + 1
ROWS84 = np.array(sorted({int(np.floor((dy + 0.5) * 2.5 - 0.5)) + k for dy in range(84) for k in (0, 1)}), dtype=np.int32)


def _build():
    spec = importlib.util.spec_from_file_location("agx_build", os.path.join(REPO, "active-gym_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.build_runner()


class _Args:
    def __init__(self, **kw):
        self.__dict__.update(kw)


@pytest.mark.parametrize("ar,clip,training,n_act", [(4, False, True, 4), (4, True, False, 6), (3, False, True, 2), (6, False, True, 4)])
def test_native_runner_equals_python_runner(ar, clip, training, n_act):
    _build()
    from active_gym.native_runner import NativeHostRunner
    from active_gym.runner import AtariHostRunner
    N = 5
    common = dict(game="g", seed=77, action_repeat=ar, clip_reward=clip, max_episode_length=108e3,
                  scripted_actions=n_act, scripted_lives=3, scripted_p_life=60, scripted_p_over=15)
    py = AtariHostRunner(_Args(frame_source=lambda a, i: LcgALE(77 + i, n_act, 3, 60, 15), **common), N, workers=2,
                         noop_fn=lambda: 3, env_offset=0)
    nv = NativeHostRunner(_Args(**common), N, workers=3, noop_fn=lambda: 3, env_offset=0, backend="scripted")
    if not training:
        py.eval()
        nv.eval()
    assert nv.num_actions == py.num_actions == n_act
    rng = np.random.default_rng(1)
    ca, cb = py.reset(), nv.reset()
    assert np.array_equal(ca, cb) and np.array_equal(py.frames[:, 0], nv.frames[:, 0])
    assert np.array_equal(py.lives, nv.lives)
    n_done = 0
    for step in range(120):
        m = rng.integers(0, n_act, N)
        a, b = py.step(m), nv.step(m)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), step
        nvalid = a[2]
        for i in range(N):                              # only the sampled slots are defined
            for s in range(int(nvalid[i])):
                assert np.array_equal(py.frames[i, s], nv.frames[i, s]), (step, i, s)
        assert np.array_equal(py.lives, nv.lives) and np.array_equal(py.life_termination, nv.life_termination)
        d = np.nonzero(a[1])[0]
        if len(d):
            n_done += len(d)
            ra = np.zeros((N, 1, 210, 160, 3), np.uint8)
            rb = np.zeros((N, 1, 210, 160, 3), np.uint8)
            # alternate the two layouts of the reset screens: env i's in row i, or - packed, what the vector env's
            # autoreset uploads as one copy - the j-th reset env's in row j
            packed = bool(step % 2)
            assert np.array_equal(py.reset(d, out=ra, packed=packed), nv.reset(d, out=rb, packed=packed))
            assert np.array_equal(ra, rb)
            rows = range(len(d)) if packed else d
            assert all(ra[r].any() for r in rows) and not np.delete(ra, list(rows), axis=0).any()
    assert n_done >= 5
    py.close()
    nv.close()


def test_native_runner_errors_and_sharding_identity():
    _build()
    from active_gym.native_runner import NativeHostRunner
    common = dict(game="g", seed=5, action_repeat=4, clip_reward=False, max_episode_length=108e3)
    r = NativeHostRunner(_Args(**common), 4, backend="scripted")
    with pytest.raises(RuntimeError, match="motor action"):
        r.step([0, 1, 9, 0])
    full = NativeHostRunner(_Args(**common), 6, noop_fn=lambda: 1, backend="scripted")
    part = NativeHostRunner(_Args(**common), 3, noop_fn=lambda: 1, env_offset=3, backend="scripted")
    full.reset(); part.reset()
    assert np.array_equal(full.frames[3:, 0], part.frames[:, 0])          # env identity = seed + global index
    with pytest.raises(RuntimeError, match="backend"):
        NativeHostRunner(_Args(**common), 2, backend="nope")


def test_chunked_async_step_equals_blocking_step():
    _build()
    from active_gym.native_runner import NativeHostRunner
    common = dict(game="g", seed=11, action_repeat=4, clip_reward=False, max_episode_length=108e3, scripted_p_life=40,
                  scripted_p_over=10)
    N = 37
    a = NativeHostRunner(_Args(**common), N, workers=4, noop_fn=lambda: 2, backend="scripted")
    b = NativeHostRunner(_Args(**common), N, workers=3, noop_fn=lambda: 2, backend="scripted")
    a.reset(); b.reset()
    rng = np.random.default_rng(0)
    for step in range(40):
        m = rng.integers(0, 4, N)
        ra = a.step(m)
        nc = b.step_begin(m, 8)
        assert nc == 5
        with pytest.raises(RuntimeError, match="in flight|not been waited"):
            b.reset([0]) if step % 2 else b.step_begin(m, 8)
        for c in range(nc):
            b.step_wait(c)
            lo, hi = c * 8, min(N, c * 8 + 8)
            for i in range(lo, hi):                                  # chunk c is complete once its wait returns
                for s_ in range(int(b._cmd[i])):
                    assert np.array_equal(a.frames[i, s_], b.frames[i, s_])
        rb = b.step_finish()
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)
        d = np.nonzero(ra[1])[0]
        if len(d):
            oa = np.zeros((N, 1, 210, 160, 3), np.uint8); ob = np.zeros_like(oa)
            assert np.array_equal(a.reset(d, out=oa), b.reset(d, out=ob)) and np.array_equal(oa, ob)
    with pytest.raises(RuntimeError, match="chunk"):
        b.step_wait(99)


def test_ale_c_backend_through_a_stand_in_libale(tmp_path, monkeypatch):
    """The dlopen backend end to end: tests/fake_libale.c exports atari_py's libale_c entry points over the scripted
    emulator (real ALE is not in the image).  Checks the reference's five settings reach the library before loadROM
    (atari_env.py:45-50), the minimal action set indirection (atari_env.py:51-52), and bit-exact agreement with the
    Python runner."""
    import subprocess
    import sys
    import types
    _build()
    so = str(tmp_path / "libale_c.so")
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", os.path.join(REPO, "tests", "fake_libale.c"), "-o", so], check=True)
    from active_gym import native_runner as nr
    from active_gym.runner import AtariHostRunner
    fake = types.ModuleType("atari_py")
    fake.get_game_path = lambda game: f"/roms/{game}.bin"
    monkeypatch.setitem(sys.modules, "atari_py", fake)
    monkeypatch.setattr(nr, "_find_libale_c", lambda: (so, fake))
    N = 4
    common = dict(game="pong", seed=31, action_repeat=4, clip_reward=False, max_episode_length=108e3)
    dflt = nr.NativeHostRunner(_Args(**common), 1, backend="ale_c")
    assert dflt.gray and dflt.frames.shape == (1, 2, 210, 160)        # real ALE: its own grayscale screens by default (atari_env.py:74)
    dflt.close()
    common["frame_format"] = "rgb"                                    # the comparison below is on RGB screens
    nv = nr.NativeHostRunner(_Args(**common), N, workers=2, noop_fn=lambda: 4, backend="ale_c")

    class _ALE(LcgALE):                                  # same script, libale's action set
        def getMinimalActionSet(self):
            return [0, 1, 3, 4]

    py = AtariHostRunner(_Args(frame_source=lambda a, i: _ALE(31 + i, 4, 3, 60, 15), **common), N, workers=1, noop_fn=lambda: 4)
    assert nv.num_actions == 4
    assert np.array_equal(py.reset(), nv.reset()) and np.array_equal(py.frames[:, 0], nv.frames[:, 0])
    assert np.array_equal(py.lives, nv.lives) and (nv.lives > 0).all()       # -99 would mean the settings were wrong
    rng = np.random.default_rng(4)
    dones = 0
    for step in range(80):
        m = rng.integers(0, 4, N)
        a, b = py.step(m), nv.step(m)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), step
        for i in range(N):
            for s in range(int(a[2][i])):
                assert np.array_equal(py.frames[i, s], nv.frames[i, s])
        d = np.nonzero(a[1])[0]
        dones += len(d)
        if len(d):
            ra = np.zeros((N, 1, 210, 160, 3), np.uint8); rb = np.zeros_like(ra)
            assert np.array_equal(py.reset(d, out=ra), nv.reset(d, out=rb)) and np.array_equal(ra, rb)
    assert dones >= 3
    nv.close(); py.close()
    # compact staging through this backend: a real emulator renders its whole screen into a per-thread scratch buffer and the listed
    # rows are copied out (Emulator::screen_rgb_rows / screen_gray_rows defaults) - RGB and the library's own grayscale screens
    for fmt in ("rgb", "gray"):
        common["frame_format"] = fmt
        full = nr.NativeHostRunner(_Args(**common), N, workers=2, noop_fn=lambda: 4, backend="ale_c")
        comp = nr.NativeHostRunner(_Args(**common), N, workers=2, noop_fn=lambda: 4, backend="ale_c", src_rows=ROWS84)
        assert comp.frames.shape[2] == 168
        assert np.array_equal(full.reset(), comp.reset()) and np.array_equal(full.frames[:, 0][:, ROWS84], comp.frames[:, 0])
        for step in range(12):
            m = rng.integers(0, 4, N)
            a, b = full.step(m), comp.step(m)
            for x, y in zip(a, b):
                assert np.array_equal(x, y), (fmt, step)
            for i in range(N):
                for sl in range(int(a[2][i])):
                    assert np.array_equal(full.frames[i, sl][ROWS84], comp.frames[i, sl]), (fmt, step, i, sl)
        full.close(); comp.close()
    common["frame_format"] = "rgb"
    with pytest.raises(RuntimeError, match="dlopen|symbol"):
        monkeypatch.setattr(nr, "_find_libale_c", lambda: (str(tmp_path / "missing.so"), fake))
        nr.NativeHostRunner(_Args(**common), 1, backend="ale_c")


def test_gray_frames_native_equals_python_runner():
    """frame_format="gray": both runners hand over ALE-style grayscale screens u8[N,2,210,160] (what the reference
    reads, atari_env.py:74); for the scripted emulator gray == ALE luminance of its RGB screen."""
    _build()
    from active_gym.native_runner import NativeHostRunner
    from active_gym.runner import AtariHostRunner
    from oracle import oracle as O
    N = 3
    common = dict(game="g", seed=9, action_repeat=4, clip_reward=False, max_episode_length=108e3, frame_format="gray",
                  scripted_p_life=40, scripted_p_over=10)
    py = AtariHostRunner(_Args(frame_source=lambda a, i: LcgALE(9 + i, 4, 3, 40, 10), **common), N, workers=1, noop_fn=lambda: 2)
    nv = NativeHostRunner(_Args(**common), N, workers=2, noop_fn=lambda: 2, backend="scripted")
    assert py.frames.shape == nv.frames.shape == (N, 2, 210, 160)
    assert np.array_equal(py.reset(), nv.reset()) and np.array_equal(py.frames[:, 0], nv.frames[:, 0])
    e = LcgALE(9, 4, 3, 40, 10)
    assert np.array_equal(e.getScreenGrayscale()[..., 0], O.ale_luminance(e.getScreenRGB()))
    rng = np.random.default_rng(2)
    for step in range(40):
        m = rng.integers(0, 4, N)
        a, b = py.step(m), nv.step(m)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        for i in range(N):
            for s_ in range(int(a[2][i])):
                assert np.array_equal(py.frames[i, s_], nv.frames[i, s_])
        d = np.nonzero(a[1])[0]
        if len(d):
            ra = np.zeros((N, 1, 210, 160), np.uint8); rb = np.zeros_like(ra)
            assert np.array_equal(py.reset(d, out=ra), nv.reset(d, out=rb)) and np.array_equal(ra, rb)
    py.close(); nv.close()



@pytest.mark.parametrize("gray", [False, True])
def test_compact_staging_is_the_listed_rows_of_whole_screens(gray):
    """agxr_config.src_rows: the runner stages only the listed rows - bit for bit the rows of what it stages otherwise, for step
    screens, reset screens and packed reset screens, native and Python runner alike."""
    _build()
    from active_gym.native_runner import NativeHostRunner
    from active_gym.runner import AtariHostRunner
    assert len(ROWS84) == 168 and ROWS84[0] == 0 and ROWS84[-1] == 209
    N, n_act = 6, 4
    common = dict(game="g", seed=31, action_repeat=4, clip_reward=False, max_episode_length=108e3, frame_format="gray" if gray else "rgb",
                  scripted_actions=n_act, scripted_lives=2, scripted_p_life=60, scripted_p_over=20)
    full = NativeHostRunner(_Args(**common), N, workers=2, noop_fn=lambda: 2, backend="scripted")
    comp = NativeHostRunner(_Args(**common), N, workers=3, noop_fn=lambda: 2, backend="scripted", src_rows=ROWS84)
    pyc = AtariHostRunner(_Args(frame_source=lambda a, i: LcgALE(31 + i, n_act, 2, 60, 20), **common), N, workers=2, noop_fn=lambda: 2,
                          src_rows=ROWS84)
    px = () if gray else (3,)
    assert full.frames.shape == (N, 2, 210, 160) + px and comp.frames.shape == pyc.frames.shape == (N, 2, 168, 160) + px
    assert np.array_equal(full.reset(), comp.reset()) and np.array_equal(full.frames[:, 0][:, ROWS84], comp.frames[:, 0])
    pyc.reset()
    assert np.array_equal(pyc.frames[:, 0], comp.frames[:, 0])
    rng = np.random.default_rng(2)
    resets = 0
    for step in range(60):
        m = rng.integers(0, n_act, N)
        a, b, c = full.step(m), comp.step(m), pyc.step(m)
        for x, y, z in zip(a, b, c):
            assert np.array_equal(x, y) and np.array_equal(x, z), step
        for i in range(N):
            for s in range(int(a[2][i])):
                assert np.array_equal(full.frames[i, s][ROWS84], comp.frames[i, s]), (step, i, s)
                assert np.array_equal(pyc.frames[i, s], comp.frames[i, s]), (step, i, s)
        d = np.nonzero(a[1])[0]
        if len(d):
            resets += len(d)
            packed = bool(step % 2)
            ra = np.zeros((N, 1, 210, 160) + px, np.uint8)
            rb = np.zeros((N, 1, 168, 160) + px, np.uint8)
            rc = np.zeros((N, 1, 168, 160) + px, np.uint8)
            ca, cb, cc = full.reset(d, out=ra, packed=packed), comp.reset(d, out=rb, packed=packed), pyc.reset(d, out=rc, packed=packed)
            assert np.array_equal(ca, cb) and np.array_equal(ca, cc)
            assert np.array_equal(ra[:, 0][:, ROWS84], rb[:, 0]) and np.array_equal(rb, rc)
    assert resets >= 4
    for r in (full, comp, pyc):
        r.close()


def test_compact_rows_are_validated():
    _build()
    from active_gym.native_runner import NativeHostRunner
    common = dict(game="g", seed=5, action_repeat=4, clip_reward=False, max_episode_length=108e3)
    for bad in ([3, 2], [0, 0], [-1, 4], [5, 210]):
        with pytest.raises(RuntimeError, match="src_rows"):
            NativeHostRunner(_Args(**common), 2, backend="scripted", src_rows=bad)


def test_workers_are_pinned_to_the_listed_cpus():
    _build()
    from active_gym.native_runner import NativeHostRunner
    common = dict(game="g", seed=5, action_repeat=4, clip_reward=False, max_episode_length=108e3)
    allowed = sorted(os.sched_getaffinity(0))
    cpus = allowed[:2]
    r = NativeHostRunner(_Args(**common), 8, workers=5, backend="scripted", cpus=cpus)
    assert r.num_workers == 5 and r.worker_cpus == [cpus[w % len(cpus)] for w in range(5)]
    r.reset()
    r.step(np.zeros(8, np.int64))                       # the pinned pool still steps every env
    r.close()
    u = NativeHostRunner(_Args(**common), 8, workers=3, backend="scripted")
    assert u.num_workers == 3 and u.worker_cpus == [-1, -1, -1]
    u.close()
    # no explicit worker count: the library's own default (usable CPUs // LOCAL_WORLD_SIZE), capped by the env count
    d = NativeHostRunner(_Args(**common), 2, backend="scripted")
    assert d.num_workers == min(2, d._lib.agxr_default_threads())
    d.close()


@pytest.mark.parametrize("sanitizer,args", [("thread", ["36", "13"]), ("address,undefined", ["120", "23"])])
def test_runner_under_sanitizers(tmp_path, sanitizer, args):
    """csrc/agx_runner.cpp compiled with ThreadSanitizer / AddressSanitizer + UBSan into tests/runner_sanitizer_harness.cpp and driven
    from plain C++ (no Python in the process, so every report is the runner's): blocking and chunked steps, full and packed resets,
    render, state queries, whole / compact and RGB / gray staging, 1 / 3 / 8 worker threads - no report, and the same checksum for
    every thread count."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / "harness"
    cmd = ["g++", "-O1", "-g", "-fno-omit-frame-pointer", f"-fsanitize={sanitizer}", "-mavx2", "-ffp-contract=off", "-std=c++17",
           "-I", os.path.join(REPO, "include"), os.path.join(REPO, "tests", "runner_sanitizer_harness.cpp"),
           os.path.join(REPO, "active-gym_amd", "csrc", "agx_runner.cpp"), "-o", str(exe), "-pthread", "-ldl"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "sanitizer" in b.stderr.lower() and "cannot find" in b.stderr.lower():
        pytest.skip("sanitizer runtime not installed: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="halt_on_error=1 detect_leaks=1 exitcode=67",
               UBSAN_OPTIONS="halt_on_error=1 print_stacktrace=1 exitcode=68")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(exe)] + args, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("gray=")]
    assert len(lines) == 12 and not any("DIFFERS" in ln for ln in lines), r.stdout
    if sanitizer != "thread":
        # the gray rows' AVX-512 VBMI table walk against the scalar one (AGXR_NO_VBMI=1): the same checksums
        r2 = subprocess.run([str(exe)] + args, capture_output=True, text=True, env=dict(env, AGXR_NO_VBMI="1"), timeout=600)
        assert r2.returncode == 0 and r2.stdout == r.stdout, (r2.returncode, r2.stdout[-800:], r2.stderr[-2000:])

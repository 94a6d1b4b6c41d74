"""CPU: host-side rules added in round 2 - screen format defaults, motor-action validation, scalar sensory actions,
per-env no-op streams (what makes a sharded run reproduce the unsharded one)."""
import numpy as np
import pytest
import torch

from active_gym.frame_source import resolve_frame_format
from active_gym.runner import AtariHostRunner, check_motor_actions, per_env_noop_seed
from fake_ale import ScriptedALE


class A:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def test_real_emulators_default_to_ale_grayscale_screens():
    # what the reference reads: ale.getScreenGrayscale() (atari_env.py:74)
    assert resolve_frame_format(A(frame_source="ale")) == "gray"
    assert resolve_frame_format(A()) == "gray"                                  # frame_source defaults to "ale"
    assert resolve_frame_format(A(frame_source="native:ale")) == "gray"
    # synthetic / scripted sources keep the metric's RGB workload; an explicit choice always wins
    assert resolve_frame_format(A(frame_source="synthetic")) == "rgb"
    assert resolve_frame_format(A(frame_source="native")) == "rgb"
    assert resolve_frame_format(A(frame_source=lambda a, i: None)) == "rgb"
    assert resolve_frame_format(A(frame_source="ale", frame_format="rgb")) == "rgb"
    assert resolve_frame_format(A(frame_source="native", frame_format="gray")) == "gray"
    with pytest.raises(ValueError):
        resolve_frame_format(A(frame_format="bgr"))


def test_motor_actions_are_validated_not_wrapped():
    assert check_motor_actions(np.array([0, 3]), 4).tolist() == [0, 3]
    assert check_motor_actions(np.array([2.0]), 4).tolist() == [2]
    assert check_motor_actions(torch.tensor([1, 2]).numpy(), 4).tolist() == [1, 2]
    for bad in ([-1], [4], [1.5]):
        with pytest.raises(ValueError):
            check_motor_actions(np.array(bad), 4)
    with pytest.raises(TypeError):
        check_motor_actions(np.array(["a"]), 4)


def _args():
    return A(game="g", seed=7, action_repeat=4, clip_reward=False,
             frame_source=lambda args, i: ScriptedALE(seed=50 + i, n_actions=4, p_life=0.2, p_over=0.05))


def test_python_runner_rejects_out_of_range_motor_action():
    r = AtariHostRunner(_args(), 2, workers=1, noop_fn=lambda: 0)
    r.reset()
    with pytest.raises(ValueError):
        r.step(np.array([0, -1]))
    r.close()


def test_per_env_noop_streams_do_not_depend_on_the_shard():
    """noop_per_env: env i draws its reset no-op counts from random.Random(per_env_noop_seed(seed, GLOBAL i)); the
    screens of envs [2, 5) stepped as a shard equal those of the same envs inside the full batch."""
    N, lo, hi = 6, 2, 5
    full = AtariHostRunner(_args(), N, workers=1, noop_per_env=True)
    part = AtariHostRunner(_args(), hi - lo, workers=1, noop_per_env=True, env_offset=lo)
    assert per_env_noop_seed(7, 3) == per_env_noop_seed(7, 3) != per_env_noop_seed(7, 4)
    cf, cp = full.reset(), part.reset()
    assert np.array_equal(cf[lo:hi], cp) and np.array_equal(full.frames[lo:hi, 0], part.frames[:, 0])
    rng = np.random.default_rng(0)
    for _ in range(12):
        m = rng.integers(0, 4, N)
        rf, df, cf, _ = full.step(m)
        rp, dp, cp, _ = part.step(m[lo:hi])
        assert np.array_equal(rf[lo:hi], rp) and np.array_equal(df[lo:hi], dp) and np.array_equal(cf[lo:hi], cp)
        assert np.array_equal(full.frames[lo:hi], part.frames)
        if df.any():
            full.reset(np.nonzero(df)[0])
        if dp.any():
            part.reset(np.nonzero(dp)[0])
        assert np.array_equal(full.frames[lo:hi, 0], part.frames[:, 0])
    full.close()
    part.close()


def test_scalar_sensory_action_is_broadcast_like_np_clip():
    """The reference's sensory_action space is a scalar Box (fov_env.py:125-129): `sample()` is a 0-d array and
    np.clip(loc, 0, obs - fov) broadcasts it to (a, a) (fov_env.py:166-167)."""
    from active_gym.fov_env import FixedFovealEnv
    one = FixedFovealEnv._one
    assert one(np.array(17), 2).tolist() == [[17, 17]]
    assert one(np.int64(3), 2).tolist() == [[3, 3]]
    assert one(torch.tensor(5.5), 2).tolist() == [[5.5, 5.5]]
    assert one(np.array([4, 9]), 2).tolist() == [[4, 9]]
    assert one(torch.tensor([[1.0, 2.0]]), 2).tolist() == [[1.0, 2.0]]


def test_device_guard_restore_semantics_with_a_mocked_runtime(tmp_path):
    import os
    """agx_device_guard.h (what every C-ABI entry point opens) against a mocked device runtime: a context on device d != 0
    used from a thread on another device - switch, restore, nesting, failing set / get.  An 8-GPU node runs this for real;
    a one-GPU box cannot (tests/test_gpu_env.py::test_vec_env_on_a_device_that_is_not_the_current_one is skipped there)."""
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "device_guard_harness")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(repo, "active-gym_amd", "csrc"),
                    os.path.join(repo, "tests", "device_guard_harness.cpp"), "-o", exe], check=True, timeout=120)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_every_device_entry_point_opens_a_device_guard():
    import os
    """Static audit of agx_api.hip: every extern "C" entry point that launches, copies or allocates names the context's
    device through DeviceGuard before its first HIP call (ordinal != 0 safety, SURVEY 8e)."""
    import re
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(repo, "active-gym_amd", "csrc", "agx_api.hip")).read()
    body = src[src.index('extern "C" {'):]
    funcs = re.findall(r"^(?:int|int64_t|const char \*)\s*(agx_\w+)\(.*?^}", body, flags=re.S | re.M)
    assert len(funcs) >= 20
    for m in re.finditer(r"^(?:int|int64_t|const char \*)\s*(agx_\w+)\((.*?)^}", body, flags=re.S | re.M):
        name, text = m.group(1), m.group(2)
        touches = re.search(r"hipLaunchKernelGGL|AGX_LAUNCH|hipMemcpy|hipMemset|hipMalloc|hipFree|hipEventRecord|hipStream", text)
        delegates = re.search(r"return (?:stack_launch|agx_ingest|agx_fovea_fixed)\(", text)
        if touches:
            first = touches.start()
            guard = text.find("DeviceGuard g(")
            assert 0 <= guard < first or (delegates and guard < 0 and name == "agx_step_fixed"), name


def test_host_observation_pool_recycles_only_unreferenced_buffers(monkeypatch):
    """active_gym/vector.py::_HostObsPool - the logic behind the default host (NumPy) outputs, without pinning (no GPU here): a
    buffer comes back only when the array handed out AND every view of it are gone; the budget caps the pinned buffers; another
    observation shape starts a new pool and stray give-backs of the old shape are ignored."""
    import gc
    import torch
    import active_gym.vector as v
    real_empty = torch.empty
    monkeypatch.setattr(v.torch, "empty", lambda shape, dtype=None, pin_memory=False: real_empty(shape, dtype=dtype))
    pool = v._HostObsPool(2)
    a = pool.hand_out(pool.take((4, 3), torch.float32))
    b = pool.hand_out(pool.take((4, 3), torch.float32))
    assert a.shape == (4, 3) and a.ctypes.data != b.ctypes.data
    assert pool.take((4, 3), torch.float32) is None                     # budget spent: the env falls back to a pageable array
    addr = a.ctypes.data
    view = a[1:]
    del a
    gc.collect()
    assert pool.take((4, 3), torch.float32) is None                     # the view keeps the buffer out
    del view
    gc.collect()
    t = pool.take((4, 3), torch.float32)
    assert t is not None and t.data_ptr() == addr
    c = pool.hand_out(t)
    # another shape: a new pool of its own budget; the old arrays stay valid and their late give-backs are dropped
    d = pool.hand_out(pool.take((2, 2), torch.float32))
    b[:] = 7
    del c
    gc.collect()
    assert len(pool._free) == 0 and pool._made == 1
    assert float(b.sum()) == 7 * 12 and d.shape == (2, 2)

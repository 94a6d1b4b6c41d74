"""agx_step_flexible_packed: one whole step of a flexible raw-crop context with packed ragged observations in two launches
(the state update + scan ride in the ingest launch) == agx_ingest* followed by agx_fovea_flexible_packed, bit for bit, for
every screen layout, across scan-block boundaries, and on geometries that take the three-launch form inside the call.
(The two stand-alone calls are held against the oracle in tests/test_gpu_parity.py.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pipes(n, obs=84, fov=30, antialias=True):
    from active_gym import ObsPipeline
    dev = torch.device("cuda:0")
    kw = dict(num_envs=n, kind="flexible", obs_size=(obs, obs), frame_stack=4, fov_size=(fov, fov), fov_init_loc=(2, 5),
              sensory_action_mode="absolute", resize_to_full=False, mask_out=False, antialias=antialias, device=dev)
    return ObsPipeline(**kw), ObsPipeline(**kw), dev


def _run(n, layout, steps=6, obs=84, fov=30, seed=3, antialias=True):
    a, b, dev = _pipes(n, obs, fov, antialias)
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    gray = "gray" in layout
    rows = torch.from_numpy(a.source_rows()).long()
    shape = (n, 2, 210, 160) + (() if gray else (3,))
    cap = n * 4 * obs * obs
    out = [dict(packed=torch.zeros(cap, dtype=torch.float32, device=dev), offsets=torch.zeros(n + 1, dtype=torch.int64, device=dev),
                loc_out=torch.zeros((n, 2), dtype=torch.int32, device=dev), res_out=torch.zeros((n, 2), dtype=torch.int32, device=dev))
           for _ in range(2)]
    for t in range(steps):
        whole = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g)
        scr = (whole.index_select(2, rows).contiguous() if "compact" in layout else whole).to(dev)
        cmd = torch.full((n,), 2, dtype=torch.uint8)
        cmd[torch.randint(0, n, (max(1, n // 6),), generator=g)] = 1
        if t == 3:
            cmd[0] = 0x04 | 1                   # CLEAR on one env, as an autoreset's ingest issues it
            cmd[n - 1] = 0x08                   # SKIP
        cmd = cmd.to(dev)
        typ = torch.randint(0, 2, (n,), dtype=torch.int32, generator=g)
        act = torch.where(typ[:, None] == 1, torch.randint(5, obs - 10, (n, 2), generator=g).float(),
                          torch.rand((n, 2), generator=g) * (obs + 10) - 5).contiguous().to(dev)
        typ = typ.to(dev)
        # A: the two stand-alone calls
        if "compact" in layout:
            (a.ingest_gray_raw_compact if gray else a.ingest_compact)(scr, cmd)
        else:
            (a.ingest_gray_raw if gray else a.ingest)(scr, cmd)
        a.fovea_packed(act, action_type=typ, **out[0])
        # B: the one call
        b.step_flexible_packed(scr, cmd, act, action_type=typ, **out[1])
        torch.cuda.synchronize()
        assert torch.equal(a.stack_u8(), b.stack_u8()), (layout, n, t)
        assert torch.equal(out[0]["offsets"], out[1]["offsets"]), (layout, n, t)
        assert torch.equal(out[0]["loc_out"], out[1]["loc_out"]) and torch.equal(out[0]["res_out"], out[1]["res_out"])
        total = int(out[0]["offsets"][-1])
        assert total == int((4 * out[0]["res_out"][:, 0].long() * out[0]["res_out"][:, 1].long()).sum())
        pa, pb = out[0]["packed"][:total].cpu().numpy(), out[1]["packed"][:total].cpu().numpy()
        assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32)), (layout, n, t)
    # the context's own state agrees as well, and the contexts stay interchangeable: one more stand-alone observation on each
    la, ra = a.fov_state()
    lb, rb = b.fov_state()
    assert torch.equal(la, lb) and torch.equal(ra, rb)
    a.fovea_packed(None, **out[0])
    b.fovea_packed(None, **out[1])
    torch.cuda.synchronize()
    assert torch.equal(out[0]["offsets"], out[1]["offsets"])


@pytest.mark.parametrize("layout", ["rgb", "rgb-compact", "gray", "gray-compact"])
@pytest.mark.parametrize("n", [5, 300])
def test_one_call_step_equals_ingest_then_packed_fovea(layout, n):
    _run(n, layout)


def test_one_call_step_across_five_scan_blocks_antialias_off():
    _run(1100, "rgb-compact", steps=3, antialias=False)


def test_one_call_step_on_a_geometry_without_the_band12_plan():
    # 64 x 64 observations: the general ingest kernel and (fov 20) the raw3 crop plan or its fallback - the call then issues the
    # stand-alone launches itself
    _run(9, "rgb", steps=4, obs=64, fov=20)
    _run(9, "gray-compact", steps=4, obs=64, fov=20)


def test_unfused_knob_takes_the_three_launch_form(monkeypatch):
    monkeypatch.setenv("AGX_STEP_PACKED_UNFUSED", "1")
    _run(40, "rgb", steps=3)


def test_one_call_step_rejects_what_the_two_calls_reject():
    from active_gym import ObsPipeline
    from active_gym import _native as nat
    dev = torch.device("cuda:0")
    p = ObsPipeline(num_envs=4, kind="flexible", obs_size=(84, 84), frame_stack=4, fov_size=(30, 30), fov_init_loc=(0, 0),
                    sensory_action_mode="absolute", resize_to_full=True, device=dev)
    with pytest.raises(RuntimeError):
        p.step_flexible_packed(torch.zeros((4, 2, 210, 160, 3), dtype=torch.uint8, device=dev), torch.full((4,), 2, dtype=torch.uint8, device=dev))
    q = ObsPipeline(num_envs=4, kind="flexible", obs_size=(84, 84), frame_stack=4, fov_size=(30, 30), fov_init_loc=(0, 0),
                    sensory_action_mode="absolute", resize_to_full=False, device=dev)
    with pytest.raises(ValueError):
        q.step_flexible_packed(torch.zeros((4, 2, 100, 160, 3), dtype=torch.uint8, device=dev), torch.full((4,), 2, dtype=torch.uint8, device=dev))
    # the ABI itself: unknown layout bits, null buffers
    import ctypes as C
    rc = q._lib.agx_step_flexible_packed(q._ctx, C.c_void_p(1), 8, C.c_void_p(1), None, 0, None, C.c_void_p(1), 0, C.c_void_p(1), None, None, None)
    assert rc == nat.E_INVALID
    rc = q._lib.agx_step_flexible_packed(q._ctx, None, 0, C.c_void_p(1), None, 0, None, C.c_void_p(1), 0, C.c_void_p(1), None, None, None)
    assert rc == nat.E_INVALID

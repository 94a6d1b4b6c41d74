"""CPU: the C-ABI library builds for gfx950, loads without a GPU, and exports
exactly the symbols include/agx.h declares (no compute calls here)."""
import ctypes
import importlib.util
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    spec = importlib.util.spec_from_file_location("agx_build", os.path.join(REPO, "active-gym_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.build()


def _declared():
    src = open(os.path.join(REPO, "include", "agx.h")).read()
    return sorted(set(re.findall(r"^AGX_API[^;(]*?\b(agx_\w+)\s*\(", src, flags=re.M)))


def test_header_declares_the_expected_surface():
    names = _declared()
    assert len(names) == 28 and "agx_step_flexible_packed" in names and "agx_ingest" in names and "agx_ingest_compact" in names and "agx_source_rows" in names and "agx_fovea_flexible" in names


def test_library_exports_every_declared_symbol():
    path = _build()
    handle = ctypes.CDLL(path)
    for name in _declared():
        assert hasattr(handle, name), name
    assert handle.agx_abi_version() == 2


def test_binding_matches_header():
    from active_gym import _native as nat
    assert sorted(nat.SIGNATURES) == _declared()
    _build()
    lib = nat.lib()
    assert lib.agx_abi_version() == nat.ABI_VERSION
    assert ctypes.sizeof(nat.AgxConfig) == 96
    spec = importlib.util.spec_from_file_location("agx_build", os.path.join(REPO, "active-gym_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert nat.build_info() == f"libagx abi {nat.ABI_VERSION} src {m.source_hash()}"      # the loaded .so is built from the tree's sources


def test_create_fails_loudly_without_gpu_or_bad_config():
    import torch
    from active_gym import _native as nat
    _build()
    lib = nat.lib()
    cfg = nat.AgxConfig()
    cfg.struct_size = 4            # wrong on purpose: rejected before any HIP call
    ctx = ctypes.c_void_p()
    rc = lib.agx_create(ctypes.byref(cfg), ctypes.byref(ctx))
    assert rc == nat.E_INVALID and "struct_size" in nat.last_error(None) and not ctx.value
    if not torch.cuda.is_available():
        from active_gym import ObsPipeline
        with pytest.raises(RuntimeError, match="no CPU implementation"):
            ObsPipeline(4, "fixed", fov_size=(30, 30))


def test_runner_library_exports_header_surface():
    """include/agx_runner.h (native host runner) <-> libagx_runner.so <-> ctypes binding."""
    spec = importlib.util.spec_from_file_location("agx_build", os.path.join(REPO, "active-gym_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    path = m.build_runner()
    src = open(os.path.join(REPO, "include", "agx_runner.h")).read()
    names = sorted(set(re.findall(r"^AGXR_API[^;(]*?\b(agxr_\w+)\s*\(", src, flags=re.M)))
    assert len(names) == 16
    handle = ctypes.CDLL(path)
    for name in names:
        assert hasattr(handle, name), name
    from active_gym import native_runner as nr
    assert sorted(nr.SIGNATURES) == names
    cfg = nr.AgxrConfig()
    cfg.struct_size = 4
    h = ctypes.c_void_p()
    assert nr.lib().agxr_create(ctypes.byref(cfg), ctypes.byref(h)) != 0 and not h.value
    assert b"struct_size" in nr.lib().agxr_last_error(None)


def test_loop_header_surface_is_exported_and_bound():
    """include/agx_loop.h (the native step loop inside libagx.so) <-> the library's exports <-> active_gym/native_loop.py; the
    callback table's layout; a loop cannot be created without a context (no compute calls here)."""
    path = _build()
    src = open(os.path.join(REPO, "include", "agx_loop.h")).read()
    names = sorted(set(re.findall(r"^AGX_API[^;(]*?\b(agx_loop_\w+)\s*\(", src, flags=re.M)))
    assert names == ["agx_loop_create", "agx_loop_destroy", "agx_loop_last_error", "agx_loop_reset", "agx_loop_reset_envs", "agx_loop_step"]
    handle = ctypes.CDLL(path)
    for name in names:
        assert hasattr(handle, name), name
    from active_gym import native_loop as nl
    assert sorted(nl.SIGNATURES) == names
    assert ctypes.sizeof(nl.AgxHostSource) == 5 * ctypes.sizeof(ctypes.c_void_p) and ctypes.sizeof(nl.AgxLoopConfig) == 16
    assert ctypes.sizeof(nl.AgxLoopResult) == 72
    lib = nl._lib()
    h = ctypes.c_void_p()
    cfg = nl.AgxLoopConfig(ctypes.sizeof(nl.AgxLoopConfig), 0, 1, 1)
    assert lib.agx_loop_create(None, ctypes.byref(nl.AgxHostSource()), ctypes.byref(cfg), ctypes.byref(h)) != 0 and not h.value
    assert b"null argument" in lib.agx_loop_last_error(None)
    # the host source's entry points are those of libagx_runner.so, with exactly the callback signatures the loop declares
    hdr = open(os.path.join(REPO, "include", "agx_runner.h")).read()
    assert "agxr_step(agxr_runner *r, const int32_t *motor, uint8_t *frames, uint8_t *cmd, double *reward,\n" in hdr.replace("AGXR_API int ", "")
    assert "agxr_reset_packed(agxr_runner *r, const int32_t *idx, int32_t k, const int32_t *noops, uint8_t *frames,\n" in hdr.replace("AGXR_API int ", "")


def test_binding_constants_mirror_the_header():
    """Every numeric `#define AGX_<NAME> <value>` of include/agx.h has a twin `<NAME>` in active_gym/_native.py with the same value
    (a constant changed on one side only would mis-drive the kernels silently)."""
    from active_gym import _native as nat
    src = open(os.path.join(REPO, "include", "agx.h")).read()
    found = 0
    for name, val in re.findall(r"^#define\s+AGX_([A-Z0-9_]+)\s+(-?(?:0x[0-9A-Fa-f]+|\d+))\b", src, flags=re.M):
        if name in ("H", "CMD_NVALID_MASK"):
            continue
        assert hasattr(nat, name), f"active_gym/_native.py lacks {name}"
        assert getattr(nat, name) == int(val, 0), (name, getattr(nat, name), val)
        found += 1
    assert found >= 30

"""Host logic of the composed-operator kernels (k_fovea_flexible3, k_fovea_peripheral3): the tables agx_create builds
(active-gym_amd/csrc/agx_host_tables.h) are replayed on the CPU by tests/host_tables_harness.cpp - the kernels'
arithmetic, float32, same index rules - and compared with the reference's chain of torchvision Resize calls
(fov_env.py:276-298, :366-377) evaluated pass by pass in double.  No GPU involved: hipcc compiles the harness as a
plain host program."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    out = str(tmp_path_factory.mktemp("harness") / "host_tables_harness")
    subprocess.run([HIPCC, "-O1", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", "-I", os.path.join(REPO, "include"),
                    "-I", os.path.join(REPO, "active-gym_amd", "csrc"), os.path.join(REPO, "tests", "host_tables_harness.cpp"),
                    "-o", out], check=True, capture_output=True, timeout=300)
    return out


def _run(harness, *a):
    r = subprocess.run([harness, *map(str, a)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout.strip()


@pytest.mark.parametrize("geom_aa", [(84, 84, 30, 30, 0), (84, 84, 30, 30, 1), (64, 48, 20, 12, 0), (64, 48, 20, 12, 1),
                                     (36, 40, 7, 11, 0), (100, 100, 30, 20, 0), (48, 40, 20, 24, 1)])
def test_flexible3_tables_reproduce_the_resize_chain(harness, geom_aa):
    out = _run(harness, *geom_aa)
    assert out.startswith("max_err"), out                   # every one of these geometries is inside the kernel's plan
    assert float(out.split()[1]) <= 2e-6, out               # float32 evaluation of a double chain; the bar is 1e-5


@pytest.mark.parametrize("geom", [(84, 84, 20, 20), (84, 84, 84, 84), (36, 40, 7, 11), (96, 96, 50, 50)])
@pytest.mark.parametrize("aa", [0, 1])
def test_peripheral3_tables_reproduce_squeeze_expand(harness, geom, aa):
    out = _run(harness, "per", *geom, aa)                   # also checks unit_fast(k) == float32(k)/255 for all k
    assert out.startswith("max_err"), out
    assert float(out.split()[1]) <= 2e-6, out


@pytest.mark.parametrize("geom_aa", [(84, 84, 30, 30, 0), (84, 84, 30, 30, 1), (64, 48, 20, 12, 1), (36, 40, 7, 11, 0), (48, 40, 20, 24, 1)])
def test_flexible_raw3_tables_reproduce_the_squeeze_and_back_chain(harness, geom_aa):
    """k_fovea_flexible_raw3 (raw-crop / mask-out / packed forms): composed (Wbck Wdwn), Hdwn and Hbck tables against
    crop -> Resize(fov_size) -> Resize(fov_res) in double, incl. the LDS image's index rules."""
    out = _run(harness, "raw", *geom_aa)
    assert out.startswith("max_err"), out
    assert float(out.split()[1]) <= 2e-6, out


def test_geometries_outside_the_plan_are_reported(harness):
    assert _run(harness, 128, 128, 31, 9, 1) == "unsupported"          # > 16 composed taps: k_fovea_flexible2 runs
    assert _run(harness, "per", 84, 84, 5, 5, 1) == "unsupported"      # > 16 squeeze taps: k_fovea_peripheral2 runs

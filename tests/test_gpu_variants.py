"""The experimental kernel forms (active-gym_amd/csrc/experiments/: built, measured equal or slower)
are not in libagx.so.  This test loads the experiments build (libagx_exp.so = libagx.so + those forms + their knobs) in a
child process and holds every form against the default kernels bit for bit - u8 ring, fov_loc, float observations - and
the experiments build's default kernels against libagx.so itself (CRCs over the same seeded inputs)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exp_lib():
    sys.path.insert(0, os.path.join(REPO, "active-gym_amd"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("agx_build", os.path.join(REPO, "active-gym_amd", "build.py"))
    bld = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bld)
    return bld.build_experiments()          # up to date when __graft_entry__.build() has run; compiles otherwise


def test_experimental_forms_bit_identical_to_default_kernels():
    lib = _exp_lib()
    env = dict(os.environ, AGX_LIB=lib)
    for k in list(env):
        if k.startswith(("AGX_INGEST_", "AGX_STEP_", "AGX_FOVEA_PAIR")):
            env.pop(k)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "variants_child.py")], capture_output=True, text=True,
                       env=env, cwd=REPO, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("VARIANTS_JSON ")][-1]
    rep = json.loads(line[len("VARIANTS_JSON "):])
    bad = {k: v for k, v in rep["variants"].items() if v != "ok"}
    assert len(rep["variants"]) == 38 and not bad, bad          # 17 knob settings x 2 geometries + the packed wave form x 2 x 2
    # the shipped library's kernels produce exactly what the experiments build's default kernels produce
    import variants_child as vc
    from active_gym import _native as nat
    assert "libagx_exp" not in open("/proc/self/maps").read()
    dev = torch.device("cuda:0")
    for geom in ("headline", "generic"):
        assert vc.default_crcs(dev, geom) == rep["default_crcs"][geom], geom
    assert rep["build"].split(" src ")[-1] == nat.build_info().split(" src ")[-1]      # same kernel sources

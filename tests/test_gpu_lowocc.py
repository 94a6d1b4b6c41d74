"""The kernel-level test the inline-asm store bug of round 3 (commit 327a14a, fixed in c803dbc) walks into - and the proof
that it does.

What the bug was: `global_store_dwordx4 ... sc1` as inline asm is invisible to the compiler's hazard recogniser, and hipcc
scheduled a VALU write of the store's first data VGPR (`v_lshl_add_u32 v4, ..` for the next pass's LDS address) into the very
next issue slot; on gfx950 the data VGPRs of a store of more than 64 bits must not be overwritten within two wait states.  With
ONE wave per SIMD the overwrite lands every time (component 0 of the first two store passes of every wave: 512 of a frame's
7,056 values, deterministically); with eight waves per SIMD other waves' instructions usually separate the pair.  tools/
canary_probe.py (profiles/r04_canary_probe.txt) shows it on today's kernels with the old store and on the library built from
commit 327a14a itself: EVERY launch sequence is hit, at every batch size - so the kernel-level parity tests of round 3 would
have been red too; they were never run on that build (it was committed on timing runs, and `pytest -x` reaches
tests/test_gpu_env.py first).  Two consequences live here:

* `test_low_occupancy_launch_sequences_match_the_oracle`: every fovea kernel at N = 1 and 5 (one workgroup per CU at most - the
  state in which a hazard of this kind is certain to bite), in the two launch sequences of the env chain (reset: K1 with CLEAR,
  fovea_reset, fovea(None) on a ring of zeros and one frame; step: K1 then the fovea kernel with an action), against the oracle.
* `test_the_same_cases_are_red_on_the_known_bad_store`: the same child under lib/libagx_canary.so - today's sources with that
  commit's store (build.py --canary; -DAGX_CANARY_ASM_OBS_STORE) - must report wrong values for the kernels that issue the
  16-byte observation store.  A test suite that stays green on this library has lost its teeth.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(REPO, "active-gym_amd", "lib")


def _run(lib=None, only=None):
    env = dict(os.environ)
    env.pop("AGX_LIB", None)
    if lib:
        env["AGX_LIB"] = lib
    if only:
        env["LOWOCC_ONLY"] = only
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "lowocc_child.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return {d["case"]: d["bad"] for d in (json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{"))}


def test_low_occupancy_launch_sequences_match_the_oracle():
    res = _run()
    assert len(res) == 8 * 2 * 2, sorted(res)
    wrong = {k: v for k, v in res.items() if v}
    assert not wrong, wrong


def test_the_same_cases_are_red_on_the_known_bad_store():
    canary = os.path.join(LIBDIR, "libagx_canary.so")
    if not os.path.exists(canary):
        pytest.skip("lib/libagx_canary.so not built (python active-gym_amd/build.py --canary; __graft_entry__.build() builds it)")
    res = _run(canary, only="fixed-resize")
    assert len(res) == 4, sorted(res)
    # K2 (resize_to_full, the headline kernel): both launch sequences, both batch sizes must see the corruption
    assert all(v > 0 for v in res.values()), res

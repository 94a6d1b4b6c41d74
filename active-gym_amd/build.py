#!/usr/bin/env python3
"""Build libagx.so (the HIP extension behind include/agx.h) for gfx950, in-tree.

    python active-gym_amd/build.py            # -> active-gym_amd/lib/libagx.so

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with
gpurun snapshots.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "agx_api.hip")
DEPS = [SRC] + sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith(".h")) + \
       [os.path.join(REPO, "include", "agx.h"), os.path.join(REPO, "include", "agx_loop.h")]
# the measured dead ends (csrc/experiments/): compiled only into libagx_exp.so, for tools/ and tests/test_gpu_variants.py
EXP_DEPS = sorted(os.path.join(HERE, "csrc", "experiments", f) for f in os.listdir(os.path.join(HERE, "csrc", "experiments")))
OUT_DIR = os.path.join(HERE, "lib")
OUT = os.path.join(OUT_DIR, "libagx.so")
EXP_OUT = os.path.join(OUT_DIR, "libagx_exp.so")
CANARY_OUT = os.path.join(OUT_DIR, "libagx_canary.so")


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=...)")


def source_hash():
    """SHA-256 prefix over the sources libagx.so is compiled from (what agx_build_info() reports)."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        h.update(os.path.basename(d).encode() + b"\0")
        h.update(open(d, "rb").read())
    return h.hexdigest()[:12]


def up_to_date(out=OUT, deps=None):
    deps = DEPS if deps is None else deps
    return os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps)


def build(force=False, verbose=False, extra=(), out=OUT, deps=None):
    if not force and up_to_date(out, deps):
        return out
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
           "-I", os.path.join(REPO, "include"), "-I", os.path.join(HERE, "csrc"),
           "-DAGX_BUILD", f'-DAGX_SRC_HASH="{source_hash()}"', *extra, SRC, "-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    return out


def build_experiments(force=False, verbose=False):
    """libagx_exp.so = libagx.so + the kernel forms under csrc/experiments/ and their environment knobs (-DAGX_EXPERIMENTS).
    Never loaded by the product: tools/ and tests/test_gpu_variants.py select it through AGX_LIB."""
    return build(force, verbose, ("-DAGX_EXPERIMENTS",), EXP_OUT, DEPS + EXP_DEPS)


def build_canary(force=False, verbose=False):
    """libagx_canary.so = libagx.so with the KNOWN-BAD observation store of commit 327a14a (an inline-asm
    `global_store_dwordx4 ... sc1`, whose data VGPRs the next VALU instructions may overwrite: -DAGX_CANARY_ASM_OBS_STORE).
    Never loaded by the product: tests/test_gpu_lowocc.py runs its own cases against it in a child process and expects them to
    FAIL there - the proof that those cases see the bug the round-3 kernel tests missed."""
    return build(force, verbose, ("-DAGX_CANARY_ASM_OBS_STORE",), CANARY_OUT, DEPS)


RUNNER_SRC = os.path.join(HERE, "csrc", "agx_runner.cpp")
RUNNER_OUT = os.path.join(OUT_DIR, "libagx_runner.so")


def build_runner(force=False, verbose=False):
    """The native host runner (include/agx_runner.h): plain C++17 + pthreads, no GPU code.  -O3 -mavx2: the scripted
    emulator's screen loops vectorise (2.4x on RGB screens); every MI355X host CPU (EPYC Zen 4/5) has AVX2."""
    deps = [RUNNER_SRC, os.path.join(REPO, "include", "agx_runner.h")]
    if not force and os.path.exists(RUNNER_OUT) and all(os.path.getmtime(RUNNER_OUT) >= os.path.getmtime(d) for d in deps):
        return RUNNER_OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-O3", "-mavx2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-I", os.path.join(REPO, "include"),
           RUNNER_SRC, "-o", RUNNER_OUT + ".tmp", "-pthread", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(RUNNER_OUT + ".tmp", RUNNER_OUT)
    return RUNNER_OUT


if __name__ == "__main__":
    extra = [a for a in sys.argv[1:] if a.startswith("-") and a not in ("-f", "-v", "--experiments", "--canary")]
    print(build(force="-f" in sys.argv, verbose=True, extra=extra))
    print(build_runner(force="-f" in sys.argv, verbose=True))
    if "--experiments" in sys.argv:
        print(build_experiments(force="-f" in sys.argv, verbose=True))
    if "--canary" in sys.argv:
        print(build_canary(force="-f" in sys.argv, verbose=True))

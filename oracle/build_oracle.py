#!/usr/bin/env python3
"""Build the C restatement of the oracle (oracle/cport.c -> oracle/_build/liboracle.so).
Checker / CPU-baseline only; building it is not using it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cport.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "liboracle.so")


def build(force=False):
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", OUT + ".tmp", SRC, "-lm"], check=True)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True))

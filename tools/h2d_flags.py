#!/usr/bin/env python3
"""H2D rate of one step's screens (165 MB) from pinned host memory allocated with different hipHostMalloc flags: default (coherent), write-combined,
non-coherent, and torch's pin_memory for reference; plus the host-side cost of filling the buffer (one thread, streaming stores vs memset).
    python tools/h2d_flags.py"""
import ctypes as C
import time
hip = C.CDLL("libamdhip64.so")
NBYTES = 1024 * 2 * 168 * 160 * 3
FLAGS = {"default": 0x0, "portable": 0x1, "write-combined": 0x4, "coherent": 0x40000000, "non-coherent": 0x80000000, "numa-user": 0x20000000}
def chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc}")
dptr = C.c_void_p()
chk(hip.hipMalloc(C.byref(dptr), C.c_size_t(NBYTES)), "hipMalloc")
stream = C.c_void_p()
chk(hip.hipStreamCreate(C.byref(stream)), "hipStreamCreate")
libc = C.CDLL("libc.so.6")
for name, flag in FLAGS.items():
    h = C.c_void_p()
    rc = hip.hipHostMalloc(C.byref(h), C.c_size_t(NBYTES), C.c_uint(flag))
    if rc != 0:
        print(f"{name:15s} hipHostMalloc failed ({rc})")
        continue
    t0 = time.perf_counter()
    libc.memset(h, 1, C.c_size_t(NBYTES))
    fill = time.perf_counter() - t0
    t0 = time.perf_counter()
    libc.memset(h, 2, C.c_size_t(NBYTES))
    fill2 = time.perf_counter() - t0
    for _ in range(3):
        chk(hip.hipMemcpyAsync(dptr, h, C.c_size_t(NBYTES), 1, stream), "copy")
    chk(hip.hipStreamSynchronize(stream), "sync")
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(4):
            chk(hip.hipMemcpyAsync(dptr, h, C.c_size_t(NBYTES), 1, stream), "copy")
        chk(hip.hipStreamSynchronize(stream), "sync")
        best = min(best, (time.perf_counter() - t0) / 4)
    print(f"{name:15s} H2D {NBYTES / best / 1e9:6.2f} GB/s ({best * 1e3:.3f} ms per 165 MB)   host memset first touch {NBYTES / fill / 1e9:5.1f} GB/s, second {NBYTES / fill2 / 1e9:5.1f} GB/s", flush=True)
    hip.hipHostFree(h)

#!/usr/bin/env python3
"""psvr_bgzf_compress on BAM-like bytes: wall per call from pageable and from page-locked host memory, output ratio.
usage: python tools/bgzf_bench.py [MB]   (PSVR_BGZF_BLOCK=<bytes> changes the member size)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pansvr_amd._lib import check, lib  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 192
rng = np.random.RandomState(3)
# records like the main file's: fixed fields, a name, 75 bytes of 4-bit sequence, 150 quality values from a small alphabet, text tags
rec = []
size = 0
i = 0
while size < mb << 20:
    r = (np.uint32(400).tobytes() + rng.randint(0, 1 << 20, size=8).astype(np.uint32).tobytes() + b"read%08d\0" % i + rng.randint(0, 256, size=75, dtype=np.uint8).tobytes()
         + (rng.randint(0, 6, size=150) * 5 + 10).astype(np.uint8).tobytes() + b"ASC\x2aOSC\x20OAZ3,%d,0,60,M;\0RCZ3_%d_0_280_60_150M\0" % (rng.randint(1 << 27), rng.randint(1 << 27)))
    rec.append(r)
    size += len(r)
    i += 1
data = b"".join(rec)
L = lib()
L.psvr_bgzf_bound.restype = C.c_int64
L.psvr_bgzf_bound.argtypes = [C.c_int64]
L.psvr_host_alloc.restype = C.c_void_p
L.psvr_host_alloc.argtypes = [C.c_size_t]
cap = L.psvr_bgzf_bound(len(data))
got = C.c_int64(0)
src = np.frombuffer(data, dtype=np.uint8).copy()
dst = np.zeros(cap, dtype=np.uint8)
pin_in, pin_out = L.psvr_host_alloc(len(data)), L.psvr_host_alloc(cap)
C.memmove(pin_in, src.ctypes.data, len(data))
for name, a, b in (("pageable", src.ctypes.data, dst.ctypes.data), ("page-locked", pin_in, pin_out)):
    check(L.psvr_bgzf_compress(0, C.c_void_p(a), C.c_int64(len(data)), C.c_void_p(b), C.c_int64(cap), C.byref(got)))
    t0 = time.time()
    for _ in range(3):
        check(L.psvr_bgzf_compress(0, C.c_void_p(a), C.c_int64(len(data)), C.c_void_p(b), C.c_int64(cap), C.byref(got)))
    dt = (time.time() - t0) / 3
    print("%-12s %6.1f MB -> %6.1f MB (ratio %.2f) in %.1f ms = %.2f GB/s" % (name, len(data) / 1e6, got.value / 1e6, len(data) / got.value, dt * 1e3, len(data) / dt / 1e9), flush=True)

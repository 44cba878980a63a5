#!/usr/bin/env python3
"""K6w latency and throughput: one full time-out alone, and batches of full time-outs (noise vectors, 10 000 cycles per
bit = 810 000 cycles each).  fano_latency.py"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import rtlsdr_wsprd_amd as w
L = w.lib()
assert L.wspr_device_ready() == 1
L.wspr_fano_batch_device_wave.argtypes = [C.c_void_p, C.c_int, C.c_uint] + [C.c_void_p] * 6
rng = np.random.default_rng(99)
for n in (1, 64, 1000, 3328, 6000):
    sym = np.clip(rng.normal(128, 45, (n, 162)), 0, 255).astype(np.uint8)
    ret = np.zeros(n, np.int32); cyc = np.zeros(n, np.uint32); met = np.zeros(n, np.uint32); mnp = np.zeros(n, np.uint32)
    dat = np.zeros((n, 10), np.uint8); steps = np.zeros(n, np.uint32)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        assert L.wspr_fano_batch_device_wave(sym.ctypes.data, n, 10000, ret.ctypes.data, cyc.ctypes.data, met.ctypes.data,
                                             mnp.ctypes.data, dat.ctypes.data, steps.ctypes.data) == 0
        best = min(best, time.perf_counter() - t0)
    print("%5d vectors: %7.2f ms (host wall incl. copies), %d time-outs, steps mean %.0f max %d, overflow %d" % (
        n, best * 1e3, int((ret == -1).sum()), steps.mean(), steps.max(), int((ret == -2).sum())))

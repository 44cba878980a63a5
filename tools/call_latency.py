#!/usr/bin/env python3
"""One wspr_decode() call on the reference's signal file, warm, median of 50 (the configs[0] figure)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import rtlsdr_wsprd_amd as w
import ctypes as C
L = w.lib()
I = np.zeros(45000, np.float32); Q = np.zeros(45000, np.float32)
n = L.wspr_read_iq_file(os.path.join(ROOT, "tests", "golden", "refSignalSnr0dB.iq").encode(), I.ctypes.data_as(C.c_void_p), Q.ctypes.data_as(C.c_void_p))
for _ in range(5):
    w.wspr_decode(I, Q, n)
ts = []
c0 = time.process_time()
for _ in range(50):
    t0 = time.perf_counter(); w.wspr_decode(I, Q, n); ts.append(time.perf_counter() - t0)
cpu = time.process_time() - c0
ts.sort()
print("wspr_decode: median %.3f ms, min %.3f, max %.3f; CPU %.3f ms per call; WSPR_BLOCKING_SYNC=%s" % (
    1e3 * ts[25], 1e3 * ts[0], 1e3 * ts[-1], 1e3 * cpu / 50, os.environ.get("WSPR_BLOCKING_SYNC", "unset")))

#!/usr/bin/env python3
"""Latency of the drop-in entry point wspr_decode() on one segment (the daemon's use), next to the CPU oracle."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rtlsdr_wsprd_amd as w
import oracle_lib as ol
I, Q, n = ol.read_iq_file(os.path.join(ol.GOLDEN, "refSignalSnr0dB.iq"))
for _ in range(3): w.wspr_decode(I, Q, 45000, w.default_options())
t0 = time.perf_counter()
for _ in range(20): spots, _, _ = w.wspr_decode(I, Q, 45000, w.default_options())
print("wspr_decode (1 segment, host buffers): %.2f ms per call, %d spot(s)" % ((time.perf_counter() - t0) / 20 * 1e3, len(spots)))
L = ol.lib()
t0 = time.perf_counter()
for _ in range(5): ol.decode(I, Q, 45000)
print("oracle (1 CPU core): %.1f ms per call" % ((time.perf_counter() - t0) / 5 * 1e3))

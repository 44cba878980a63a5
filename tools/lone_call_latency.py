#!/usr/bin/env python3
"""Latency of ONE call in flight by batch size (resident input, configs[1]- and configs[2]-type segments), median of 9:
what a caller with a single batch sees, and what the waiting policy (WSPR_BLOCKING_SYNC) costs it."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import rtlsdr_wsprd_amd as w  # noqa: E402

NS = 45000
dev = torch.device("cuda", 0)
L = w.lib()
K = 32
opt = w.default_options()
for nsig, label in ((1, "1 signal"), (10, "10 signals")):
    big = 1024
    if nsig == 1:
        I, Q, _ = bench.synth_batch_gpu(big, 1234, dev, 1, -20.0, -20.0, 1.0)
    else:
        I, Q, _ = bench.synth_batch_gpu(big, 4321, dev, 10, -10.0, -28.0, 0.3)
    torch.cuda.synchronize()
    out = (w.decoder_results * (big * K))()
    n = (C.c_int * big)()
    row = []
    for nseg in (1, 4, 16, 17, 32, 64, 127, 128, 256, 1024):
        ts = []
        for rep in range(11):
            t0 = time.perf_counter()
            assert L.wspr_decode_batch_device(I.data_ptr(), Q.data_ptr(), nseg, NS, I.stride(0), opt, C.addressof(out), K, C.addressof(n)) == 0
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[2:])
        row.append("%d: %.2f" % (nseg, 1e3 * ts[len(ts) // 2]))
    print("%s, WSPR_BLOCKING_SYNC=%s, ms per lone call by segments:  %s" % (label, os.environ.get("WSPR_BLOCKING_SYNC", "unset"), "  ".join(row)), flush=True)

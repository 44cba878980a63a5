#!/usr/bin/env python3
"""FFT+sync stage (K1, K2, K3) kernel times vs resident batch size (Infinity-Cache effect)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtlsdr_wsprd_amd as w
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
L = w.lib()
I, Q, _ = bench.synth_batch_gpu(2048, 7, dev, 1, -20.0, -20.0, 1.0)
torch.cuda.synchronize()
for nseg in (128, 256, 512, 1024, 2048):
    ms = (C.c_double * 8)()
    L.wspr_bench_fft_sync(I.data_ptr(), Q.data_ptr(), nseg, 45000, I.stride(0), 20, C.addressof(ms))
    tot = ms[0] + ms[1] + ms[2]
    print("nseg %5d  k1 %.1f us (%d launches)  k2 %.1f us  k3 %.1f us  wall %.1f us | per-seg ns %.1f  stage GB/s %.0f (wall %.0f)  K1 GB/s %.0f" % (
        nseg, ms[0] * 1e3, int(ms[3]), ms[1] * 1e3, ms[2] * 1e3, ms[4] * 1e3, tot * 1e6 / nseg,
        bench.STAGE_BYTES * nseg / tot / 1e6, bench.STAGE_BYTES * nseg / ms[4] / 1e6, bench.K1_BYTES * nseg / ms[0] / 1e6))

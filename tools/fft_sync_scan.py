#!/usr/bin/env python3
"""FFT+sync stage (K1, K2, K3) kernel times vs resident batch size and signals per segment.
usage: fft_sync_scan.py [signals_per_segment] [nseg ...]"""
import ctypes as C, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtlsdr_wsprd_amd as w
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
L = w.lab()          # timing / calibration entry points: the lab library (include/wspr_mi355x_bench.h)
nsig = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sizes = [int(x) for x in sys.argv[2:]] or [128, 256, 512, 1024, 2048]
if nsig == 1:
    I, Q, _ = bench.synth_batch_gpu(max(sizes), 7, dev, 1, -20.0, -20.0, 1.0)
else:
    I, Q, _ = bench.synth_batch_gpu(max(sizes), 7, dev, nsig, -10.0, -28.0, 0.3)
torch.cuda.synchronize()
n_copy = 1 << 28
src = torch.empty(n_copy, device=dev, dtype=torch.float32).normal_(); dst = torch.empty_like(src)
torch.cuda.synchronize()
L.wspr_calib_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(n_copy), 2)
t0 = time.perf_counter()
L.wspr_calib_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(n_copy), 10)
print("4-byte-per-lane copy: %.0f GB/s   (%s)" % (10 * 8.0 * n_copy / (time.perf_counter() - t0) / 1e9,
      " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("WSPR_"))))
del src, dst
for nseg in sizes:
    ms = (C.c_double * 8)()
    L.wspr_bench_fft_sync(I.data_ptr(), Q.data_ptr(), nseg, 45000, I.stride(0), 10, C.addressof(ms))
    tot = ms[0] + ms[1] + ms[2]
    print("nseg %5d  k1 %.1f us (%d launches)  k2 %.1f us  k3 %.1f us  wall %.1f us | per-seg ns %.1f  stage GB/s %.0f (wall %.0f)  K1 GB/s %.0f" % (
        nseg, ms[0] * 1e3, int(ms[3]), ms[1] * 1e3, ms[2] * 1e3, ms[4] * 1e3, tot * 1e6 / nseg,
        bench.STAGE_BYTES * nseg / tot / 1e6, bench.STAGE_BYTES * nseg / ms[4] / 1e6, bench.K1_BYTES * nseg / ms[0] / 1e6))
# candidates per segment (what K3 works on)
import numpy as np
ns = min(256, I.shape[0])
Ih = I[:ns].cpu().numpy(); Qh = Q[:ns].cpu().numpy()
cands = (w.cand * (200 * ns))(); npk = (C.c_int * ns)()
L.wspr_stage_candidates(Ih.ctypes.data_as(C.c_void_p), Qh.ctypes.data_as(C.c_void_p), ns, 45000, 45000, 1, 4, C.addressof(cands),
                        C.addressof(npk), None, None)
print("candidates per segment: mean %.1f max %d" % (np.mean(list(npk)), max(npk)))

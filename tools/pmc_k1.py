#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE):
a calibration stream copy of known size, then the FFT+sync stage (K1, K2, K3) on resident segments.
usage: pmc_k1.py [segments] [signals per segment].  tools/pmc_summarise.py turns the two counter CSVs
into profiles/*.json."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtlsdr_wsprd_amd as w
import bench
nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nsig = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L = w.lab()          # timing / calibration entry points: the lab library (include/wspr_mi355x_bench.h)
L.wspr_calib_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
n = 1 << 28                                    # 1 GiB of floats: far beyond the 256 MiB Infinity Cache
src = torch.rand(n, device=dev); dst = torch.empty_like(src)
if nsig == 1:
    I, Q, _ = bench.synth_batch_gpu(nseg, 99, dev, 1, -20.0, -20.0, 1.0)
else:
    I, Q, _ = bench.synth_batch_gpu(nseg, 99, dev, nsig, -10.0, -28.0, 0.3)
torch.cuda.synchronize()
L.wspr_calib_copy(src.data_ptr(), dst.data_ptr(), n, 3)
ms = (C.c_double * 8)()
L.wspr_bench_fft_sync(I.data_ptr(), Q.data_ptr(), nseg, 45000, I.stride(0), 5, C.addressof(ms))
print("k1 k2 k3 ms:", list(ms))

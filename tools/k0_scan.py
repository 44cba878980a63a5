#!/usr/bin/env python3
"""K0 alone: average launch time of the front end (20 launches, HIP events) and of the read-only calibration
kernel over the same resident rows.  k0_scan.py [segments]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtlsdr_wsprd_amd as w
nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 32
RAW = 576_000_000
dev = torch.device("cuda", 0)
raw = torch.randint(1, 256, (nseg, RAW), device=dev, dtype=torch.uint8)      # no 0x00: the dot-product path
L = w.lab()          # timing / calibration entry points: the lab library (include/wspr_mi355x_bench.h)
stride = int(L.wspr_iq_stride())
I = torch.zeros(nseg, stride, device=dev); Q = torch.zeros_like(I)
ms = (C.c_double * 1)()
torch.cuda.synchronize()
for rep in range(3):
    L.wspr_bench_decimate(raw.data_ptr(), RAW, nseg, I.data_ptr(), Q.data_ptr(), 20, C.addressof(ms))
    k0 = ms[0]
    L.wspr_calib_read(raw.data_ptr(), RAW, nseg, 20, C.addressof(ms))
    print("K0 %.3f ms = %.0f GB/s (%.3f of 8 TB/s)   read-only %.3f ms = %.0f GB/s" %
          (k0, RAW * nseg / k0 / 1e6, RAW * nseg / k0 / 1e6 / 8000, ms[0], RAW * nseg / ms[0] / 1e6))

#!/usr/bin/env python3
"""The three fp32 launch sets (lag scan, frequency scan + first rung, subtraction) of wspr_bench_valu on resident
synthetic segments, for use under rocprofv3 (--kernel-trace / --pmc).  valu_probe.py [segments] [signals] [iters]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import rtlsdr_wsprd_amd as w
import bench
nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nsig = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
if nsig == 1:
    I, Q, _ = bench.synth_batch_gpu(nseg, 99, dev, 1, -20.0, -20.0, 1.0)
else:
    I, Q, _ = bench.synth_batch_gpu(nseg, 99, dev, nsig, -10.0, -28.0, 0.3)
torch.cuda.synchronize()
L = w.lab()          # timing / calibration entry points: the lab library (include/wspr_mi355x_bench.h)
L.wspr_bench_valu.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p]
ms = (C.c_double * 8)()
L.wspr_bench_valu(I.data_ptr(), Q.data_ptr(), nseg, 45000, I.stride(0), iters, C.addressof(ms))
print("lag scan %.3f ms, frequency scan + rung 0 %.3f ms, subtraction %.3f ms; %d candidates, %d jobs" %
      (ms[0], ms[4], ms[1], int(ms[2]), int(ms[3])))

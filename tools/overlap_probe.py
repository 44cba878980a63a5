#!/usr/bin/env python3
"""Do the HBM-bound front end (K0) and the fp32-VALU-bound decoder stages overlap on this GPU?  Times K0 alone,
the lag scan + frequency scan + subtraction launch set (wspr_bench_valu) alone, and both at once from two host
threads on two lanes (own HIP streams)."""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import rtlsdr_wsprd_amd as w
import bench
dev = torch.device("cuda", 0)
L = w.lab()          # timing / calibration entry points: the lab library (include/wspr_mi355x_bench.h)
nraw, RAW = 32, 576_000_000
raw = torch.randint(1, 256, (nraw, RAW), device=dev, dtype=torch.uint8)
stride = int(L.wspr_iq_stride())
rI = torch.zeros(nraw, stride, device=dev); rQ = torch.zeros_like(rI)
I, Q, _ = bench.synth_batch_gpu(2048, 99, dev, 1, -20.0, -20.0, 1.0)
L.wspr_bench_valu.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p]
torch.cuda.synchronize()

def k0(iters, out):
    L.wspr_bind_thread_lane(0)
    ms = (C.c_double * 1)()
    t = time.perf_counter()
    L.wspr_bench_decimate(raw.data_ptr(), RAW, nraw, rI.data_ptr(), rQ.data_ptr(), iters, C.addressof(ms))
    out["k0"] = (time.perf_counter() - t) / iters * 1e3

def valu(iters, out):
    L.wspr_bind_thread_lane(1)
    ms = (C.c_double * 8)()
    t = time.perf_counter()
    L.wspr_bench_valu(I.data_ptr(), Q.data_ptr(), 2048, 45000, I.stride(0), iters, C.addressof(ms))
    out["valu"] = (time.perf_counter() - t) / iters * 1e3

for f, n in ((k0, 5), (valu, 3)):            # settle
    o = {}; th = threading.Thread(target=f, args=(n, o)); th.start(); th.join()
a = {}; th = threading.Thread(target=k0, args=(40, a)); th.start(); th.join()
b = {}; th = threading.Thread(target=valu, args=(20, b)); th.start(); th.join()
print("alone   : K0 %.3f ms per launch (32 segments), VALU set %.3f ms per launch set (2048 candidates)" % (a["k0"], b["valu"]))
n0 = max(4, int(round(20 * b["valu"] / a["k0"])))         # equal wall time alone
c = {}
t0 = time.perf_counter()
t1 = threading.Thread(target=k0, args=(n0, c)); t2 = threading.Thread(target=valu, args=(20, c))
t1.start(); t2.start(); t1.join(); t2.join()
wall = (time.perf_counter() - t0) * 1e3
print("together: K0 %.3f ms, VALU set %.3f ms; wall %.1f ms for work that takes %.1f ms back to back" %
      (c["k0"], c["valu"], wall, n0 * a["k0"] + 20 * b["valu"]))

#!/usr/bin/env python3
"""configs[4]: why do the front end (K0, HBM-bound) and the decoder (vector-issue-bound) ADD instead of overlapping?
The decisive pairs (verdict of round 4): each kernel alone, then both at once from two host threads on two lanes:
  K0          || decoder launch set     the case itself
  plain READ  || decoder launch set     a kernel with K0's access pattern and NO arithmetic: what remains is the load path
  K0          || register-only VALU     a kernel with NO memory traffic: what remains is issue sharing
wall time together / (sum of the times alone) = 1.0: they add; = max/sum: perfect overlap."""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import rtlsdr_wsprd_amd as w
import bench
dev = torch.device("cuda", 0)
L = w.lab()
nraw, RAW = 32, 576_000_000
raw = torch.randint(1, 256, (nraw, RAW), device=dev, dtype=torch.uint8)
stride = int(L.wspr_iq_stride())
rI = torch.zeros(nraw, stride, device=dev); rQ = torch.zeros_like(rI)
I, Q, _ = bench.synth_batch_gpu(2048, 99, dev, 1, -20.0, -20.0, 1.0)
torch.cuda.synchronize()


def k0(iters):
    ms = (C.c_double * 1)()
    L.wspr_bench_decimate(raw.data_ptr(), RAW, nraw, rI.data_ptr(), rQ.data_ptr(), iters, C.addressof(ms))


def read(iters):
    ms = (C.c_double * 1)()
    L.wspr_calib_read(raw.data_ptr(), RAW, nraw, iters, C.addressof(ms))


def valu(iters):
    ms = (C.c_double * 8)()
    L.wspr_bench_valu(I.data_ptr(), Q.data_ptr(), 2048, 45000, I.stride(0), iters, C.addressof(ms))


def regs(iters):
    tf = C.c_double(0.0)
    L.wspr_calib_valu(iters, C.addressof(tf))


def run(fn, iters, lane, out, key):
    L.wspr_bind_thread_lane(lane)
    t = time.perf_counter()
    fn(iters)
    out[key] = (time.perf_counter() - t) * 1e3


def alone(fn, iters, lane):
    o = {}
    th = threading.Thread(target=run, args=(fn, iters, lane, o, "t")); th.start(); th.join()
    return o["t"] / iters


# shader clock and socket power, sampled three times a second and tagged with what is running: is the chip at its power
# limit, so that two kernels together get a lower clock each -- i.e. ENERGY, not a pipe, is what they share?
import re
import subprocess
phase = ["idle"]
samples = {}
stop_sampling = threading.Event()


def sampler():
    while not stop_sampling.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.search(r"sclk clock level[^(]*\((\d+)Mhz\)", out)
            pw = re.search(r"Package Power[^:]*:\s*([0-9.]+)", out)
            if sclk and pw:
                samples.setdefault(phase[0], []).append((int(sclk.group(1)), float(pw.group(1))))
        except Exception:
            pass
        time.sleep(0.3)


threading.Thread(target=sampler, daemon=True).start()


def report(name):
    v = samples.get(name, [])
    if len(v) > 2:
        v = v[1:]                                           # the first sample straddles the phase's start
    if v:
        print("    %-28s sclk %4.0f MHz (min %d), socket power %4.0f W over %d samples" %
              (name, sum(a for a, _ in v) / len(v), min(a for a, _ in v), sum(b for _, b in v) / len(v), len(v)))


for fn, n, lane in ((k0, 5, 0), (read, 5, 0), (valu, 3, 1), (regs, 5, 1)):
    alone(fn, n, lane)
for name, fn, n, lane in (("K0 alone", k0, 700, 0), ("read alone", read, 800, 0), ("decoder set alone", valu, 400, 1),
                          ("register VALU alone", regs, 550, 1)):
    phase[0] = name
    alone(fn, n, lane)
    phase[0] = "idle"
    report(name)
t = {"K0": alone(k0, 40, 0), "read": alone(read, 40, 0), "decoder set": alone(valu, 20, 1), "register VALU": alone(regs, 40, 1)}
print("alone, ms per launch (set): " + ", ".join("%s %.3f" % kv for kv in t.items()))
fns = {"K0": k0, "read": read, "decoder set": valu, "register VALU": regs}
for a, b in (("K0", "decoder set"), ("read", "decoder set"), ("K0", "register VALU"), ("read", "register VALU")):
    target = 2000.0                                          # ms of work each, alone (long enough for the power samples)
    phase[0] = "%s || %s" % (a, b)
    na, nb = max(4, int(round(target / t[a]))), max(4, int(round(target / t[b])))
    o = {}
    t0 = time.perf_counter()
    ta = threading.Thread(target=run, args=(fns[a], na, 0, o, "a")); tb = threading.Thread(target=run, args=(fns[b], nb, 1, o, "b"))
    ta.start(); tb.start(); ta.join(); tb.join()
    wall = (time.perf_counter() - t0) * 1e3
    wa, wb = na * t[a], nb * t[b]
    print("%-6s || %-14s: together %.1f ms; alone %.1f + %.1f = %.1f back to back, %.1f if they overlapped perfectly -> %.2f of the sum"
          % (a, b, wall, wa, wb, wa + wb, max(wa, wb), wall / (wa + wb)))
    name = phase[0]
    phase[0] = "idle"
    report(name)

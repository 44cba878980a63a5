#!/usr/bin/env python3
"""Throughput with two batches in flight (two host threads bound to two lanes) vs one."""
import os, sys, time, threading
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
import rtlsdr_wsprd_amd as w
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
nseg = 1024
I, Q, exp = bench.synth_batch_gpu(nseg, 1234, dev, 1, -20.0, -20.0, 1.0)
L = w.lib()
def bind(lane):
    torch.cuda.set_device(0)
    return L.wspr_bind_thread_lane(lane)
for depth in (1, 2, 3):
    ex = [ThreadPoolExecutor(1) for _ in range(depth)]
    for k, e in enumerate(ex): assert e.submit(bind, k).result() == k
    decs = [w.BatchDecoder(nseg, 16) for _ in range(depth)]
    def run(k): decs[k].decode(I, Q); return decs[k].total_spots()
    for k in range(depth):
        for _ in range(3): ex[k].submit(run, k).result()
    K = 60
    t0 = time.perf_counter()
    futs = []
    for s in range(K):
        k = s % depth
        if len(futs) >= depth: futs.pop(0).result()
        futs.append(ex[k].submit(run, k))
    spots = [f.result() for f in futs]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("batches in flight %d: %.0f segments/s, %.2f ms per batch, spots %s" % (depth, K * nseg / dt, dt / K * 1e3, spots[-1]))
    for e in ex: e.shutdown()

#!/usr/bin/env python3
"""Completion time of every batch with two lanes in flight (looking for start-up effects)."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
import rtlsdr_wsprd_amd as w
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
nseg = 1024
I, Q, exp = bench.synth_batch_gpu(nseg, 1234, dev, 1, -20.0, -20.0, 1.0)
L = w.lib()
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ex = [ThreadPoolExecutor(1) for _ in range(depth)]
def bind(lane):
    torch.cuda.set_device(0); return L.wspr_bind_thread_lane(lane)
for k, e in enumerate(ex): e.submit(bind, k).result()
decs = [w.BatchDecoder(nseg, 16) for _ in range(depth)]
def run(k):
    t0 = time.perf_counter(); decs[k].decode(I, Q); return k, t0, time.perf_counter()
pending, log = [], []
T0 = time.perf_counter()
for s in range(40):
    if len(pending) >= depth: log.append(pending.pop(0).result())
    pending.append(ex[s % depth].submit(run, s % depth))
for f in pending: log.append(f.result())
prev = T0
for i, (k, a, b) in enumerate(log):
    print("step %2d lane %d start %7.2f end %7.2f dur %6.2f  since prev end %6.2f" % (i, k, (a - T0) * 1e3, (b - T0) * 1e3, (b - a) * 1e3, (b - prev) * 1e3))
    prev = b

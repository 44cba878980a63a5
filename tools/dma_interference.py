#!/usr/bin/env python3
"""Does host-to-device traffic slow the decoder down?  configs[2] resident, twelve lanes, while a background thread
keeps the link busy in different ways.  usage: dma_interference.py [steps]"""
import ctypes as C
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import rtlsdr_wsprd_amd as w  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
nseg, inflight, K = 8192, 12, 32
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L = w.lib()
I, Q, _ = bench.synth_batch_gpu(nseg, 4321, dev, 10, -10.0, -28.0, 0.3)
torch.cuda.synchronize()
opt = w.default_options()
lanes = [ThreadPoolExecutor(1) for _ in range(inflight)]
for k, ex in enumerate(lanes):
    ex.submit(lambda k=k: (torch.cuda.set_device(0), L.wspr_set_thread_slots(1), L.wspr_bind_thread_lane(k))).result()
outs = [((w.decoder_results * (nseg * K))(), (C.c_int * nseg)()) for _ in range(inflight)]


def call(k):
    o, n = outs[k]
    assert L.wspr_decode_batch_device(I.data_ptr(), Q.data_ptr(), nseg, 45000, I.stride(0), opt, C.addressof(o), K, C.addressof(n)) == 0


def run(n):
    pend = []
    for s in range(n):
        if len(pend) >= inflight:
            pend.pop(0).result()
        pend.append(lanes[s % inflight].submit(call, s % inflight))
    for f in pend:
        f.result()


def timed(label, background=None):
    stop = threading.Event()
    moved = [0]
    th = None
    if background:
        th = threading.Thread(target=background, args=(stop, moved))
        th.start()
    run(inflight)
    m0, t0 = moved[0], time.perf_counter()
    run(steps)
    el = time.perf_counter() - t0
    gbs = (moved[0] - m0) / el / 1e9
    stop.set()
    if th:
        th.join()
    print("%-46s %.1f ms/step  %.0f segments/s   background %.1f GB/s" % (label, 1e3 * el / steps, nseg * steps / el, gbs), flush=True)


src = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
dst = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()


def bg_copy(piece, pace_gbs=None, d2h=False):
    def f(stop, moved):
        torch.cuda.set_device(0)
        n = (1 << 30) // piece
        with torch.cuda.stream(side):
            while not stop.is_set():
                t0 = time.perf_counter()
                for i in range(n):
                    a, b = (src, dst) if d2h else (dst, src)
                    a[i * piece:(i + 1) * piece].copy_(b[i * piece:(i + 1) * piece], non_blocking=True)
                ev = torch.cuda.Event(); ev.record(side)
                while not ev.query():
                    time.sleep(0.0005)
                moved[0] += 1 << 30
                if pace_gbs:
                    rest = (1 << 30) / (pace_gbs * 1e9) - (time.perf_counter() - t0)
                    if rest > 0:
                        time.sleep(rest)
    return f


def bg_multi(nstreams, priority=0):
    """nstreams copy streams at once, each moving its own quarter of the buffers in 64 MiB pieces"""
    def f(stop, moved):
        torch.cuda.set_device(0)
        streams = [torch.cuda.Stream(priority=priority) for _ in range(nstreams)]
        part = (1 << 30) // nstreams
        piece = 1 << 26
        while not stop.is_set():
            evs = []
            for k, st in enumerate(streams):
                with torch.cuda.stream(st):
                    for off in range(k * part, (k + 1) * part, piece):
                        dst[off:off + piece].copy_(src[off:off + piece], non_blocking=True)
                    ev = torch.cuda.Event(); ev.record(st); evs.append(ev)
            for ev in evs:
                while not ev.query():
                    time.sleep(0.0005)
            moved[0] += 1 << 30
    return f


mode = sys.argv[2] if len(sys.argv) > 2 else "all"
if mode == "streams":
    timed("resident alone")
    timed("+ H2D, 1 stream, 64 MiB pieces", bg_multi(1))
    timed("+ H2D, 4 streams", bg_multi(4))
    timed("+ H2D, 8 streams", bg_multi(8))
    timed("+ H2D, 1 high-priority stream", bg_multi(1, -1))
    timed("+ H2D, 4 high-priority streams", bg_multi(4, -1))
    sys.exit(0)
timed("resident alone")
timed("+ H2D 1 GiB copies, back to back", bg_copy(1 << 30))
timed("+ H2D 16 MiB pieces, back to back", bg_copy(1 << 24))
timed("+ H2D 1 GiB copies paced to 14 GB/s", bg_copy(1 << 30, 14.0))
timed("+ H2D 16 MiB pieces paced to 14 GB/s", bg_copy(1 << 24, 14.0))
timed("+ D2H 1 GiB copies, back to back", bg_copy(1 << 30, None, True))
timed("resident alone again")

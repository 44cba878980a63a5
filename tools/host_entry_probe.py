#!/usr/bin/env python3
"""Host-buffer entry point at speed (the reference's own calling convention, wsprd.h:106-111 / rtlsdr_wsprd.c:316):
wspr_decode_batch() with caller memory pageable or pinned, `inflight` calls on as many lanes, against the same
batch resident in HBM (wspr_decode_batch_device), plus the plain H2D rate of the box (pinned and pageable) that bounds it.
usage: host_entry_probe.py [config 2|3] [nseg] [inflight] [steps]"""
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import rtlsdr_wsprd_amd as w  # noqa: E402

NS = 45000


def h2d_rate(pinned, nbytes=1 << 30, reps=5):
    src = torch.empty(nbytes, dtype=torch.uint8, pin_memory=pinned)
    src.fill_(1)
    dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    return reps * nbytes / (time.perf_counter() - t0) / 1e9


def main():
    config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    nseg = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if config == 2 else 8192)
    inflight = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else (120 if config == 2 else 24)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = w.lib()
    assert L.wspr_device_ready() == 1
    out = {"config": config, "nseg": nseg, "inflight": inflight, "steps": steps,
           "h2d_pinned_GBs": h2d_rate(True), "h2d_pageable_GBs": h2d_rate(False)}
    if config == 2:
        I, Q, exp = bench.synth_batch_gpu(nseg, 1234, dev, 1, -20.0, -20.0, 1.0)
    else:
        I, Q, exp = bench.synth_batch_gpu(nseg, 4321, dev, 10, -10.0, -28.0, 0.3)
    torch.cuda.synchronize()
    opt = w.default_options()
    lanes = [ThreadPoolExecutor(1) for _ in range(inflight)]

    def bind(k):
        torch.cuda.set_device(0)
        L.wspr_set_thread_slots(1 if inflight >= 4 else 0)
        return L.wspr_bind_thread_lane(k)
    for k, ex in enumerate(lanes):
        assert ex.submit(bind, k).result() == k
    K = 16 if config == 2 else 32
    outs = [((w.decoder_results * (nseg * K))(), (C.c_int * nseg)()) for _ in range(inflight)]

    def run(fn, n):
        pend = []
        for s in range(n):
            if len(pend) >= inflight:
                pend.pop(0).result()
            pend.append(lanes[s % inflight].submit(fn, s % inflight))
        for f in pend:
            f.result()

    def timed(fn, label):
        run(fn, 2 * inflight)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(fn, steps)
        el = time.perf_counter() - t0
        out[label] = {"segments_per_s": nseg * steps / el, "ms_per_step": 1e3 * el / steps,
                      "GBs_of_iq": 2 * 4 * NS * nseg * steps / el / 1e9}
        print(label, out[label], flush=True)

    def dev_call(k):
        o, n = outs[k]
        assert L.wspr_decode_batch_device(I.data_ptr(), Q.data_ptr(), nseg, NS, I.stride(0), opt, C.addressof(o), K, C.addressof(n)) == 0
    timed(dev_call, "resident")
    ref_counts = [list(outs[k][1]) for k in range(1)][0]
    # caller memory: pageable numpy rows, then pinned (torch pinned tensors: hipHostMalloc'ed)
    Ih, Qh = I.cpu().numpy(), Q.cpu().numpy()

    def host_call_of(Ia, Qa):
        pi, pq = Ia.ctypes.data_as(C.c_void_p), Qa.ctypes.data_as(C.c_void_p)

        def f(k):
            o, n = outs[k]
            assert L.wspr_decode_batch(pi, pq, nseg, NS, NS, opt, C.addressof(o), K, C.addressof(n), 0) == 0
        return f
    timed(host_call_of(Ih, Qh), "host_pageable")
    assert list(outs[0][1]) == ref_counts
    Ip = torch.empty(nseg, NS, dtype=torch.float32, pin_memory=True)
    Qp = torch.empty(nseg, NS, dtype=torch.float32, pin_memory=True)
    Ip.copy_(torch.from_numpy(Ih)); Qp.copy_(torch.from_numpy(Qh))
    timed(host_call_of(Ip.numpy(), Qp.numpy()), "host_pinned")
    assert list(outs[0][1]) == ref_counts
    out["pcie_bound_segments_per_s"] = out["h2d_pinned_GBs"] * 1e9 / (2 * 4 * NS)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Coarse-sync value of pass-0 candidates vs whether a spot was decoded at that frequency
(input to the speculative-window policy of the scheduler)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtlsdr_wsprd_amd as w
import oracle_lib as ol
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
NS = 45000
for name, args in (("config2", (256, 5, dev, 1, -20.0, -20.0, 1.0)), ("config3", (128, 6, dev, 10, -10.0, -28.0, 0.3))):
    I, Q, exp = bench.synth_batch_gpu(*args)
    nseg = args[0]
    Ih = np.ascontiguousarray(I.cpu().numpy()); Qh = np.ascontiguousarray(Q.cpu().numpy())
    cands = (ol.Cand * (200 * nseg))(); npk = (C.c_int * nseg)()
    assert w.lib().wspr_stage_candidates(ol.ptr(Ih), ol.ptr(Qh), nseg, NS, NS, 1, 4, C.addressof(cands), npk, None, None) == 0
    spots = w.wspr_decode_batch(Ih, Qh, w.default_options())
    dec_sync, non_sync = [], []
    for s in range(nseg):
        fr = [(x.freq - w.default_options().freq / 1e6) * 1e6 - 1500.0 for x in spots[s]]
        for j in range(npk[s]):
            cd = cands[s * 200 + j]
            hit = any(abs(cd.freq - f) < 2.5 for f in fr)
            (dec_sync if hit else non_sync).append(cd.sync)
    d = np.array(dec_sync); n = np.array(non_sync)
    print(name, "candidates/segment %.1f; decoded-at-freq %d, others %d" % (sum(npk) / nseg, d.size, n.size))
    for th in (0.10, 0.12, 0.15, 0.18, 0.20, 0.25, 0.30):
        print("  theta %.2f: decoded below %.3f, others at/above %.3f" % (th, (d < th).mean() if d.size else 0, (n >= th).mean() if n.size else 0))

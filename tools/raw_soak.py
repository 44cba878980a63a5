#!/usr/bin/env python3
"""Soak of the raw path: n full-size 2-minute raw segments (576 000 000 bytes each, varying signal and noise
levels) through K0 and the decoder on the GPU against the oracle front end + oracle decoder on the CPU:
decimated IQ bit for bit, every spot field.  raw_soak.py [segments] [first_seed]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, oracle_lib as ol
import rtlsdr_wsprd_amd as w
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
NS, RAW = 45000, bench.RAW_BYTES
rng = np.random.default_rng(seed0)
raws = []
for i in range(n):
    snr = float(rng.uniform(-26, -8)); noise = float(rng.choice([6.0, 10.0, 20.0, 60.0]))
    raws.append(bench.synth_raw_gpu(1, seed0 + i, dev, snr, noise_lsb=noise)[0])
raw = torch.cat(raws)
stride = int(w.lib().wspr_iq_stride())
dI = torch.zeros(n, stride, device=dev); dQ = torch.zeros(n, stride, device=dev)
w.sync_torch()
assert w.lib().wspr_decimate_u8_batch_device(raw.data_ptr(), RAW, n, dI.data_ptr(), dQ.data_ptr(), 1) == 0
dec = w.BatchDecoder(n, 32); dec.decode_ptr(dI.data_ptr(), dQ.data_ptr(), NS, stride)
gi, gq = dI.cpu().numpy(), dQ.cpu().numpy()
L = ol.lib()
def tup(s): return (s.message, s.call, s.loc, s.pwr, s.cycles, s.jitter, s.drift, s.sync, s.dt, s.freq, round(float(s.snr), 3))
bad = 0
for s in range(n):
    host = raw[s].cpu().numpy()
    st = L.orc_decim_new()
    oi = np.zeros(NS, np.float32); oq = np.zeros(NS, np.float32)
    fill = L.orc_decim_feed(C.c_void_p(st), ol.ptr(host), RAW, ol.ptr(oi), ol.ptr(oq), 0, NS)
    L.orc_decim_free(C.c_void_p(st))
    L.orc_normalise(ol.ptr(oi), ol.ptr(oq), C.c_int(fill), C.c_int(NS))
    iq_ok = np.array_equal(gi[s, :NS], oi) and np.array_equal(gq[s, :NS], oq)
    ref, _, _ = ol.decode(oi, oq, NS)
    got = [tup(x) for x in dec.spots(s)]; exp = [tup(x) for x in ref]
    ok = iq_ok and got == exp
    bad += not ok
    print("segment %d: IQ %s, %d spots %s" % (s, "equal" if iq_ok else "DIFFERENT", len(exp), "equal" if got == exp else "DIFFERENT"))
print("raw soak: %d of %d segments equal the oracle" % (n - bad, n))
sys.exit(1 if bad else 0)

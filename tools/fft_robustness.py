#!/usr/bin/env python3
"""How much do the SPOTS depend on which float32 FFT produced the spectrogram?

The one stage where the product's arithmetic is not the reference's is the 512-point FFT (reference: FFTW single
precision, wsprd/wsprd.c:496-500, :544 -- absent here and not reproducible bit for bit; product and oracle: one float32
radix-2 DIF).  `ps` differs between any two float32 FFTs by ~1e-6 relative, and the decoder decides on `ps`: noise
quantile / threshold / local maxima (wsprd.c:590-631) and the strict-> coarse argmax (:654-667).  This script decodes
the same segments through the CPU oracle with each alternative FFT of oracle/orc_fft_alt.c and counts what changes in
the spot lists relative to variant 0 (= the product, bit for bit):

  workloads   c1      BASELINE configs[1]: 1 024 segments x 1 signal at -20 dB            (tests/synth.py)
              c2      BASELINE configs[2]: 8 192 segments x 10 signals, -10..-28 dB       (tests/synth.py, the numpy twin
                      of bench.py's generator: same distribution, other random numbers)
              scenes  3 000 randomised scenes of tests/test_gpu_parity.py (0-6 signals, types 1-3, drift, CW carrier)

  python tools/fft_robustness.py [--workloads c1,c2,scenes] [--variants 1,2,3,4,5,6,7] [--n-c2 8192] [--n-scenes 3000]
                                 [--threads N] [--out profiles/r06_fft_robustness.json]

north_star tolerances: call/loc/pwr exact; SNR +-0.1 dB, dt +-10 ms, freq +-0.1 Hz.  CPU only (test infrastructure).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol      # noqa: E402
import synth                 # noqa: E402

NS = 45000
VARIANT_NAMES = {0: "radix-2 DIF f32 (product)", 1: "float64 FFT rounded to f32", 2: "DIT 4,4,4,4,2 f32",
                 3: "DIT 8,8,8 f32", 4: "DIT 16,32 f32", 5: "DIT 32,16 f32", 6: "DIT 4,4,4,4,2 f32 + FMA twiddles",
                 7: "DIT 2x9 f32"}
TOL = {"snr": 0.1, "dt": 0.010, "freq_hz": 0.1}

_sym = {}


def symbols_of(msg):
    if msg not in _sym:
        ok, s = ol.channel_symbols(msg)
        assert ok, msg
        _sym[msg] = s
    return _sym[msg]


def gen_c1(n):
    return [synth.make_segment(1_000_000 + s, symbols_of, n_signals=1, snr_db=-20.0)[:2] for s in range(n)]


def gen_c2_one(s):
    return synth.make_segment(2_000_000 + s, symbols_of, n_signals=10, snr_db=-10.0, snr_span=18.0, t_jitter=0.3)[:2]


def gen_scenes(n, seed=777):
    import test_gpu_parity as tp
    I, Q = tp.random_scenes(count=n, seed=seed)
    return [(I[k], Q[k]) for k in range(n)]


def spot_rec(s):
    return (s.message.decode(), s.call.decode(), s.loc.decode(), s.pwr.decode(), float(s.snr), float(s.dt),
            float(s.freq), float(s.drift), int(s.jitter), int(s.cycles), float(s.sync))


def decode_all(segs, threads, with_cands):
    def run(k):
        I, Q = segs[k]
        if with_cands:
            spots, _, _, tr = ol.decode(I, Q, NS, trace=True)
            n0 = tr.npk[0]
            cands = tuple((tr.cand_coarse[0][i].freq, tr.cand_coarse[0][i].shift, tr.cand_coarse[0][i].drift)
                          for i in range(n0))
            return [spot_rec(s) for s in spots], cands
        spots, _, _ = ol.decode(I, Q, NS)
        return [spot_rec(s) for s in spots], None
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(run, range(len(segs))))


def compare(base, var):
    """Counts over all segments of what a variant's spot lists differ in from variant 0's."""
    r = dict(segments=len(base), spots_base=0, spots_variant=0, segments_identical=0, segments_spot_set_differs=0,
             spots_lost=0, spots_gained=0, spots_text_changed=0, spots_order_changed_segments=0,
             spots_bit_identical=0, max_dsnr_db=0.0, max_ddt_s=0.0, max_dfreq_hz=0.0, spots_beyond_tolerance=0,
             spots_drift_changed=0, spots_jitter_changed=0, spots_cycles_changed=0, spots_dt_changed=0,
             segments_coarse_candidates_differ=0)
    for (b, bc), (v, vc) in zip(base, var):
        r["spots_base"] += len(b)
        r["spots_variant"] += len(v)
        if bc is not None and bc != vc:
            r["segments_coarse_candidates_differ"] += 1
        bm = {x[0]: x for x in b}
        vm = {x[0]: x for x in v}
        same_everything = (len(b) == len(v)) and all(x[:4] == y[:4] and x[5:] == y[5:] and abs(x[4] - y[4]) < 1e-3
                                                     for x, y in zip(b, v))
        r["segments_identical"] += bool(same_everything)
        lost = [m for m in bm if m not in vm]
        gained = [m for m in vm if m not in bm]
        r["spots_lost"] += len(lost)
        r["spots_gained"] += len(gained)
        r["segments_spot_set_differs"] += bool(lost or gained)
        if not (lost or gained) and [x[0] for x in b] != [x[0] for x in v]:
            r["spots_order_changed_segments"] += 1
        for m, x in bm.items():
            y = vm.get(m)
            if y is None:
                continue
            if x[1:4] != y[1:4]:
                r["spots_text_changed"] += 1          # same message text but other call/loc/pwr fields: cannot happen
            dsnr, ddt, dfr = abs(x[4] - y[4]), abs(x[5] - y[5]), abs(x[6] - y[6]) * 1e6
            r["max_dsnr_db"] = max(r["max_dsnr_db"], dsnr)
            r["max_ddt_s"] = max(r["max_ddt_s"], ddt)
            r["max_dfreq_hz"] = max(r["max_dfreq_hz"], dfr)
            r["spots_beyond_tolerance"] += bool(dsnr > TOL["snr"] or ddt > TOL["dt"] or dfr > TOL["freq_hz"])
            r["spots_drift_changed"] += x[7] != y[7]
            r["spots_jitter_changed"] += x[8] != y[8]
            r["spots_cycles_changed"] += x[9] != y[9]
            r["spots_dt_changed"] += x[5] != y[5]
            r["spots_bit_identical"] += (x[5:] == y[5:] and x[4] == y[4])
    return r


def ps_perturbation(segs, variants):
    """max and rms relative difference of ps (rows 48..464, relative to the row mean) between variant v and variant 0."""
    L = ol.lib()
    out = {}
    blocks = L.orc_blocks_for(NS)
    ref = []
    L.orc_set_fft_variant(0)
    for I, Q in segs:
        ps = np.zeros((512, blocks), np.float32)
        L.orc_fft_bank(ol.ptr(I), ol.ptr(Q), C.c_int(NS), ol.ptr(ps))
        ref.append(ps[48:465].astype(np.float64))
    for v in variants:
        L.orc_set_fft_variant(v)
        mx, sq, n = 0.0, 0.0, 0
        for (I, Q), r0 in zip(segs, ref):
            ps = np.zeros((512, blocks), np.float32)
            L.orc_fft_bank(ol.ptr(I), ol.ptr(Q), C.c_int(NS), ol.ptr(ps))
            d = np.abs(ps[48:465].astype(np.float64) - r0) / r0.mean()
            mx = max(mx, float(d.max())); sq += float((d ** 2).sum()); n += d.size
        out[str(v)] = {"max_rel_to_mean": mx, "rms_rel_to_mean": (sq / n) ** 0.5}
    L.orc_set_fft_variant(0)
    return out


def detail(a):
    L = ol.lib()
    wl = a.workloads.split(",")[0]
    if wl == "c1":
        segs = gen_c1(a.n_c1)
    elif wl == "c2":
        with ThreadPoolExecutor(a.threads) as ex:
            segs = list(ex.map(gen_c2_one, range(a.n_c2)))
    else:
        segs = gen_scenes(a.n_scenes)

    def run(k):
        I, Q = segs[k]
        spots, _, _, tr = ol.decode(I, Q, NS, trace=True)
        cands = [[(round(tr.cand_peaks[p][i].freq, 4), round(tr.cand_peaks[p][i].snr, 4), tr.decoded[p][i])
                  for i in range(tr.npk[p])] for p in range(tr.passes_run)]
        return [spot_rec(s) for s in spots], cands, [float(tr.noise_level[p]) for p in range(tr.passes_run)]
    out = {}
    for v in (0, a.detail):
        L.orc_set_fft_variant(v)
        with ThreadPoolExecutor(a.threads) as ex:
            out[v] = list(ex.map(run, range(len(segs))))
    L.orc_set_fft_variant(0)
    n = 0
    for k, (b, v) in enumerate(zip(out[0], out[a.detail])):
        differs = len(b[0]) != len(v[0]) or any(x[:4] != y[:4] or x[5:] != y[5:] or abs(x[4] - y[4]) > 1e-3 for x, y in zip(b[0], v[0]))
        if not differs:
            continue
        n += 1
        print("segment %d (%s): noise level per pass %r vs %r" % (k, wl, b[2], v[2]))
        for x, y in zip(b[0], v[0]):
            if x != y:
                print("   spot  variant 0:", x, "\n         variant %d:" % a.detail, y)
        for p in range(min(len(b[1]), len(v[1]))):
            if b[1][p] != v[1][p]:
                print("   pass %d candidates (freq, snr, decoded), variant 0: %r" % (p, b[1][p]))
                print("   pass %d candidates (freq, snr, decoded), variant %d: %r" % (p, a.detail, v[1][p]))
    print("%d of %d segments differ beyond the SNR's last digits" % (n, len(segs)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="c1,c2,scenes")
    ap.add_argument("--variants", default="1,2,3,4,5,6,7")
    ap.add_argument("--n-c1", type=int, default=1024)
    ap.add_argument("--n-c2", type=int, default=8192)
    ap.add_argument("--n-scenes", type=int, default=3000)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_fft_robustness.json"))
    ap.add_argument("--detail", type=int, default=None,
                    help="instead of the study: decode the (single) workload with variant 0 and with this variant and print, for "
                         "every segment whose spots differ in anything but the SNR's last digits, both spot lists and both "
                         "passes' candidate lists (which decision flipped)")
    a = ap.parse_args()
    if a.detail is not None:
        return detail(a)
    variants = [int(x) for x in a.variants.split(",") if x]
    L = ol.lib()
    result = {"tolerances": TOL, "variants": {str(v): VARIANT_NAMES[v] for v in [0] + variants}, "workloads": {},
              "threads": a.threads}
    for wl in a.workloads.split(","):
        t0 = time.time()
        if wl == "c1":
            segs = gen_c1(a.n_c1)
        elif wl == "c2":
            with ThreadPoolExecutor(a.threads) as ex:
                segs = list(ex.map(gen_c2_one, range(a.n_c2)))
        elif wl == "scenes":
            segs = gen_scenes(a.n_scenes)
        else:
            raise SystemExit("unknown workload " + wl)
        print("%s: %d segments generated in %.0f s" % (wl, len(segs), time.time() - t0), flush=True)
        L.orc_set_fft_variant(0)
        t0 = time.time()
        base = decode_all(segs, a.threads, True)
        print("%s: variant 0 decoded in %.0f s, %d spots" % (wl, time.time() - t0, sum(len(b) for b, _ in base)),
              flush=True)
        block = {"segments": len(segs), "spots_variant0": sum(len(b) for b, _ in base),
                 "ps_perturbation_first_3_segments": ps_perturbation(segs[:3], variants), "by_variant": {}}
        for v in variants:
            L.orc_set_fft_variant(v)
            t0 = time.time()
            var = decode_all(segs, a.threads, True)
            block["by_variant"][str(v)] = compare(base, var)
            print("%s: variant %d (%s) in %.0f s: %s" % (wl, v, VARIANT_NAMES[v], time.time() - t0,
                                                        json.dumps(block["by_variant"][str(v)])), flush=True)
        L.orc_set_fft_variant(0)
        result["workloads"][wl] = block
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(result, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()

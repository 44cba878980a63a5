"""The wave-parallel Fano search the GPU runs (fano_wave.h), emulated on the host with the kernel's
step structure, against the product's serial host decoder and the golden vectors from the reference
objects: same return code, cycle count and decoded bytes; metric on decoded frames.  Step widths 1,
7 and 64, a small pending-visit store (forces narrow steps), budgets from 50 to 10 000 cycles/bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import rtlsdr_wsprd_amd as w

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fw(tmp_path_factory):
    so = tmp_path_factory.mktemp("fw") / "fano_wave_check.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-mpopcnt", "-shared", "-fPIC", "-o", str(so),
                    os.path.join(ROOT, "tests", "helpers", "fano_wave_check.cpp")], check=True)
    return C.CDLL(str(so))


def _serial(L, mt, soft, maxcycles):
    s = (C.c_ubyte * 162)(*soft)
    dec = (C.c_ubyte * 11)(); a = C.c_uint(); b = C.c_uint(); c = C.c_uint()
    r = L.fano(C.byref(a), C.byref(b), C.byref(c), dec, s, C.c_uint(81), mt, C.c_int(60), C.c_uint(maxcycles))
    return r, a.value, b.value, c.value, list(dec)[:10]


def _wave(fw, mt, soft, maxcycles, width=64, cap=1024):
    s = (C.c_ubyte * 162)(*soft)
    dec = (C.c_ubyte * 10)(); a = C.c_uint(); b = C.c_uint(); c = C.c_uint(); st = C.c_uint(); mx = C.c_uint()
    r = fw.fano_wave_host(C.byref(a), C.byref(b), C.byref(c), dec, s, C.c_uint(81), mt, C.c_int(60), C.c_uint(maxcycles),
                          C.c_int(width), C.c_int(cap), C.byref(st), C.byref(mx))
    return r, a.value, b.value, c.value, list(dec), st.value, mx.value


def _check(ser, wav, tag):
    assert wav[0] == ser[0] and wav[2] == ser[2], (tag, ser[:4], wav[:4])          # ret, cycles
    if ser[0] == 0:
        assert wav[1] == ser[1] and wav[3] == ser[3] == 80 and wav[4] == ser[4], (tag, ser, wav)


def test_wave_search_equals_serial_decoder_and_golden(fw, golden_vectors):
    L = w.lib()
    mt = (C.c_int * 256 * 2)()
    L.wspr_fano_metric_table(mt)
    for v in golden_vectors["fano"]:                       # outputs of the real reference fano.c
        r = _wave(fw, mt, v["symbols"], v["maxcycles"])
        assert (r[0], r[2]) == (v["ret"], v["cycles"])
        if r[0] == 0:
            assert r[1] == v["metric"] and r[4][:len(v["decdata"])] == v["decdata"][:10]
    rng = np.random.default_rng(33)
    enc = (C.c_ubyte * 176)()
    ndec = nto = 0
    worst_store = 0
    for t in range(600):
        data = [int(x) for x in rng.integers(0, 256, 7)] + [0, 0, 0, 0]
        data[6] &= 0xC0
        L.encode(enc, (C.c_ubyte * 11)(*data), C.c_uint(11))
        sigma = [5, 25, 40, 50, 55, 60, 65, 70, 80, 100, 150, 400][t % 12]
        soft = np.clip(np.where(np.frombuffer(enc, np.uint8)[:162] > 0, 178, 78) + rng.normal(0, sigma, 162), 0, 255)
        soft = soft.astype(np.uint8).tolist()
        mc = 10000 if t % 30 == 0 else [50, 300, 1500][(t // 12) % 3]
        ser = _serial(L, mt, soft, mc)
        for width, cap in ((64, 1024), (1, 1024), (7, 1024), (64, 160)):
            wav = _wave(fw, mt, soft, mc, width, cap)
            _check(ser, wav, (t, sigma, mc, width, cap))
            if (width, cap) == (64, 1024):
                worst_store = max(worst_store, wav[6])
                if ser[0] != 0 and mc == 10000:
                    # a full time-out: 810 000 serial cycles in a few thousand wave steps
                    assert wav[5] < 16000, wav[5]
        ndec += ser[0] == 0
        nto += ser[0] != 0
    assert ndec > 150 and nto > 150
    assert worst_store <= 1024


def test_wave_search_degenerate_vectors(fw):
    """All-erasure, saturated and alternating soft symbols: thin and bushy trees, thresholds far below 0."""
    L = w.lib()
    mt = (C.c_int * 256 * 2)()
    L.wspr_fano_metric_table(mt)
    rng = np.random.default_rng(4)
    cases = [[128] * 162, [0] * 162, [255] * 162, [0, 255] * 81, [127, 129] * 81,
             rng.integers(0, 256, 162).tolist(), rng.integers(100, 156, 162).tolist(),
             ([255] * 40 + [0] * 40 + [128] * 82)]
    for k, soft in enumerate(cases):
        for mc in (100, 2000):
            ser = _serial(L, mt, soft, mc)
            for width in (64, 3):
                _check(ser, _wave(fw, mt, soft, mc, width), (k, mc, width))

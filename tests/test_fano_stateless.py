"""The storage-free Fano search the GPU runs (fano_stateless.h), compiled for the host, against the
product's host decoder and the golden vectors from the reference objects: same return code,
metric, cycle count, maxnp and decoded bytes."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import rtlsdr_wsprd_amd as w

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sl(tmp_path_factory):
    so = tmp_path_factory.mktemp("fs") / "fano_stateless_check.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-mpopcnt", "-shared", "-fPIC", "-o", str(so),
                    os.path.join(ROOT, "tests", "helpers", "fano_stateless_check.cpp")], check=True)
    return C.CDLL(str(so))


def _run(lib, fn, mt, soft, maxcycles):
    s = (C.c_ubyte * 162)(*soft)
    dec = (C.c_ubyte * 11)(); a = C.c_uint(); b = C.c_uint(); c = C.c_uint()
    r = getattr(lib, fn)(C.byref(a), C.byref(b), C.byref(c), dec, s, C.c_uint(81), mt, C.c_int(60), C.c_uint(maxcycles))
    return r, a.value, b.value, c.value, (list(dec)[:10] if r == 0 else None)


def test_stateless_equals_host_decoder_and_golden(sl, golden_vectors):
    L = w.lib()
    mt = (C.c_int * 256 * 2)()
    L.wspr_fano_metric_table(mt)
    for v in golden_vectors["fano"]:
        r = _run(sl, "fano_stateless_host", mt, v["symbols"], v["maxcycles"])
        assert r[:4] == (v["ret"], v["metric"], v["cycles"], v["maxnp"])
        if r[0] == 0:
            assert r[4] == v["decdata"]
    rng = np.random.default_rng(21)
    enc = (C.c_ubyte * 176)()
    ndec = 0
    for t in range(400):
        data = [int(x) for x in rng.integers(0, 256, 7)] + [0, 0, 0, 0]
        data[6] &= 0xC0
        L.encode(enc, (C.c_ubyte * 11)(*data), C.c_uint(11))
        sigma = [5, 20, 35, 45, 55, 65, 80, 120][t % 8]
        soft = np.clip(np.where(np.frombuffer(enc, np.uint8)[:162] > 0, 178, 78) + rng.normal(0, sigma, 162), 0, 255)
        soft = soft.astype(np.uint8).tolist()
        mc = 10000 if t % 40 == 0 else [50, 300, 1500][t % 3]
        a = _run(sl, "fano_stateless_host", mt, soft, mc)
        b = _run(L, "fano", mt, soft, mc)
        assert a == b, (t, sigma, mc)
        ndec += a[0] == 0
    assert 100 < ndec < 390          # both outcomes were exercised

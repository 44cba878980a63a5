"""An independent derivation of the receiver front end (rtlsdr_wsprd.c:126-244), written from the reference source alone
and WITHOUT consulting oracle/orc_frontend.c, in NumPy: closed forms instead of the sample loop.

The decimator's oracle is "parity unpinned" (the callback is a `static` function of a translation unit that needs
<rtl-sdr.h>, libusb and libcurl, and the reference holds no decimator fixture -- DESIGN.md section 2).  This test does
not change that; it makes a shared misreading of the source by oracle and kernel much less likely: a third reading,
structured differently (whole-array integer prefix sums modulo 2^32, combs in closed form, the FIR as 33 array
operations in tap order), must give the same bits as both.

  mixer   four samples per 8 bytes: (s0, s1), (-s3, s2), (-s4, -s5), (s7, -s6), s = int8(byte ^ 0x80), the negations
          in int8 (so -(-128) = -128), :170-181
  CIC     two integrators on int32 (wrapping), output at every 6401st sample (decimationIndex <= DOWNSAMPLING skips
          6400), two combs each with a delay of TWO outputs (z, y registers), :189-217
  FIR     out[m] = sum_{j=0..31} float(y2[m-32+j]) * z[j]  +  float(y2[m]) * z[32], accumulated in float32 in that
          order (x86-64 SSE: separately rounded), :219-234
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol

R = 6401                                    # DOWNSAMPLING + 1 input samples per output
_HALF = [-0.0027772683, -0.0005058826, 0.0049745750, -0.0034059318, -0.0077557814, 0.0139375423, 0.0039896935,
         -0.0299394142, 0.0162250643, 0.0405130860, -0.0580746013, -0.0272104968, 0.1183705475, -0.0306029022,
         -0.2011241667, 0.1615898423]
ZCOEF = np.array(_HALF + [0.5] + _HALF[::-1], dtype=np.float32)        # zCoef[33], :142-152 (symmetric)


def first_principles(raw):
    """raw: uint8 interleaved I/Q, length a multiple of 8.  Returns (I, Q) float32, one value per 6401 samples."""
    s = (raw ^ 0x80).view(np.int8).astype(np.int16).reshape(-1, 8)
    neg = lambda v: (-v).astype(np.int8).astype(np.int16)              # int8 negation: -(-128) wraps to -128
    i_rail = np.stack([s[:, 0], neg(s[:, 3]), neg(s[:, 4]), s[:, 7]], axis=1).reshape(-1).astype(np.int64)
    q_rail = np.stack([s[:, 1], s[:, 2], neg(s[:, 5]), neg(s[:, 6])], axis=1).reshape(-1).astype(np.int64)
    out = []
    for x in (i_rail, q_rail):
        m = x.size // R
        x1 = np.cumsum(x)                                             # exact in int64; int32 wrapping = mod 2^32
        x2 = np.cumsum(x1 & 0xFFFFFFFF)                               # (any representative mod 2^32 will do)
        X = (x2[R - 1::R][:m] & 0xFFFFFFFF).astype(np.uint64)         # second integrator at the decimation instants
        Xp = np.concatenate([np.zeros(4, np.uint64), X])
        # comb(comb(X)) with delay 2: y2[m] = X[m] - 2 X[m-2] + X[m-4]  (mod 2^32, then read as int32)
        y2 = (Xp[4:] + ((1 << 33) - 2 * Xp[2:-2]) + Xp[:-4]) & 0xFFFFFFFF
        y2 = y2.astype(np.uint32).view(np.int32).astype(np.float32)   # (float)Iy2
        yp = np.concatenate([np.zeros(32, np.float32), y2])
        acc = np.zeros(m, np.float32)
        for j in range(32):                                           # tap order, separately rounded float32 ops
            acc = acc + yp[j:j + m] * ZCOEF[j]
        acc = acc + y2 * ZCOEF[32]
        out.append(acc)
    return out[0], out[1]


def fixtures():
    rng = np.random.default_rng(20260928)
    n = 8 * ((R * 301 + 1234) // 8)                                   # 301 whole blocks and a ragged tail
    f = {"random": rng.integers(0, 256, n, dtype=np.uint8)}
    clip = rng.integers(0, 256, n, dtype=np.uint8)
    clip[rng.random(n) < 0.3] = 0                                     # 0x00 -> -128: the int8 negation wraps
    clip[rng.random(n) < 0.3] = 255
    f["clipped"] = clip
    dc = np.clip(rng.normal(235, 6, n), 0, 255).astype(np.uint8)      # strong DC: the second integrator wraps int32 often
    f["dc_wrapping"] = dc
    f["all_zero_bytes"] = np.zeros(8 * (R * 40 // 8), np.uint8)       # every sample -128, negated or not
    f["short"] = rng.integers(0, 256, 8 * 700, dtype=np.uint8)        # less than one block: no output
    return f


def _oracle(raw):
    O = ol.lib()
    st = O.orc_decim_new()
    I = np.zeros(45000, np.float32); Q = np.zeros(45000, np.float32)
    n = O.orc_decim_feed(C.c_void_p(st), ol.ptr(raw), raw.size, ol.ptr(I), ol.ptr(Q), 0, 45000)
    O.orc_decim_free(C.c_void_p(st))
    return I[:n], Q[:n]


def test_first_principles_constants():
    taps = np.zeros(33, np.float32); r = C.c_int(0)
    ol.lib().orc_front_end_constants(ol.ptr(taps), C.byref(r))
    assert r.value == R and np.array_equal(taps, ZCOEF)


@pytest.mark.parametrize("name", ["random", "clipped", "dc_wrapping", "all_zero_bytes", "short"])
def test_oracle_equals_first_principles(name):
    raw = fixtures()[name]
    fi, fq = first_principles(raw)
    oi, oq = _oracle(raw)
    assert fi.size == raw.size // 2 // R == oi.size
    assert np.array_equal(fi.view(np.uint32), oi.view(np.uint32)) and np.array_equal(fq.view(np.uint32), oq.view(np.uint32))
    if name == "dc_wrapping":       # the fixture really exercises the wrap: the unwrapped integrator leaves int32
        assert np.cumsum(np.cumsum((raw[0::2] ^ 0x80).view(np.int8).astype(np.int64)))[-1] > 2 ** 40


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["random", "clipped", "dc_wrapping", "all_zero_bytes"])
def test_gpu_front_end_equals_first_principles(name):
    import rtlsdr_wsprd_amd as w
    L = w.lib()
    assert L.wspr_device_ready() == 1
    raw = fixtures()[name]
    fi, fq = first_principles(raw)
    I = np.zeros(45000, np.float32); Q = np.zeros(45000, np.float32); n = C.c_uint32(0)
    assert L.wspr_decimate_u8(ol.ptr(raw), raw.size, ol.ptr(I), ol.ptr(Q), C.byref(n), 0) == 0
    assert n.value == fi.size
    assert np.array_equal(I[:n.value].view(np.uint32), fi.view(np.uint32))
    assert np.array_equal(Q[:n.value].view(np.uint32), fq.view(np.uint32))
    # and the streaming form (carried state, librtlsdr-sized callbacks)
    st = (C.c_uint32 * (1 + 2 + 2 + 72))()
    L.wspr_decim_stream_reset(st)
    I2 = np.zeros(45000, np.float32); Q2 = np.zeros(45000, np.float32); fill = C.c_uint32(0)
    for pos in range(0, raw.size - raw.size % 16, 65536):
        chunk = np.ascontiguousarray(raw[pos:pos + 65536][: (min(65536, raw.size - pos) // 16) * 16])
        assert L.wspr_decimate_u8_stream(st, ol.ptr(chunk), C.c_size_t(chunk.size), ol.ptr(I2), ol.ptr(Q2), fill, 45000,
                                         C.byref(fill)) == 0
    m = fill.value
    assert m >= fi.size - 1 and np.array_equal(I2[:m].view(np.uint32), fi[:m].view(np.uint32))
    assert np.array_equal(Q2[:m].view(np.uint32), fq[:m].view(np.uint32))

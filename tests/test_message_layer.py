"""Host-side message layer of the product (C ABI names of the reference) against
  * tests/golden/message_vectors.json (outputs of the real reference objects), and
  * the CPU oracle on random inputs.  No GPU needed: these entry points are pure host code."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
import rtlsdr_wsprd_amd as w


@pytest.fixture(scope="module")
def L():
    return w.lib()


def test_nhash_pack_golden(L, golden_vectors):
    for call, h in golden_vectors["nhash"]:
        assert L.nhash(call.encode(), len(call), 146) == h
    for call, n in golden_vectors["pack_call"]:
        assert L.pack_call(call.encode()) == n, call
    for grid, p, m in golden_vectors["pack_grid4_power"]:
        codes = bytes(L.get_locator_character_code(C.c_char(ch.encode())) & 0xFF for ch in grid)
        assert L.pack_grid4_power(codes, C.c_int(p)) == m
    assert [L.get_callsign_character_code(C.c_char(c.encode())) for c in "09AZ "] == [0, 9, 10, 35, 36]
    assert [L.get_locator_character_code(C.c_char(c.encode())) for c in "09AR "] == [0, 9, 0, 17, 36]


def test_channel_symbols_golden(golden_vectors):
    for v in golden_vectors["channel_symbols"]:
        ok, sym = w.get_wspr_channel_symbols(v["message"])
        assert ok == v["ok"], v["message"]
        if ok:
            assert "".join(map(str, sym)) == v["symbols"], v["message"]


def test_interleaver_golden(L, golden_vectors):
    a = (C.c_ubyte * 162)(*range(162)); L.interleave(a)
    assert list(a) == golden_vectors["interleave_identity"]
    a = (C.c_ubyte * 162)(*range(162)); L.deinterleave(a)
    assert list(a) == golden_vectors["deinterleave_identity"]
    a = (C.c_ubyte * 162)(*range(162)); L.interleave(a); L.deinterleave(a)
    assert list(a) == list(range(162))


def _unpk(lib, fn, data, pre_hash=()):
    hashtab = C.create_string_buffer(32768 * 13); loctab = C.create_string_buffer(32768 * 5)
    for idx, txt in pre_hash:
        C.memmove(C.addressof(hashtab) + idx * 13, txt.encode(), len(txt))
    msg = (C.c_byte * 12)(*[(b - 256 if b > 127 else b) for b in data] + [0] * 5)
    clp = C.create_string_buffer(23); call = C.create_string_buffer(13); loc = C.create_string_buffer(7)
    pwr = C.create_string_buffer(3); cs = C.create_string_buffer(13)
    r = getattr(lib, fn)(msg, hashtab, loctab, clp, call, loc, pwr, cs)
    return (int(r), clp.value.decode("latin1"), call.value.decode("latin1"), loc.value.decode("latin1"),
            pwr.value.decode("latin1"), cs.value.decode("latin1"))


def test_unpk_golden_and_oracle(L, golden_vectors):
    for v in golden_vectors["unpk"]:
        want = (v["noprint"], v["call_loc_pow"], v["call"], v["loc"], v["pwr"], v["callsign"])
        assert _unpk(L, "unpk_", v["data"], v["pre_hash"]) == want, v["data"]
    rng = np.random.default_rng(3)
    O = ol.lib()
    for _ in range(3000):
        n = int(rng.integers(0, 1 << 28)); m = int(rng.integers(0, 1 << 22))
        d = [(n >> 20) & 255, (n >> 12) & 255, (n >> 4) & 255, ((n & 15) << 4) | ((m >> 18) & 15),
             (m >> 10) & 255, (m >> 2) & 255, (m & 3) << 6]
        assert _unpk(L, "unpk_", d) == _unpk(O, "orc_unpk", d)


def _fano(lib, fn, mt, soft, maxcycles):
    s = (C.c_ubyte * 162)(*soft)
    dec = (C.c_ubyte * 11)(); a = C.c_uint(); b = C.c_uint(); c = C.c_uint()
    r = getattr(lib, fn)(C.byref(a), C.byref(b), C.byref(c), dec, s, C.c_uint(81), mt, C.c_int(60), C.c_uint(maxcycles))
    return r, a.value, b.value, c.value, (list(dec)[:10] if r == 0 else None)


def test_fano_golden_and_oracle(L, golden_vectors):
    mt = (C.c_int * 256 * 2)()
    L.wspr_fano_metric_table(mt)
    assert [mt[0][i] for i in range(256)] == golden_vectors["mettab0"]
    assert [mt[1][i] for i in range(256)] == golden_vectors["mettab0"][::-1]
    for v in golden_vectors["fano"]:
        r = _fano(L, "fano", mt, v["symbols"], v["maxcycles"])
        assert r[:4] == (v["ret"], v["metric"], v["cycles"], v["maxnp"])
        if r[0] == 0:
            assert r[4] == v["decdata"]
    rng = np.random.default_rng(9)
    O = ol.lib()
    for t in range(80):
        soft = (rng.integers(0, 256, 162) if t % 4 == 0 else
                np.clip(np.where(rng.integers(0, 2, 162) > 0, 180, 76) + rng.normal(0, 35, 162), 0, 255)).astype(np.uint8)
        assert _fano(L, "fano", mt, soft.tolist(), 300) == _fano(O, "orc_fano", mt, soft.tolist(), 300)


def test_encode_fano_roundtrip_like_reference_unit_test(L):
    """Reference tests/test_wsprd.c:168-220: hard 0/255 symbols decode back to the 7 payload bytes."""
    n = L.pack_call(b"K1JT")
    g4 = bytes(L.get_locator_character_code(C.c_char(c.encode())) & 0xFF for c in "FN20")
    m = L.pack_grid4_power(g4, C.c_int(20))
    data = [(n >> 20) & 255, (n >> 12) & 255, (n >> 4) & 255, ((n & 15) << 4) + ((m >> 18) & 15),
            (m >> 10) & 255, (m >> 2) & 255, (m & 3) << 6, 0, 0, 0, 0]
    enc = (C.c_ubyte * 176)()
    L.encode(enc, (C.c_ubyte * 11)(*data), C.c_uint(11))
    mt = (C.c_int * 256 * 2)(); L.wspr_fano_metric_table(mt)
    r = _fano(L, "fano", mt, [255 if enc[i] else 0 for i in range(162)], 10000)
    assert r[0] == 0 and r[4][:7] == data[:7]
    n1 = C.c_int32(); n2 = C.c_int32()
    L.unpack50((C.c_byte * 11)(*[(x - 256 if x > 127 else x) for x in data]), C.byref(n1), C.byref(n2))
    assert (n1.value, n2.value) == (n, m)
    out = C.create_string_buffer(13)
    assert L.unpackcall(C.c_int32(262177560), out) == 0
    g = C.create_string_buffer(5)
    assert L.unpackgrid(C.c_int32(32400 << 7), g) == 0 and g.value == b"XXXX"
    assert L.pack_call(b"TOOLONG1") == 0
    partab = (C.c_ubyte * 256).in_dll(L, "Partab")
    assert [partab[i] for i in (0, 1, 3, 7, 255, 128)] == [0, 1, 0, 1, 0, 1]


def test_unpack_and_channel_symbols_against_the_real_reference_objects(L):
    """The decoder's per-decode host work (unpk_ on the decoded bits, get_wspr_channel_symbols on the text for the
    subtraction) is written for speed since round 6 (texts put together by hand, the encoder + interleaver as XORs of
    fixed patterns): against oracle/_ref (the reference's own wsprd_utils.c / wsprsim_utils.c / fano.c / nhash.c) on
    20 000 random 50-bit messages of every type, and on the texts they unpack to, with the hash tables carried along."""
    R = ol.ref_lib()
    if R is None:
        pytest.skip("oracle/_ref/libwsprd_ref.so not present")
    import synth
    rng = np.random.default_rng(2026)

    def tables():
        return C.create_string_buffer(32768 * 13), C.create_string_buffer(32768 * 5)
    ph, pl = tables(); rh, rl = tables()

    def unpk(lib, fn, d, hashtab, loctab):
        msg = (C.c_byte * 12)(*[(b - 256 if b > 127 else b) for b in d] + [0] * 5)
        clp = C.create_string_buffer(23); call = C.create_string_buffer(13); loc = C.create_string_buffer(7)
        pwr = C.create_string_buffer(3); cs = C.create_string_buffer(13)
        r = getattr(lib, fn)(msg, hashtab, loctab, clp, call, loc, pwr, cs)
        return int(r), clp.value, call.value, loc.value, pwr.value, cs.value

    def symbols(lib, text, hashtab, loctab):
        sym = (C.c_ubyte * 162)(); msg = C.create_string_buffer(text, 32)
        ok = lib.get_wspr_channel_symbols(msg, hashtab, loctab, sym)
        return int(ok), bytes(sym) if ok else None
    L.get_wspr_channel_symbols.restype = C.c_int
    n_types = [0, 0, 0]
    for t in range(20000):
        if t % 4 == 0:                                   # a valid plain message (the common case) ...
            msg = synth.message_wide(int(rng.integers(0, 1 << 62))).encode()
            ok, s = symbols(R, msg, rh, rl)
            assert (ok, s) == symbols(L, msg, ph, pl), msg
            continue
        n = int(rng.integers(0, 1 << 28)); m = int(rng.integers(0, 1 << 22))        # ... and arbitrary bits
        d = [(n >> 20) & 255, (n >> 12) & 255, (n >> 4) & 255, ((n & 15) << 4) | ((m >> 18) & 15),
             (m >> 10) & 255, (m >> 2) & 255, (m & 3) << 6]
        want = unpk(R, "unpk_", d, rh, rl)
        assert unpk(L, "unpk_", d, ph, pl) == want, d
        ntype = (m & 127) - 64
        n_types[0 if ntype < 0 else (1 if ntype % 10 in (0, 3, 7) else 2)] += 1
        if want[0] == 0 and b"." not in want[1] and want[1].isascii():          # the decoder re-encodes what it prints
            ok, s = symbols(R, want[1], rh, rl)
            assert (ok, s) == symbols(L, want[1], ph, pl), want[1]
    assert min(n_types) > 1000
    assert ph.raw == rh.raw and pl.raw == rl.raw          # the hash memories saw the same stores

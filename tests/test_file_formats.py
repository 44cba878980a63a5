"""recorded-file readers/writer and the playback spot line of the C ABI
(host code, no GPU) against the oracle's reader semantics and the reference's documented output."""
import ctypes as C
import os
import struct

import numpy as np

import oracle_lib as ol
import rtlsdr_wsprd_amd as w

REF_IQ = os.path.join(ol.GOLDEN, "refSignalSnr0dB.iq")


def test_iq_reader_equals_oracle_reader():
    L = w.lib()
    I = np.zeros(45000, np.float32); Q = np.zeros(45000, np.float32)
    n = L.wspr_read_iq_file(REF_IQ.encode(), ol.ptr(I), ol.ptr(Q))
    oi, oq, on = ol.read_iq_file(REF_IQ)
    assert n == on == 45000
    assert np.array_equal(I, oi) and np.array_equal(Q, oq)
    assert max(np.abs(I).max(), np.abs(Q).max()) == np.float32(0.5)
    assert L.wspr_read_iq_file(b"/nonexistent/file.iq", ol.ptr(I), ol.ptr(Q)) == 0


def test_iq_write_read_roundtrip_and_c2(tmp_path):
    L = w.lib()
    rng = np.random.default_rng(2)
    I = rng.normal(0, 0.1, 45000).astype(np.float32); Q = rng.normal(0, 0.1, 45000).astype(np.float32)
    p = str(tmp_path / "x.iq").encode()
    assert L.wspr_write_iq_file(p, ol.ptr(I), ol.ptr(Q)) == 45000
    raw = np.fromfile(p.decode(), np.float32)
    assert raw.size == 90000 and np.array_equal(raw[0::2], I) and np.array_equal(raw[1::2], -Q)   # Q negated on disk
    I2 = np.zeros(45000, np.float32); Q2 = np.zeros(45000, np.float32)
    assert L.wspr_read_iq_file(p, ol.ptr(I2), ol.ptr(Q2)) == 45000
    scale = np.float32(0.5 / float(max(np.abs(I).max(), np.abs(Q).max())))
    assert np.array_equal(I2, I * scale) and np.array_equal(Q2, Q * scale)
    # .c2: 14-byte name + int + double + payload (rtlsdr_wsprd.c:634-640); short record
    c2 = tmp_path / "x.c2"
    with open(c2, "wb") as f:
        f.write(b"150426_0918.c2")
        f.write(struct.pack("<i", 2))
        f.write(struct.pack("<d", 14095600.0))
        raw[:80000].tofile(f)
    dial = C.c_double()
    I3 = np.zeros(45000, np.float32); Q3 = np.zeros(45000, np.float32)
    L.wspr_read_c2_file.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.wspr_read_c2_file(str(c2).encode(), ol.ptr(I3), ol.ptr(Q3), C.byref(dial)) == 40000
    assert dial.value == 14095600.0
    sc = np.float32(0.5 / float(max(np.abs(I[:40000]).max(), np.abs(Q[:40000]).max())))
    assert np.array_equal(I3[:40000], I[:40000] * sc) and not I3[40000:].any()


def test_spot_line_format_matches_reference_report():
    L = w.lib()
    r = w.decoder_results(freq=144.490550005, sync=0.9, snr=-0.0706653595, dt=0.00533333328, drift=0.0,
                          jitter=0, message=b"K1JT FN20 20", call=b"K1JT", loc=b"FN20", pwr=b"20", cycles=82)
    buf = C.create_string_buffer(128)
    L.wspr_format_spot(C.byref(r), buf, 128)
    assert buf.value.decode() == "Spot :  -0.07   0.01 144.490550  0    K1JT   FN20 20"    # REPORT.md:202


def test_daemon_line_and_wsprnet_url_formats():
    """rtlsdr_wsprd.c:447-474 and :390-429 (text only)."""
    L = w.lib()
    r = w.decoder_results(freq=14.097150, sync=0.9, snr=-21.4, dt=0.31, drift=-1.0, jitter=0,
                          message=b"K1JT FN20 20", call=b"K1JT", loc=b"FN20", pwr=b"20", cycles=82)
    buf = C.create_string_buffer(600)
    L.wspr_format_spot_timestamped(C.byref(r), 2026, 9, 28, 1, 30, buf, 600)
    assert buf.value.decode() == "Spot :  2026-09-28 01:30z -21.40   0.31  14.097150 -1    K1JT   FN20 20"
    opt = w.default_options(freq=14095600)
    opt.rcall = b"VA2GKA/P"
    opt.rloc = b"FN35"
    L.wspr_format_wsprnet_url.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_char_p, C.c_char_p, C.c_size_t]
    L.wspr_format_wsprnet_url(C.addressof(r), C.addressof(opt), 14095600.0, 2026, 9, 28, 1, 30, b"rtlsdr-056", buf, 600)
    assert buf.value.decode() == ("https://wsprnet.org/post?function=wspr&rcall=VA2GKA%2FP&rgrid=FN35&rqrg=14.097150&date=260928"
                                  "&time=0130&sig=-21&dt=0.3&tqrg=14.097150&tcall=K1JT&tgrid=FN20&dbm=20&version=rtlsdr-056&mode=2")
    L.wspr_format_wsprnet_url(None, C.addressof(opt), 14095600.0, 2026, 9, 28, 1, 30, b"rtlsdr-056", buf, 600)
    assert buf.value.decode() == ("https://wsprnet.org/post?function=wsprstat&rcall=VA2GKA%2FP&rgrid=FN35&rqrg=14.095600&tpct=0.00"
                                  "&tqrg=14.095600&dbm=0&version=rtlsdr-056&mode=2")


def test_slot_timing_helpers():
    """Main-loop sleep (rtlsdr_wsprd.c:1170-1175) and the frame's time stamp gmtime(now - 120 + 1) (:307-310)."""
    L = w.lib()
    L.wspr_usec_to_next_slot.restype = C.c_uint32
    L.wspr_usec_to_next_slot.argtypes = [C.c_long, C.c_long]
    assert L.wspr_usec_to_next_slot(1200, 0) == 120000000
    assert L.wspr_usec_to_next_slot(1200 + 119, 999999) == 1
    assert L.wspr_usec_to_next_slot(1790567424, 437248) == 120000000 - ((1790567424 % 120) * 1000000 + 437248)
    L.wspr_frame_time.argtypes = [C.c_long] + [C.c_void_p] * 5
    v = [C.c_int() for _ in range(5)]
    L.wspr_frame_time(1790567520, *[C.byref(x) for x in v])        # 2026-09-28 04:32:00 UTC
    import datetime
    t = datetime.datetime.fromtimestamp(1790567520 - 120 + 1, datetime.timezone.utc)
    assert [x.value for x in v] == [t.year, t.month, t.day, t.hour, t.minute]
    # a session object exists without a GPU; an empty buffer is "too short": nothing is decoded, nothing fails
    L.wspr_session_create.restype = C.c_void_p
    L.wspr_session_create.argtypes = [w.decoder_options]
    s = L.wspr_session_create(w.default_options())
    assert s
    L.wspr_session_rollover.argtypes = [C.c_void_p]
    L.wspr_session_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.wspr_session_fill.argtypes = [C.c_void_p, C.c_int]
    L.wspr_session_fill.restype = C.c_uint32
    assert L.wspr_session_rollover(s) == 0 and L.wspr_session_rollover(s) == 1
    n = C.c_int(5)
    out = (w.decoder_results * 50)()
    assert L.wspr_session_decode(s, 0, C.addressof(out), C.byref(n)) == 0 and n.value == 0
    assert L.wspr_session_fill(s, 0) == 0
    L.wspr_session_destroy.argtypes = [C.c_void_p]
    L.wspr_session_destroy(s)

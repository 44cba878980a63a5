"""ctypes bindings of the CPU oracle (oracle/liboracle.so) and of the real reference
message-layer objects (oracle/_ref/libwsprd_ref.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")

NSYM, MAXCAND, NSAMP = 162, 200, 45000


class Options(C.Structure):            # reference wsprd/wsprd.h:44-52
    _fields_ = [("freq", C.c_int), ("rcall", C.c_char * 13), ("rloc", C.c_char * 7),
                ("quickmode", C.c_int), ("usehashtable", C.c_int),
                ("npasses", C.c_int), ("subtraction", C.c_int)]


class Spot(C.Structure):               # reference wsprd/wsprd.h:62-74
    _fields_ = [("freq", C.c_double), ("sync", C.c_float), ("snr", C.c_float),
                ("dt", C.c_float), ("drift", C.c_float), ("jitter", C.c_int),
                ("message", C.c_char * 23), ("call", C.c_char * 13),
                ("loc", C.c_char * 7), ("pwr", C.c_char * 3), ("cycles", C.c_int)]

    def key(self):
        return (self.message.decode(), self.call.decode(), self.loc.decode(), self.pwr.decode())

    def as_dict(self):
        return dict(freq=self.freq, sync=self.sync, snr=self.snr, dt=self.dt, drift=self.drift,
                    jitter=self.jitter, message=self.message.decode(), call=self.call.decode(),
                    loc=self.loc.decode(), pwr=self.pwr.decode(), cycles=self.cycles)


class Cand(C.Structure):               # reference wsprd/wsprd.h:54-60
    _fields_ = [("freq", C.c_float), ("snr", C.c_float), ("shift", C.c_int),
                ("drift", C.c_float), ("sync", C.c_float)]


P = 3


class Trace(C.Structure):
    _fields_ = [("passes_run", C.c_int), ("blocks", C.c_int),
                ("noise_level", C.c_float * P), ("npk", C.c_int * P),
                ("smspec_raw", (C.c_float * 411) * P),
                ("cand_peaks", (Cand * MAXCAND) * P),
                ("cand_coarse", (Cand * MAXCAND) * P),
                ("cand_fine", (Cand * MAXCAND) * P),
                ("mode0_shift", (C.c_int * MAXCAND) * P),
                ("mode0_sync", (C.c_float * MAXCAND) * P),
                ("n_visited", C.c_int * P),
                ("attempts", (C.c_int * MAXCAND) * P),
                ("fano_calls", (C.c_int * MAXCAND) * P),
                ("decoded", (C.c_int * MAXCAND) * P),
                ("subtracted", (C.c_int * MAXCAND) * P),
                ("first_rms", (C.c_float * MAXCAND) * P),
                ("first_sync2", (C.c_float * MAXCAND) * P),
                ("first_symbols", ((C.c_ubyte * NSYM) * MAXCAND) * P),
                ("fano_metric", (C.c_uint * MAXCAND) * P),
                ("fano_cycles", (C.c_uint * MAXCAND) * P),
                ("fano_maxnp", (C.c_uint * MAXCAND) * P),
                ("decdata", ((C.c_ubyte * 11) * MAXCAND) * P),
                ("fano_cycles_total", C.c_long)]


def default_options(freq=144489000, npasses=2, subtraction=1, quickmode=0):
    """Decoder defaults of rtlsdr_wsprd.c:357-362; dial 144.489 MHz = the '2m' band."""
    return Options(freq=freq, quickmode=quickmode, usehashtable=0,
                   npasses=npasses, subtraction=subtraction)


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.orc_wspr_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, Options,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_wspr_decode.restype = C.c_int
        L.orc_nhash.restype = C.c_uint32
        L.orc_nhash.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
        L.orc_pack_call.restype = C.c_ulong
        L.orc_pack_call.argtypes = [C.c_char_p]
        L.orc_pack_grid4_power.restype = C.c_ulong
        L.orc_fano.restype = C.c_int
        L.orc_unpk.restype = C.c_int
        L.orc_channel_symbols.restype = C.c_int
        L.orc_pick_peaks.restype = C.c_int
        L.orc_blocks_for.restype = C.c_int
        L.orc_decim_new.restype = C.c_void_p
        L.orc_decim_feed.restype = C.c_uint32
        L.orc_decim_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                     C.c_uint32, C.c_uint32]
        L.orc_decim_free.argtypes = [C.c_void_p]
        L.orc_sync_demod.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_subtract.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_float, C.c_int,
                                   C.c_float, C.c_void_p]
        L.orc_iq_from_interleaved.restype = C.c_int
        _lib = L
    return _lib


def ref_lib():
    """Real reference objects (fano.c, nhash.c, wsprd_utils.c, wsprsim_utils.c, tab.c).
    Returns None when the prebuilt library is not present."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libwsprd_ref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.nhash.restype = C.c_uint32
        R.nhash.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
        R.pack_call.restype = C.c_ulong
        R.pack_call.argtypes = [C.c_char_p]
        R.pack_grid4_power.restype = C.c_ulong
        R.fano.restype = C.c_int
        R.unpk_.restype = C.c_int
        R.get_wspr_channel_symbols.restype = C.c_int
        _ref = R
    return _ref


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def read_iq_file(path):
    """.iq reader semantics of rtlsdr_wsprd.c:555-592 via the oracle."""
    raw = np.fromfile(path, dtype=np.float32)
    I = np.zeros(NSAMP, np.float32)
    Q = np.zeros(NSAMP, np.float32)
    n = lib().orc_iq_from_interleaved(ptr(raw), C.c_int(raw.size), ptr(I), ptr(Q))
    return I, Q, n


def decode(I, Q, samples=None, opt=None, trace=False):
    """Run the oracle decoder on copies of I/Q. Returns (spots, residual I, Q[, trace])."""
    L = lib()
    I = np.ascontiguousarray(I, dtype=np.float32).copy()
    Q = np.ascontiguousarray(Q, dtype=np.float32).copy()
    n = int(samples if samples is not None else I.size)
    opt = opt or default_options()
    spots = (Spot * 100)()
    nres = C.c_int(0)
    tr = Trace() if trace else None
    L.orc_wspr_decode(ptr(I), ptr(Q), n, opt, C.addressof(spots), C.addressof(nres),
                      C.addressof(tr) if trace else None)
    out = [spots[i] for i in range(nres.value)]
    return (out, I, Q, tr) if trace else (out, I, Q)


def channel_symbols(message, L=None):
    L = L or lib()
    hashtab = C.create_string_buffer(32768 * 13)
    loctab = C.create_string_buffer(32768 * 5)
    sym = (C.c_ubyte * NSYM)()
    msg = C.create_string_buffer(message.encode(), 32)
    ok = L.orc_channel_symbols(msg, hashtab, loctab, sym)
    return ok, np.frombuffer(sym, dtype=np.uint8).copy()


def spot_line(s):
    """-r print format, rtlsdr_wsprd.c:691-701."""
    return "Spot : %6.2f %6.2f %10.6f %2d %7s %6s %2s" % (
        s.snr, s.dt, s.freq, int(s.drift), s.call.decode(), s.loc.decode(), s.pwr.decode())

"""rtlsdr-wsprd_amd -- MI355X-native WSPR decode path.

Thin ctypes mirror of the C ABI in include/wspr_mi355x.h (the product is the shared
library rtlsdr-wsprd_amd/libwspr_mi355x.so: hand-written HIP kernels for gfx950 + a
C++ host scheduler).  Names follow the reference's C interface
(wsprd/wsprd.h:44-111): decoder_options, decoder_results, wspr_decode.

There is NO CPU fallback: every entry point raises if the library is missing and
the library itself refuses to run without a HIP device.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwspr_mi355x.so")

NSAMPLES = 45000
NSYM = 162
MAX_CANDIDATES = 200


class decoder_options(C.Structure):          # wsprd/wsprd.h:44-52
    _fields_ = [("freq", C.c_int), ("rcall", C.c_char * 13), ("rloc", C.c_char * 7),
                ("quickmode", C.c_int), ("usehashtable", C.c_int),
                ("npasses", C.c_int), ("subtraction", C.c_int)]


class decoder_results(C.Structure):          # wsprd/wsprd.h:62-74
    _fields_ = [("freq", C.c_double), ("sync", C.c_float), ("snr", C.c_float),
                ("dt", C.c_float), ("drift", C.c_float), ("jitter", C.c_int),
                ("message", C.c_char * 23), ("call", C.c_char * 13),
                ("loc", C.c_char * 7), ("pwr", C.c_char * 3), ("cycles", C.c_int)]

    def as_dict(self):
        return dict(freq=self.freq, sync=self.sync, snr=self.snr, dt=self.dt, drift=self.drift,
                    jitter=self.jitter, message=self.message.decode(), call=self.call.decode(),
                    loc=self.loc.decode(), pwr=self.pwr.decode(), cycles=self.cycles)


class cand(C.Structure):                     # wsprd/wsprd.h:54-60
    _fields_ = [("freq", C.c_float), ("snr", C.c_float), ("shift", C.c_int),
                ("drift", C.c_float), ("sync", C.c_float)]


TRACE_PASSES = 3


class cand_trace(C.Structure):               # include/wspr_mi355x.h: wspr_cand_trace
    _fields_ = [("visited", C.c_int), ("mode0_shift", C.c_int), ("mode0_sync", C.c_float),
                ("freq", C.c_float), ("shift", C.c_int), ("drift", C.c_float), ("sync", C.c_float),
                ("attempts", C.c_int), ("fano_calls", C.c_int), ("first_sync", C.c_float), ("first_rms", C.c_float),
                ("decoded", C.c_int), ("subtracted", C.c_int), ("jitter", C.c_int), ("cycles", C.c_uint),
                ("first_symbols", C.c_ubyte * NSYM), ("decdata", C.c_ubyte * 11), ("pad", C.c_ubyte * 3)]


class trace(C.Structure):                    # include/wspr_mi355x.h: wspr_trace
    _fields_ = [("passes_run", C.c_int), ("npk", C.c_int * TRACE_PASSES), ("n_visited", C.c_int * TRACE_PASSES),
                ("cand", (cand_trace * MAX_CANDIDATES) * TRACE_PASSES)]


def default_options(freq=144489000, npasses=2, subtraction=1, quickmode=0):
    """initDecoder_options(), rtlsdr_wsprd.c:357-362."""
    return decoder_options(freq=freq, quickmode=quickmode, usehashtable=0,
                           npasses=npasses, subtraction=subtraction)


def build(verbose=False):
    """Compile the HIP extension in-tree for gfx950 (works without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(["bash", os.path.join(_HERE, "csrc", "build.sh")], check=True, stdout=out, stderr=out)


LAB_PATH = os.path.join(_HERE, "libwspr_mi355x_lab.so")
_lib = None
_lab = None


def _bind(path):
    """CDLL + argument types of every entry point the library exports (the lab-only ones where present)."""
    L = C.CDLL(path)
    L.wspr_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, decoder_options, C.c_void_p, C.c_void_p]
    L.wspr_decode.restype = C.c_int
    L.wspr_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, decoder_options,
                                    C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.wspr_decode_batch.restype = C.c_int
    L.wspr_decode_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t,
                                           decoder_options, C.c_void_p, C.c_int, C.c_void_p]
    L.wspr_decode_batch_device.restype = C.c_int
    L.wspr_iq_stride.restype = C.c_size_t
    L.wspr_mi355x_version.restype = C.c_char_p
    L.wspr_device_ready.restype = C.c_int
    L.sync_and_demodulate.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_int]
    L.sync_and_demodulate.restype = None
    L.subtract_signal2.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_float, C.c_int, C.c_float, C.c_void_p]
    L.subtract_signal2.restype = None
    L.wspr_last_timings.argtypes = [C.c_void_p, C.c_int]
    L.wspr_decimate_u8.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.wspr_decimate_u8_batch_device.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.wspr_pin_host_buffer.argtypes = [C.c_void_p, C.c_size_t]
    L.wspr_unpin_host_buffer.argtypes = [C.c_void_p]
    L.wspr_release_buffers.restype = C.c_size_t
    L.wspr_set_fano_fast_budget.restype = C.c_uint
    L.nhash.restype = C.c_uint32
    L.nhash.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
    L.pack_call.restype = C.c_ulong
    L.pack_call.argtypes = [C.c_char_p]
    L.pack_grid4_power.restype = C.c_ulong
    L.get_callsign_character_code.restype = C.c_byte
    L.get_locator_character_code.restype = C.c_byte
    L.fano.restype = C.c_int
    L.unpk_.restype = C.c_int
    L.get_wspr_channel_symbols.restype = C.c_int
    if hasattr(L, "wspr_stage_fft_bank"):            # include/wspr_mi355x_bench.h: the lab build only
        L.wspr_stage_fft_bank.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p]
        L.wspr_stage_fft_bank.restype = C.c_int
        L.wspr_stage_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wspr_stage_candidates.restype = C.c_int
        L.wspr_bench_fft_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p]
        L.wspr_bench_valu.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p]
        L.wspr_bench_decimate.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.wspr_calib_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        L.wspr_calib_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.wspr_calib_copy16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        L.wspr_calib_valu.argtypes = [C.c_int, C.c_void_p]
        L.wspr_decode_batch_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, decoder_options,
                                              C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    return L


def lib():
    """The PRODUCT library, libwspr_mi355x.so (raises if it has not been built).  WSPR_USE_LAB=1 makes this the lab
    library instead, for whole-suite runs of the lab build and for tests that need one of its environment switches."""
    global _lib
    if os.environ.get("WSPR_USE_LAB") == "1":
        return lab()
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libwspr_mi355x.so is not built: run rtlsdr-wsprd_amd/csrc/build.sh "
                               "(there is no CPU fallback)")
        _lib = _bind(LIB_PATH)
    return _lib


def lab():
    """The LAB build of the same sources, libwspr_mi355x_lab.so: everything the product exports plus the stage-level
    parity hooks, the per-candidate trace, kernel timings and calibration kernels of include/wspr_mi355x_bench.h.
    A separate library with its own contexts and streams: tests and bench.py use it for those calls only."""
    global _lab
    if _lab is None:
        if not os.path.exists(LAB_PATH):
            raise RuntimeError("libwspr_mi355x_lab.so is not built: run rtlsdr-wsprd_amd/csrc/build.sh")
        _lab = _bind(LAB_PATH)
    return _lab


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def get_wspr_channel_symbols(message):
    """wsprsim_utils.h:9 -- returns (ok, symbols[162] uint8) with fresh hash tables."""
    hashtab = C.create_string_buffer(32768 * 13)
    loctab = C.create_string_buffer(32768 * 5)
    sym = (C.c_ubyte * NSYM)()
    ok = lib().get_wspr_channel_symbols(C.create_string_buffer(message.encode(), 32), hashtab, loctab, sym)
    return int(ok), np.frombuffer(sym, dtype=np.uint8).copy()


def wspr_decode(idat, qdat, samples=None, options=None):
    """The reference entry point (wsprd/wsprd.h:106-111) on one segment.
    Returns (spots, residual_i, residual_q); inputs are left untouched."""
    I = np.ascontiguousarray(idat, dtype=np.float32).copy()
    Q = np.ascontiguousarray(qdat, dtype=np.float32).copy()
    n = int(samples if samples is not None else I.size)
    out = (decoder_results * 100)()
    nres = C.c_int(0)
    rc = lib().wspr_decode(_ptr(I), _ptr(Q), n, options or default_options(), C.addressof(out), C.addressof(nres))
    if rc < 0:
        raise RuntimeError("wspr_decode failed (no usable HIP device?)")
    return [out[i] for i in range(nres.value)], I, Q


def wspr_decode_batch(I, Q, options=None, max_results=50):
    """Host arrays [nseg, samples] -> list of spot lists."""
    I = np.ascontiguousarray(I, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    nseg, samples = I.shape
    out = (decoder_results * (nseg * max_results))()
    nres = (C.c_int * nseg)()
    rc = lib().wspr_decode_batch(_ptr(I), _ptr(Q), nseg, samples, samples, options or default_options(),
                                 C.addressof(out), max_results, C.addressof(nres), 0)
    if rc < 0:
        raise RuntimeError("wspr_decode_batch failed (rc %d: no usable HIP device, or usehashtable on a batch)" % rc)
    return [[out[s * max_results + i] for i in range(nres[s])] for s in range(nseg)]


def wspr_decode_batch_trace(I, Q, options=None, max_results=50):
    """wspr_decode_batch() + the per-candidate trace of the fine search (what the production kernels produced for
    every candidate the reference's loop enters), through the LAB library (include/wspr_mi355x_bench.h).
    Returns (spot lists, trace array [nseg])."""
    I = np.ascontiguousarray(I, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    nseg, samples = I.shape
    out = (decoder_results * (nseg * max_results))()
    nres = (C.c_int * nseg)()
    tr = (trace * nseg)()
    L = lab()
    rc = L.wspr_decode_batch_trace(_ptr(I), _ptr(Q), nseg, samples, samples, options or default_options(),
                                   C.addressof(out), max_results, C.addressof(nres), C.addressof(tr))
    if rc < 0:
        raise RuntimeError("wspr_decode_batch_trace failed (rc %d)" % rc)
    return [[out[s * max_results + i] for i in range(nres[s])] for s in range(nseg)], tr


def sync_torch():
    """The library runs on streams of its own (include/wspr_mi355x.h, stream contract): whatever torch still has
    in flight for a tensor -- the index or fill kernel that built it a moment ago -- must have landed before a
    device pointer to it is handed over.  Call this between the torch code and a raw-pointer entry point."""
    import sys
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.current_stream().synchronize()


class BatchDecoder:
    """Decodes segments already resident in HBM (torch tensors or raw device pointers)."""

    def __init__(self, nseg, max_results=16, options=None):
        self.nseg, self.max_results = nseg, max_results
        self.options = options or default_options()
        self.out = (decoder_results * (nseg * max_results))()
        self.nres = (C.c_int * nseg)()

    def decode_ptr(self, d_i, d_q, samples, stride):
        rc = lib().wspr_decode_batch_device(d_i, d_q, self.nseg, samples, stride, self.options,
                                            C.addressof(self.out), self.max_results, C.addressof(self.nres))
        if rc < 0:
            raise RuntimeError("wspr_decode_batch_device failed")
        return self.nres

    def decode(self, ti, tq):
        """ti, tq: contiguous float32 CUDA tensors [nseg, samples]."""
        assert ti.is_cuda and tq.is_cuda and ti.is_contiguous() and tq.is_contiguous()
        sync_torch()
        return self.decode_ptr(ti.data_ptr(), tq.data_ptr(), ti.shape[1], ti.stride(0))

    def spots(self, s):
        return [self.out[s * self.max_results + i] for i in range(self.nres[s])]

    def total_spots(self):
        return int(sum(self.nres))


def last_timings():
    ms = (C.c_double * 26)()
    n = lib().wspr_last_timings(C.addressof(ms), 26)
    names = ["fft_sync_ms", "host_bookkeeping_ms", "device_fano_tail_ms", "demod_ms", "subtract_ms", "host_fano_ms",
             "total_ms", "fano_calls", "fano_timeouts", "fano_cycles", "candidates_refined", "gpu_waves",
             "fano_left_to_device", "segments_redecoded", "candidates_consumed", "subtractions",
             "cpu_ms_call", "cpu_ms_pass_start", "cpu_ms_build_wave", "cpu_ms_refine", "cpu_ms_ladder", "cpu_ms_books",
             "cpu_ms_subtract", "cpu_ms_finish", "message_cache_lookups", "message_cache_hits"]
    return {names[i]: ms[i] for i in range(n)}

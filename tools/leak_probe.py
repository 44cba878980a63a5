#!/usr/bin/env python3
"""Does a long-running service grow?  Host RSS and free HBM after 60 and after 460 rounds of: a host-buffer batch call
(pageable, then pinned + unpinned again), a usehashtable batch call, a resident call, a receiver session created, fed,
rolled over, decoded and destroyed.  Growth between the two marks is what repeats; the first rounds allocate what the
library keeps (contexts, grow-only buffers, pools)."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np
import psutil
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import rtlsdr_wsprd_amd as w  # noqa: E402

NS = 45000


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 460
    dev = torch.device("cuda", 0)
    L = w.lib()
    L.wspr_session_create.restype = C.c_void_p
    L.wspr_session_create.argtypes = [w.decoder_options]
    L.wspr_session_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.wspr_session_rollover.argtypes = [C.c_void_p]
    L.wspr_session_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.wspr_session_destroy.argtypes = [C.c_void_p]
    nseg, K = 48, 8
    I, Q, _ = bench.synth_batch_gpu(nseg, 5, dev, 2, -14.0, -20.0, 0.5, frac23=0.2)
    Ih, Qh = I.cpu().numpy()[:, :NS].copy(), Q.cpu().numpy()[:, :NS].copy()
    raw = np.random.default_rng(1).integers(100, 156, 1 << 22, dtype=np.uint8)
    out = (w.decoder_results * (nseg * K))()
    n = (C.c_int * nseg)()
    opt, hopt = w.default_options(), w.default_options()
    hopt.usehashtable = 1
    proc = psutil.Process()
    os.chdir(tempfile.mkdtemp(prefix="wspr_leak_", dir="/tmp"))
    marks = {}
    for r in range(rounds + 1):
        if r in (60, rounds):
            torch.cuda.synchronize()
            free, _ = torch.cuda.mem_get_info()
            marks[r] = (proc.memory_info().rss, free, proc.num_threads(), proc.num_fds())
        pi, pq = Ih.ctypes.data_as(C.c_void_p), Qh.ctypes.data_as(C.c_void_p)
        assert L.wspr_decode_batch(pi, pq, nseg, NS, NS, opt, C.addressof(out), K, C.addressof(n), 0) == 0
        assert L.wspr_pin_host_buffer(pi, Ih.nbytes) == 0 and L.wspr_pin_host_buffer(pq, Qh.nbytes) == 0
        assert L.wspr_decode_batch(pi, pq, nseg, NS, NS, opt, C.addressof(out), K, C.addressof(n), 0) == 0
        assert L.wspr_unpin_host_buffer(pi) == 0 and L.wspr_unpin_host_buffer(pq) == 0
        assert L.wspr_decode_batch(pi, pq, nseg, NS, NS, hopt, C.addressof(out), K, C.addressof(n), 0) == 0
        assert L.wspr_decode_batch_device(I.data_ptr(), Q.data_ptr(), nseg, NS, I.stride(0), opt, C.addressof(out), K, C.addressof(n)) == 0
        s = L.wspr_session_create(opt)
        assert L.wspr_session_feed(s, raw.ctypes.data_as(C.c_void_p), raw.size) >= 0
        b = L.wspr_session_rollover(s)
        res = (w.decoder_results * 50)()
        k = C.c_int(0)
        assert L.wspr_session_decode(s, b, res, C.byref(k)) == 0          # too short: the path, not the decode
        L.wspr_session_destroy(s)
    a, b = marks[60], marks[rounds]
    per = rounds - 60
    print("rounds 60 -> %d: host RSS %+.1f MB (%.1f kB per round), free HBM %+.1f MB, threads %d -> %d, open files %d -> %d"
          % (rounds, (b[0] - a[0]) / 1e6, (b[0] - a[0]) / per / 1e3, (b[1] - a[1]) / 1e6, a[2], b[2], a[3], b[3]))
    ok = (b[0] - a[0]) < 64e6 and (a[1] - b[1]) < 64e6 and b[2] <= a[2] + 2 and b[3] <= a[3] + 4
    print("LEAK PROBE", "OK" if ok else "GROWS")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

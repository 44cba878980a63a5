"""Per-candidate parity of the PRODUCTION fine-search kernels (lag scan, frequency scan + rung 0, 43-lag ladder
block, Fano) against the oracle's trace of the reference's candidate loop (wsprd.c:697-822): every candidate the loop
enters, decoded or not.  Used by tests/test_gpu_trace.py in-process and, as a script, in subprocesses that set one of
the library's environment switches (they are read once per process):

    python tests/trace_parity.py parity scenes        -> exit 0 if every field of every visited candidate is equal
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
NS = 45000


def compare_segment(gt, ot, where):
    """gt: product wspr_trace of one segment, ot: oracle Trace.  Returns the number of candidates compared."""
    assert gt.passes_run == ot.passes_run, (where, gt.passes_run, ot.passes_run)
    n = 0
    for p in range(gt.passes_run):
        assert gt.npk[p] == ot.npk[p], (where, p)
        assert gt.n_visited[p] == ot.n_visited[p], (where, p, gt.n_visited[p], ot.n_visited[p])
        for j in range(ot.n_visited[p]):
            g, at = gt.cand[p][j], (where, "pass", p, "cand", j)
            assert g.visited == 1, at
            assert (g.mode0_shift, g.mode0_sync) == (ot.mode0_shift[p][j], ot.mode0_sync[p][j]), at       # lag scan
            f = ot.cand_fine[p][j]
            assert (g.freq, g.shift, g.drift, g.sync) == (f.freq, f.shift, f.drift, f.sync), at           # frequency scan
            assert (g.attempts, g.fano_calls) == (ot.attempts[p][j], ot.fano_calls[p][j]), at             # ladder walk
            assert (g.first_sync, g.first_rms) == (ot.first_sync2[p][j], ot.first_rms[p][j]), at          # rung 0
            assert bytes(g.first_symbols) == bytes(ot.first_symbols[p][j]), at
            assert (g.decoded, g.subtracted) == (ot.decoded[p][j], ot.subtracted[p][j]), at
            if g.decoded:
                assert g.cycles == ot.fano_cycles[p][j] and bytes(g.decdata) == bytes(ot.decdata[p][j]), at
            n += 1
        for j in range(ot.n_visited[p], 200):
            assert gt.cand[p][j].visited == 0, (where, p, j)
    return n


def check(I, Q, w, ol, opts=None, name=""):
    opts = opts or {}
    spots, tr = w.wspr_decode_batch_trace(I, Q, w.default_options(**opts), max_results=32)
    total = undecoded = 0
    for s in range(I.shape[0]):
        ref, _, _, ot = ol.decode(I[s], Q[s], NS, ol.default_options(**opts), trace=True)
        total += compare_segment(tr[s], ot, (name, "segment", s))
        undecoded += sum(1 for p in range(ot.passes_run) for j in range(ot.n_visited[p]) if not ot.decoded[p][j])
        assert [x.message for x in spots[s]] == [x.message for x in ref], (name, s)
    return total, undecoded


def parity_batch():
    import oracle_lib as ol
    import synth
    symf = lambda m: ol.channel_symbols(m)[1]
    segs = [synth.make_segment(1000 + s, symf, snr_db=-20.0) for s in range(6)]
    segs.append(synth.make_segment(77, symf, n_signals=4, snr_db=-8.0, snr_span=12.0, t_jitter=0.3))
    segs.append(synth.make_segment(78, symf, snr_db=-15.0, drift=2.0))
    return np.stack([s[0] for s in segs]), np.stack([s[1] for s in segs])


def main(argv):
    import torch  # noqa: F401  (first: see tests/conftest.py)
    import oracle_lib as ol
    import rtlsdr_wsprd_amd as w
    assert w.lib().wspr_device_ready() == 1
    for what in argv:
        if what == "parity":
            I, Q = parity_batch()
            for o in (dict(), dict(quickmode=1), dict(subtraction=0), dict(npasses=3)):
                print(what, o, check(I, Q, w, ol, o, what), flush=True)
        elif what == "scenes":
            from test_gpu_parity import random_scenes
            I, Q = random_scenes(int(os.environ.get("WSPR_TRACE_SCENES", "40")))
            print(what, check(I, Q, w, ol, None, what), flush=True)
        elif what == "config3":
            import bench
            dev = torch.device("cuda", 0)
            torch.cuda.set_device(0)
            n = int(os.environ.get("WSPR_TRACE_CONFIG3", "64"))
            I, Q, _ = bench.synth_batch_gpu(n, 4321, dev, 10, -10.0, -28.0, 0.3)
            print(what, check(I.cpu().numpy(), Q.cpu().numpy(), w, ol, None, what), flush=True)
        else:
            raise SystemExit("unknown set " + what)
    print("TRACE PARITY OK")


if __name__ == "__main__":
    main(sys.argv[1:] or ["parity"])

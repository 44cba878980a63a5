"""usehashtable (-H) on a BATCH, decoded in parallel (reference wsprd.c:481-494, 842-852,
wsprd_utils.c:264-311, wsprsim_utils.c:280-300).  The hash memory orders the segments; the product decodes them in parallel
against a logged, versioned view of that memory and decodes again only the segments whose look-ups would have seen
something else (wspr_pipeline.h, HashBatch).  Here: spots AND hashtable.txt of one batched call equal those of the oracle
called segment by segment in index order, and of the product called segment by segment, on batches with 0 / 5 / 50 %
type-2/3 traffic; the sharded form of the call (wspr_decode_batch_hashed with other shards' stores + wspr_hash_commit)
gives the same again."""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle_lib as ol
import synth

pytestmark = pytest.mark.gpu
NS = 45000

STATIONS = synth.STATIONS


@pytest.fixture(scope="module")
def w():
    import rtlsdr_wsprd_amd as mod
    assert mod.lib().wspr_device_ready() == 1
    return mod


def _symbols(msg):
    ok, s = ol.channel_symbols(msg)
    assert ok, msg
    return s


def _traffic(nseg, frac23, seed, nsig=3, snr=-9.0):
    """nseg segments of nsig signals; a signal is, with probability frac23, one of the compound-call stations, which
    alternate between their type-2 and type-3 transmissions from segment to segment (as real stations do from slot to
    slot) -- so a type-3 "<call>" usually resolves only through what an EARLIER segment of the batch stored."""
    rng = np.random.default_rng(seed)
    sigma = np.sqrt((375.0 / 2500.0) / 2.0)
    I = np.empty((nseg, NS), np.float32); Q = np.empty((nseg, NS), np.float32)
    texts = []
    for s in range(nseg):
        i = rng.normal(0, sigma, NS); q = rng.normal(0, sigma, NS)
        msgs = []
        for k in range(nsig):
            if rng.random() < frac23:
                m = synth.station_message(int(rng.integers(0, len(STATIONS))), s)
            else:
                m = synth.message_for(int(rng.integers(0, 1 << 20)))
            msgs.append(m)
            si, sq = synth.tone_signal(_symbols(m), -90.0 + 180.0 * k / max(1, nsig - 1) + rng.uniform(-3, 3),
                                       2.0 + rng.uniform(-0.3, 0.3), 10.0 ** ((snr - 1.5 * k) / 20.0))
            i += si; q += sq
        I[s], Q[s] = synth.normalise(i.astype(np.float32), q.astype(np.float32))
        texts.append(msgs)
    return I, Q, texts


def _tup(x):
    return (x.message, x.call, x.loc, x.pwr, x.cycles, x.jitter, x.drift, x.sync, x.snr, x.dt, x.freq)


def _opt(mod, use):
    o = mod.default_options()
    o.usehashtable = use
    return o


def _in_dir(d, fn):
    cwd = os.getcwd()
    d.mkdir()
    os.chdir(d)
    try:
        out = fn()
        txt = open("hashtable.txt").read() if os.path.exists("hashtable.txt") else ""
        return out, txt
    finally:
        os.chdir(cwd)


def _hashed(w, I, Q, seg0=0, prior=None, flags=0, out=None, nres=None, K=16):
    """wspr_decode_batch_hashed through ctypes; returns (spots per segment, stores, n_redecoded, out, nres)."""
    L = w.lib()
    nseg = I.shape[0]
    out = out if out is not None else (w.decoder_results * (nseg * K))()
    nres = nres if nres is not None else (C.c_int * nseg)()
    cap = 64 * nseg + 64
    stores = np.zeros((cap, 32), np.uint8)
    n_st = C.c_int(0); n_re = C.c_int(0)
    pr = np.ascontiguousarray(prior if prior is not None else np.zeros((0, 32), np.uint8))
    L.wspr_decode_batch_hashed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, w.decoder_options, C.c_void_p,
                                           C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                           C.c_int, C.c_void_p, C.c_void_p]
    rc = L.wspr_decode_batch_hashed(ol.ptr(I), ol.ptr(Q), nseg, NS, NS, _opt(w, 1), C.addressof(out), K, C.addressof(nres), 0,
                                    seg0, ol.ptr(pr) if len(pr) else None, len(pr), flags, ol.ptr(stores), cap,
                                    C.byref(n_st), C.byref(n_re))
    assert rc == 0, rc
    spots = [[_tup(out[s * K + i]) for i in range(nres[s])] for s in range(nseg)]
    return spots, stores[:n_st.value].copy(), n_re.value, out, nres


@pytest.mark.parametrize("frac23,nseg", [(0.0, 48), (0.05, 192), (0.5, 160)])
def test_batch_with_hashtable_equals_the_serial_walk(w, tmp_path, frac23, nseg):
    if frac23 == 0.05:                                              # a longer soak: WSPR_HASH_SEGMENTS=1536
        nseg = int(os.environ.get("WSPR_HASH_SEGMENTS", nseg))
    I, Q, texts = _traffic(nseg, frac23, 1000 + int(frac23 * 100) + int(os.environ.get("WSPR_HASH_SEED", "0")))
    n23 = sum(m.startswith("<") or "/" in m for seg in texts for m in seg)

    def batch():
        got = w.wspr_decode_batch(I, Q, _opt(w, 1), max_results=16)
        return [[_tup(x) for x in g] for g in got]

    def singles():
        return [[_tup(x) for x in w.wspr_decode(I[s], Q[s], NS, _opt(w, 1))[0]] for s in range(nseg)]

    def oracle():
        o = ol.default_options()
        o.usehashtable = 1
        return [[_tup(x) for x in ol.decode(I[s], Q[s], NS, o)[0]] for s in range(nseg)]

    b, bf = _in_dir(tmp_path / "batch", batch)
    r, rf = _in_dir(tmp_path / "oracle", oracle)
    strip = lambda res: [[t[:8] + t[9:] for t in seg] for seg in res]           # snr: host libm both, compared to 1e-4 below
    assert strip(b) == strip(r)
    assert all(abs(x[8] - y[8]) < 1e-4 for sb, sr in zip(b, r) for x, y in zip(sb, sr))
    assert bf == rf
    if nseg <= 64 or (frac23 == 0.05 and nseg <= 256):
        s_, sf = _in_dir(tmp_path / "singles", singles)
        assert s_ == b and sf == bf
    msgs = [m.decode() for seg in b for (m, *_) in seg]
    resolved = sum(m.startswith("<") and not m.startswith("<...>") for m in msgs)
    unresolved = sum(m.startswith("<...>") for m in msgs)
    # how much of the batch the ordered memory really touched: the direct call reports the segments decoded twice
    (_, stores, n_re, _, _), hf = _in_dir(tmp_path / "hashed", lambda: _hashed(w, I, Q))
    assert hf == bf
    print("frac23 %.2f: %d segments, %d type-2/3 signals sent, %d resolved / %d unresolved type-3 spots, %d hash stores, "
          "%d segments decoded again" % (frac23, nseg, n23, resolved, unresolved, len(stores), n_re))
    if frac23 == 0.0:
        assert n_re == 0 and resolved == 0
    else:
        assert resolved > 0 and n_re > 0
    # a second pass over the same traffic in the SAME directory starts from the file the first one wrote: every look-up
    # is answered by the file as its predecessors would answer it, nothing is decoded twice
    if frac23 == 0.05:
        def twice():
            _hashed(w, I, Q)
            return _hashed(w, I, Q)
        (sp2, _, n_re2, _, _), _ = _in_dir(tmp_path / "twice", twice)
        assert n_re2 == 0
        # (a station heard only in its hashed form stays "<...>": never more of those than in the first pass)
        assert sum(m.startswith(b"<...>") for seg in sp2 for (m, *_) in seg) <= unresolved


def test_sharded_hashed_calls_equal_one_batch(w, tmp_path):
    """The protocol of a job sharded over several processes (include/wspr_mi355x.h, wspr_decode_batch_hashed), here with
    three shards on three threads of one process: every shard decodes with no knowledge of the others, the stores are
    exchanged, shards whose predecessors' stores changed revisit, one commit writes the file.  Same spots, same file."""
    nseg = 150
    I, Q, _ = _traffic(nseg, 0.3, 77)
    whole, wf = _in_dir(tmp_path / "whole", lambda: [[_tup(x) for x in g] for g in w.wspr_decode_batch(I, Q, _opt(w, 1), max_results=16)])
    L = w.lib()
    bounds = [(0, 50), (50, 100), (100, 150)]
    ranks = [ThreadPoolExecutor(1) for _ in bounds]
    for k, ex in enumerate(ranks):
        ex.submit(L.wspr_bind_thread_lane, 4 + k).result()

    def job():
        state = [None] * 3
        stores = [np.zeros((0, 32), np.uint8)] * 3
        seen_prior = [b""] * 3
        rounds = 0
        while True:
            rounds += 1
            changed = False
            new_stores = list(stores)
            for r, (lo, hi) in enumerate(bounds):
                prior = np.concatenate(stores[:r]) if r else np.zeros((0, 32), np.uint8)
                if state[r] is not None and prior.tobytes() == seen_prior[r]:
                    continue
                flags = 1 | (2 if state[r] is not None else 0)           # KEEP_FILE | REVISIT
                out, nres = (state[r][3], state[r][4]) if state[r] is not None else (None, None)
                state[r] = ranks[r].submit(_hashed, w, I[lo:hi], Q[lo:hi], lo, prior, flags, out, nres).result()
                seen_prior[r] = prior.tobytes()
                if state[r][1].tobytes() != stores[r].tobytes():
                    changed = True
                new_stores[r] = state[r][1]
            stores = new_stores
            if not changed:
                break
            assert rounds <= 4
        allst = np.ascontiguousarray(np.concatenate(stores))
        L.wspr_hash_commit.argtypes = [C.c_void_p, C.c_int]
        assert L.wspr_hash_commit(ol.ptr(allst), len(allst)) == 0
        return [sp for st in state for sp in st[0]], rounds
    (sharded, rounds), sf = _in_dir(tmp_path / "sharded", job)
    print("sharded -H: %d rounds" % rounds)
    assert sharded == whole and sf == wf and rounds >= 2


def test_hashed_calls_in_flight_keep_their_order(w, tmp_path):
    """Calls with the option are ordered by definition (each reads the file the previous one wrote): their order is the
    order in which they enter the library.  Since round 5 they need not wait for each other to START: a call decodes its
    first round ahead of its turn, beside the calls before it on other lanes, then waits, takes the file they wrote as its
    base and decodes again what that changes.  Four batches of 128 segments on four lanes, submitted 6 ms apart (each
    takes longer than that, so they overlap): spots and hashtable.txt equal the oracle walking all 512 segments in order."""
    import time
    nb, per = 4, 128
    I, Q, _ = _traffic(nb * per, 0.3, 4242)
    L = w.lib()
    lanes = [ThreadPoolExecutor(1) for _ in range(nb)]
    for k, ex in enumerate(lanes):
        ex.submit(L.wspr_bind_thread_lane, 8 + k).result()

    def one(k):
        t0 = time.perf_counter()
        got = w.wspr_decode_batch(I[k * per:(k + 1) * per], Q[k * per:(k + 1) * per], _opt(w, 1), max_results=16)
        return [[_tup(x) for x in g] for g in got], t0, time.perf_counter()

    def job():
        futs = []
        for k in range(nb):
            futs.append(lanes[k].submit(one, k))
            time.sleep(0.006)
        res = [f.result() for f in futs]
        overlap = sum(1 for k in range(1, nb) if res[k][1] < res[k - 1][2])     # started before its predecessor ended
        return [sp for r in res for sp in r[0]], overlap

    def oracle():
        o = ol.default_options()
        o.usehashtable = 1
        return [[_tup(x) for x in ol.decode(I[s], Q[s], NS, o)[0]] for s in range(nb * per)]
    (got, overlap), gf = _in_dir(tmp_path / "flight", job)
    ref, rf = _in_dir(tmp_path / "oracle", oracle)
    strip = lambda res: [[t[:8] + t[9:] for t in seg] for seg in res]
    assert strip(got) == strip(ref) and gf == rf
    print("calls in flight: %d of %d started before their predecessor had returned" % (overlap, nb - 1))
    assert overlap >= 1


def test_store_buffer_too_small_commits_nothing_and_is_completed_by_a_revisit(w, tmp_path):
    """cap smaller than the stores the call logs: -3, *n_stores = the size needed, hashtable.txt NOT written (the call
    failed before its commit), results in place; the same call with WSPR_HASH_REVISIT and a buffer of that size returns
    the stores and writes the file -- equal to the one-call form."""
    nseg, K = 40, 16
    I, Q, _ = _traffic(nseg, 0.3, 991)
    whole, wf = _in_dir(tmp_path / "whole", lambda: [[_tup(x) for x in g] for g in w.wspr_decode_batch(I, Q, _opt(w, 1), max_results=K)])
    L = w.lib()
    L.wspr_decode_batch_hashed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, w.decoder_options, C.c_void_p,
                                           C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                           C.c_int, C.c_void_p, C.c_void_p]

    def job():
        out = (w.decoder_results * (nseg * K))(); nres = (C.c_int * nseg)()
        small = np.zeros((2, 32), np.uint8); n_st = C.c_int(0)
        rc = L.wspr_decode_batch_hashed(ol.ptr(I), ol.ptr(Q), nseg, NS, NS, _opt(w, 1), C.addressof(out), K, C.addressof(nres), 0,
                                        0, None, 0, 0, ol.ptr(small), 2, C.byref(n_st), None)
        assert rc == -3 and n_st.value > 2
        assert not os.path.exists("hashtable.txt")
        first = [[_tup(out[s * K + i]) for i in range(nres[s])] for s in range(nseg)]
        big = np.zeros((n_st.value, 32), np.uint8); n2 = C.c_int(0); n_re = C.c_int(-1)
        rc = L.wspr_decode_batch_hashed(ol.ptr(I), ol.ptr(Q), nseg, NS, NS, _opt(w, 1), C.addressof(out), K, C.addressof(nres), 0,
                                        0, None, 0, 2, ol.ptr(big), n_st.value, C.byref(n2), C.byref(n_re))
        assert rc == 0 and n2.value == n_st.value and n_re.value == 0
        assert big[:, :4].view(np.int32).ravel().tolist() == sorted(big[:, :4].view(np.int32).ravel().tolist())
        return first, [[_tup(out[s * K + i]) for i in range(nres[s])] for s in range(nseg)]
    (first, second), f = _in_dir(tmp_path / "retry", job)
    assert first == whole and second == whole and f == wf


def test_revisit_is_refused_when_its_state_is_gone(w, tmp_path):
    """WSPR_HASH_REVISIT works on what the calling thread's previous hashed call left (its log, the decoded rows in the
    library's working buffers).  Without such a call, after wspr_release_buffers(), or with another slot layout, it must
    fail with a negative code -- not write through a null or too small buffer (advisor, round 5) -- and a fresh call on
    the same thread works again."""
    nseg, K = 24, 16
    I, Q, _ = _traffic(nseg, 0.3, 4711)
    L = w.lib()
    L.wspr_decode_batch_hashed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, w.decoder_options, C.c_void_p,
                                           C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                           C.c_int, C.c_void_p, C.c_void_p]
    L.wspr_release_buffers.restype = C.c_size_t
    ex = ThreadPoolExecutor(1)
    ex.submit(L.wspr_bind_thread_lane, 5).result()

    out = (w.decoder_results * (nseg * K))(); nres = (C.c_int * nseg)()      # a revisit rewrites only what it decodes again

    def call(flags, prior=None):
        st = np.zeros((64 * nseg, 32), np.uint8); n_st = C.c_int(0)
        pr = np.ascontiguousarray(prior if prior is not None else np.zeros((0, 32), np.uint8))
        rc = L.wspr_decode_batch_hashed(ol.ptr(I), ol.ptr(Q), nseg, NS, NS, _opt(w, 1), C.addressof(out), K, C.addressof(nres), 0,
                                        100, ol.ptr(pr) if len(pr) else None, len(pr), flags, ol.ptr(st), len(st), C.byref(n_st), None)
        return rc, [[_tup(out[s * K + i]) for i in range(nres[s])] for s in range(nseg)]

    def job():
        assert ex.submit(call, 1 | 2).result()[0] < 0                 # no previous call on this thread
        rc, first = ex.submit(call, 1).result()
        assert rc == 0
        rc, again = ex.submit(call, 1 | 2).result()                   # a legitimate revisit: nothing changed, same spots
        assert rc == 0 and again == first
        assert L.wspr_release_buffers() > 0                           # the rows are gone ...
        fake = np.zeros((1, 32), np.uint8); fake[0, :12] = np.frombuffer(np.array([0, 77, 2], np.int32).tobytes(), np.uint8)
        fake[0, 12:16] = np.frombuffer(b"ZZ9Z", np.uint8)
        rc, _ = ex.submit(call, 1 | 2, fake).result()                 # ... a revisit that has nothing to decode again may pass,
        rc2, _ = ex.submit(call, 1 | 2).result()                      # but none may crash; a fresh call works
        assert rc <= 0 and rc2 <= 0
        rc, fresh = ex.submit(call, 1).result()
        assert rc == 0 and fresh == first
        return True
    assert _in_dir(tmp_path / "revisit", job)[0]

"""N>1 path on CPU: world_size-2 gloo run of the fan-out / gather helpers bench.py uses.
Each rank decodes its shard of a small synthetic set with the CPU oracle (the checker; no GPU
here) and packs spot records exactly as the GPU path does; rank 0 gathers and compares with the
single-process answer."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NSEG, K = 6, 4


def _segments():
    import oracle_lib as ol
    import synth
    symf = lambda m: ol.channel_symbols(m)[1]
    return [synth.make_segment(500 + s, symf, snr_db=-14.0) for s in range(NSEG)]


def _decode_shard(lo, hi, segs, opt, product=False):
    import oracle_lib as ol
    import rtlsdr_wsprd_amd as w
    n = hi - lo
    out = (w.decoder_results * (n * K))()
    cnt = (C.c_int * n)()
    if product:          # the HIP library itself (GPU box): one batch call for the rank's shard
        I = np.stack([segs[s][0] for s in range(lo, hi)]); Q = np.stack([segs[s][1] for s in range(lo, hi)])
        rc = w.lib().wspr_decode_batch(I.ctypes.data_as(C.c_void_p), Q.ctypes.data_as(C.c_void_p), n, 45000, 45000, opt,
                                       C.addressof(out), K, C.addressof(cnt), 0)
        assert rc == 0
        return out, cnt, n
    o = ol.Options.from_buffer_copy(bytes(opt))
    for i, s in enumerate(range(lo, hi)):
        spots, _, _ = ol.decode(segs[s][0], segs[s][1], 45000, o)
        cnt[i] = min(len(spots), K)
        for k in range(cnt[i]):
            C.memmove(C.addressof(out) + (i * K + k) * 80, C.addressof(spots[k]), 80)
    return out, cnt, n


def _worker(rank, world, port, q, product=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rtlsdr_wsprd_amd as w
    from rtlsdr_wsprd_amd import dist as wd
    segs = _segments()
    opt = w.default_options(npasses=2 if rank == 0 else 1)     # only rank 0 holds the real options
    opt = wd.broadcast_options(opt, src=0)
    assert opt.npasses == 2 and opt.subtraction == 1 and opt.freq == 144489000
    lo, hi = wd.shard_range(NSEG, rank, world)
    out, cnt, n = _decode_shard(lo, hi, segs, opt, product)
    g = wd.gather_spots(wd.pack_spots(out, cnt, n, K, 80), dst=0)
    g2 = wd.SpotGatherer(out, cnt, n, K, 80, dst=0).gather()        # the preallocated path bench.py uses
    if rank == 0:
        assert g2[0].tolist() == wd.unpack_counts(g).tolist()
        assert bytes(g2[1].numpy().tobytes()) == bytes(g[:, :, 4:].contiguous().numpy().tobytes())
        q.put((wd.unpack_counts(g).tolist(), wd.unpack_messages(g, K, 80)))
    else:
        assert g is None and g2 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_world2_product_shards_equal_single_process_oracle():
    """The same two-rank run on the GPU box, every rank decoding its shard THROUGH THE PRODUCT (both ranks
    share the one GPU; the collectives stay on gloo): the gathered spots equal the single-process oracle's."""
    _run_world2(product=True)


def test_world2_gather_equals_single_process():
    _run_world2(product=False)


def _run_world2(product):
    sys.path.insert(0, ROOT)
    import rtlsdr_wsprd_amd as w
    from rtlsdr_wsprd_amd import dist as wd
    assert [wd.shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port + (7 if product else 0), q, product)) for r in range(2)]
    for p in procs:
        p.start()
    counts, msgs = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    segs = _segments()
    out, cnt, n = _decode_shard(0, NSEG, segs, w.default_options())
    single = [[bytes(out[i * K + k].message).split(b"\0")[0].decode() for k in range(cnt[i])] for i in range(NSEG)]
    flat = [m for r in msgs for m in r]
    assert flat == single
    assert sum(map(sum, counts)) == sum(cnt) and sum(cnt) >= NSEG - 1


# ---- usehashtable across ranks: the order-dependent hash memory -------------------------------
_HT_MSGS = [["PJ4/K1ABC 37"], ["W1AW FN31 10"], ["<PJ4/K1ABC> FK52UD 37", "W1AW FN31 10"], ["<PJ4/K1ABC> FK52UD 37"]]


def _ht_msgs18():
    """Eighteen segments for eight ranks (shards of 3, 3, 2, 2, ...): every segment carries a plain type-1 signal and the
    transmissions of two compound-call stations, which alternate between their type-2 and type-3 forms from segment to
    segment -- a "<call>" resolves only through what an EARLIER segment (usually another rank's) stored."""
    import synth
    return [[synth.message_for(1000 + 37 * s), synth.station_message(s % 8, s), synth.station_message((s + 3) % 8, s)]
            for s in range(18)]


def _ht_segments(msgs_by_segment=None):
    import oracle_lib as ol
    import synth
    symf = lambda m: ol.channel_symbols(m)[1]
    segs = []
    for k, msgs in enumerate(_HT_MSGS if msgs_by_segment is None else msgs_by_segment):
        rng = np.random.default_rng(7100 + k)
        sigma = np.sqrt((375.0 / 2500.0) / 2.0)
        I = rng.normal(0, sigma, synth.NS); Q = rng.normal(0, sigma, synth.NS)
        for j, m in enumerate(msgs):
            si, sq = synth.tone_signal(symf(m), -40.0 + 60.0 * j, 2.0, 10.0 ** (-8.0 / 20.0))
            I += si; Q += sq
        segs.append(synth.normalise(I.astype(np.float32), Q.astype(np.float32)))
    return segs


def _ht_decode(segs, lo, hi, product):
    """Segments lo..hi-1 one after the other with usehashtable = 1 (hashtable.txt in the working directory)."""
    import oracle_lib as ol
    import rtlsdr_wsprd_amd as w
    out = []
    for s in range(lo, hi):
        if product:
            po = w.default_options(); po.usehashtable = 1
            spots, _, _ = w.wspr_decode(segs[s][0], segs[s][1], 45000, po)
        else:
            o = ol.default_options(); o.usehashtable = 1
            spots, _, _ = ol.decode(segs[s][0], segs[s][1], 45000, o)
        out.append(sorted(x.message.split(b"\0")[0].decode() for x in spots))
    return out


def _ht_worker(rank, world, port, q, workdir, product):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.chdir(workdir)                                            # the ranks share hashtable.txt
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtlsdr_wsprd_amd import dist as wd
    segs = _ht_segments()
    lo, hi = wd.shard_range(len(segs), rank, world)
    mine = wd.in_rank_order(lambda: _ht_decode(segs, lo, hi, product))
    allm = [None] * world
    dist.all_gather_object(allm, mine)
    if rank == 0:
        q.put([m for part in allm for m in part])
    dist.barrier()
    dist.destroy_process_group()


def _run_hashtable_world2(tmp_path, product):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    shared = tmp_path / "ranks"; shared.mkdir()
    port = 31500 + os.getpid() % 2000 + (11 if product else 0)
    procs = [ctx.Process(target=_ht_worker, args=(r, 2, port, q, str(shared), product)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # one process walking the four segments in order, in a directory of its own (the oracle: the checker)
    alone = tmp_path / "alone"; alone.mkdir()
    cwd = os.getcwd()
    try:
        os.chdir(alone)
        ref = _ht_decode(_ht_segments(), 0, len(_HT_MSGS), False)
        ref_file = open("hashtable.txt").read()
    finally:
        os.chdir(cwd)
    assert got == ref
    assert open(shared / "hashtable.txt").read() == ref_file
    # the hashed call of the last two segments (rank 1's shard) is resolved by what rank 0 heard
    assert got[0] == ["PJ4/K1ABC 37"] and "<PJ4/K1ABC> FK52UD 37" in got[2] and got[3] == ["<PJ4/K1ABC> FK52UD 37"]


def test_world2_hashtable_segments_in_rank_order(tmp_path):
    """usehashtable = 1 over two ranks (gloo, the oracle decoding): with in_rank_order() over contiguous shards the
    ranks produce the spots and the hashtable.txt of ONE process walking all segments in index order."""
    _run_hashtable_world2(tmp_path, product=False)


@pytest.mark.gpu
def test_world2_hashtable_segments_in_rank_order_through_the_product(tmp_path):
    _run_hashtable_world2(tmp_path, product=True)


# ---- usehashtable over ranks without turns: hashed_rounds() ------------------------------------------------------
def _op(seg, slot, kind, call):
    a = np.zeros(32, np.uint8)
    a[:12] = np.frombuffer(np.array([seg, slot, kind], np.int32).tobytes(), np.uint8)
    b = call.encode()[:12]
    a[12:12 + len(b)] = np.frombuffer(b, np.uint8)
    return a


def _script():
    """A toy job of 9 'segments' over 3 ranks: each segment is a list of steps, ('put', slot, call) or ('get', slot).
    A segment's output is what its gets see; if a get sees 'ZZ' the segment ALSO stores into slot 7 (its behaviour
    depends on the look-up, as a decode's subtraction does)."""
    return [[("put", 1, "AA")], [("get", 1)], [("put", 2, "BB")],
            [("get", 2), ("put", 3, "ZZ")], [("get", 3)], [("get", 1), ("put", 1, "CC")],
            [("get", 3), ("get", 7)], [("get", 1)], [("put", 2, "DD"), ("get", 2)]]


def _run_segment(steps, table):
    """-> (outputs, stores [(slot, call)]); table: slot -> call as the segment finds it"""
    own, outs, stores = {}, [], []
    for st in steps:
        if st[0] == "put":
            own[st[1]] = st[2]; stores.append((st[1], st[2]))
        else:
            v = own.get(st[1], table.get(st[1], ""))
            outs.append(v)
            if v == "ZZ":
                own[7] = "QQ"; stores.append((7, "QQ"))
    return outs, stores


def _hr_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtlsdr_wsprd_amd import dist as wd
    script = _script()
    lo, hi = wd.shard_range(len(script), rank, world)
    result = {}
    calls = []

    def decode_shard(prior, revisit):
        calls.append(revisit)
        table = {}
        for row in prior:                                   # the lower ranks' stores, in segment order
            seg, slot, kind = np.frombuffer(row[:12].tobytes(), np.int32)
            table[int(slot)] = bytes(row[12:25]).split(b"\0")[0].decode()
        ops = []
        for s in range(lo, hi):
            outs, stores = _run_segment(script[s], table)
            result[s] = outs
            for slot, call in stores:
                table[slot] = call
                ops.append(_op(s, slot, 2, call))
        return np.stack(ops) if ops else np.zeros((0, 32), np.uint8)

    committed = []
    rounds = wd.hashed_rounds(decode_shard, lambda allst: committed.append(allst.copy()))
    every = [None] * world
    dist.all_gather_object(every, (result, rounds, calls))
    if rank == 0:
        q.put((every, [(int(np.frombuffer(r[:4].tobytes(), np.int32)[0]), int(np.frombuffer(r[4:8].tobytes(), np.int32)[0]),
                        bytes(r[12:25]).split(b"\0")[0].decode()) for r in committed[0]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [3, 8])
def test_world3_hashed_rounds_reach_the_serial_result(world):
    """hashed_rounds() on CPU with a toy decoder whose behaviour depends on its look-ups: three gloo ranks (and eight: the
    size of the node configs[3] is quoted on -- nine segments over eight ranks, shards of one or two), everybody
    decodes at once, the stores are exchanged until they stop changing -- outputs and the committed store list are
    those of one process walking the segments in order; rank 0 never revisits, nobody needs more than world + 1 rounds."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_hr_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    every, committed = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    script = _script()
    table, want, want_stores = {}, {}, []
    for s, steps in enumerate(script):
        outs, stores = _run_segment(steps, table)
        want[s] = outs
        for slot, call in stores:
            table[slot] = call
            want_stores.append((s, slot, call))
    got = {}
    for result, rounds, calls in every:
        got.update(result)
    assert got == want
    assert committed == want_stores
    assert want[4] == ["ZZ"] and want[6] == ["ZZ", "QQ"] and want[7] == ["CC"]      # resolved across rank boundaries
    assert every[0][2] == [False] and all(r[1] <= world + 1 for r in every)


def _hr_fail_worker(rank, world, port, q, when):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtlsdr_wsprd_amd import dist as wd

    def decode_shard(prior, revisit):
        if rank == 1 and revisit == when:
            raise RuntimeError("the GPU of rank 1 has gone")
        return np.stack([_op(rank, 1 + rank, 2, "R%d" % rank)])   # every rank stores something: rank 1 and 2 must revisit
    try:
        wd.hashed_rounds(decode_shard, lambda allst: q.put(("committed", rank)))
        q.put(("returned", rank))
    except RuntimeError as e:
        q.put(("raised", rank, str(e)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("when", [False, True])
def test_world3_hashed_rounds_a_failing_shard_stops_every_rank(when):
    """A rank whose decode fails (first round or a revisit) used to leave hashed_rounds() and the others waited in the
    exchange for ever; now the failure travels with the exchange: every rank raises, nothing is committed, nobody hangs."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000 + (7 if when else 0)
    procs = [ctx.Process(target=_hr_fail_worker, args=(r, 3, port, q, when)) for r in range(3)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[0] for g in got] == ["raised"] * 3 and sorted(g[1] for g in got) == [0, 1, 2]
    assert all("rank 1" in g[2] and "has gone" in g[2] for g in got)


def _hs_worker(rank, world, port, q, workdir, store_cap=None, eighteen=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.chdir(workdir)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rtlsdr_wsprd_amd as w
    from rtlsdr_wsprd_amd import dist as wd
    segs = _ht_segments(_ht_msgs18() if eighteen else None)
    lo, hi = wd.shard_range(len(segs), rank, world)
    I = np.stack([segs[s][0] for s in range(lo, hi)]); Q = np.stack([segs[s][1] for s in range(lo, hi)])
    out, cnt, rounds = wd.decode_batch_hashed_sharded(I, Q, len(segs), w.default_options(), max_results=8, store_cap=store_cap)
    mine = [sorted(out[i * 8 + k].message.split(b"\0")[0].decode() for k in range(cnt[i])) for i in range(hi - lo)]
    allm = [None] * world
    dist.all_gather_object(allm, (mine, rounds))
    if rank == 0:
        q.put(([m for part in allm for m in part[0]], [part[1] for part in allm]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("store_cap", [None, 1])
def test_world2_hashtable_without_turns_through_the_product(tmp_path, store_cap):
    """decode_batch_hashed_sharded(): two gloo ranks decode their shards AT ONCE with usehashtable (no in_rank_order
    turns); spots and hashtable.txt equal those of the oracle walking the four segments in order.  store_cap = 1: the
    store buffer is too small on the first call of every round (-3 from the library, nothing committed) and the call is
    completed with WSPR_HASH_REVISIT and the size the library asked for."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    shared = tmp_path / "ranks"; shared.mkdir()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_hs_worker, args=(r, 2, port, q, str(shared), store_cap)) for r in range(2)]
    for p in procs:
        p.start()
    got, rounds = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    alone = tmp_path / "alone"; alone.mkdir()
    cwd = os.getcwd()
    try:
        os.chdir(alone)
        ref = _ht_decode(_ht_segments(), 0, len(_HT_MSGS), False)
        ref_file = open("hashtable.txt").read()
    finally:
        os.chdir(cwd)
    assert got == ref and open(shared / "hashtable.txt").read() == ref_file
    # (rank 1 revisits once it has seen rank 0's stores; its own stores do not change -- a type-3 decode stores nothing --
    # so the exchange may already be over after the first round)
    assert got[3] == ["<PJ4/K1ABC> FK52UD 37"] and all(1 <= r <= 3 for r in rounds)


@pytest.mark.gpu
def test_world8_hashtable_without_turns_through_the_product(tmp_path):
    """The same at the world size configs[3] is quoted on: EIGHT gloo ranks (sharing the one GPU of this box) decode their
    shards of eighteen segments at once with usehashtable; spots and hashtable.txt equal those of the oracle walking the
    eighteen segments in order, and type-3 calls resolve across rank boundaries."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    shared = tmp_path / "ranks"; shared.mkdir()
    port = 36500 + os.getpid() % 2000
    procs = [ctx.Process(target=_hs_worker, args=(r, 8, port, q, str(shared), None, True)) for r in range(8)]
    for p in procs:
        p.start()
    got, rounds = q.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    alone = tmp_path / "alone"; alone.mkdir()
    cwd = os.getcwd()
    segs = _ht_segments(_ht_msgs18())
    try:
        os.chdir(alone)
        ref = _ht_decode(segs, 0, len(segs), False)
        ref_file = open("hashtable.txt").read()
    finally:
        os.chdir(cwd)
    assert got == ref and open(shared / "hashtable.txt").read() == ref_file
    resolved = sum(m.startswith("<") and not m.startswith("<...>") for seg in got for m in seg)
    print("world 8 -H: rounds per rank %r, %d resolved type-3 spots of %d spots" % (rounds, resolved, sum(len(g) for g in got)))
    # (a rank revisits once it has seen its predecessors' stores, but a type-3 decode stores nothing: the store lists do not
    # change and the exchange is over after the first round)
    assert resolved >= 6 and len(rounds) == 8 and all(1 <= r <= 9 for r in rounds)


# ---- real-input fan-out: rank 0 holds the IQ, the other ranks receive their rows -----------------
def _root_worker(rank, world, port, q, product, nseg):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rtlsdr_wsprd_amd as w
    from rtlsdr_wsprd_amd import dist as wd
    if rank == 0:                                                # ONLY the root has the segments and the options
        segs = _segments()[:nseg]
        I = torch.from_numpy(np.stack([s[0] for s in segs])); Q = torch.from_numpy(np.stack([s[1] for s in segs]))
        opt = w.default_options(npasses=2)
    else:
        I = Q = None
        opt = w.default_options(npasses=1, subtraction=0)

    def decode_shard(mi, mq, o):
        rows = [(mi[k].numpy(), mq[k].numpy()) for k in range(mi.shape[0])]
        out, cnt, _ = _decode_shard(0, len(rows), rows, o, product)
        return out, cnt

    lo, hi = wd.shard_range(nseg, rank, world)
    res = wd.decode_from_root(I, Q, nseg, 45000, opt, decode_shard, max_results=K, record_size=80, root=0)
    if rank == 0:
        cnt, rec = res
        msgs = [[bytes(rec[s, k * 80 + 28:k * 80 + 51]).split(b"\0")[0].decode() for k in range(cnt[s])] for s in range(nseg)]
        q.put((cnt.tolist(), msgs))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def _root_fail_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rtlsdr_wsprd_amd as w
    from rtlsdr_wsprd_amd import dist as wd
    I = Q = torch.zeros(4, 45000) if rank == 0 else None

    def decode_shard(mi, mq, o):
        if rank == 1:
            raise RuntimeError("no usable HIP device")
        n = mi.shape[0]
        return (w.decoder_results * (n * K))(), (C.c_int * n)()
    try:
        wd.decode_from_root(I, Q, 4, 45000, w.default_options(), decode_shard, max_results=K, record_size=80, root=0)
        q.put(("returned", rank))
    except RuntimeError as e:
        q.put(("raised", rank, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_a_failing_shard_stops_every_rank_of_decode_from_root():
    """Rank 1's decode fails (no GPU, say): both ranks raise instead of rank 0 waiting in the gather for ever."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + os.getpid() % 2000
    procs = [ctx.Process(target=_root_fail_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[0] for g in got] == ["raised", "raised"] and all("rank 1" in g[2] for g in got)


def _run_root_scatter(product, world=2, nseg=5):
    sys.path.insert(0, ROOT)
    import rtlsdr_wsprd_amd as w
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000 + (13 if product else 0) + world
    procs = [ctx.Process(target=_root_worker, args=(r, world, port, q, product, nseg)) for r in range(world)]
    for p in procs:
        p.start()
    counts, msgs = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    segs = _segments()[:nseg]
    out, cnt, n = _decode_shard(0, nseg, segs, w.default_options())
    single = [[bytes(out[i * K + k].message).split(b"\0")[0].decode() for k in range(cnt[i])] for i in range(nseg)]
    assert msgs == single and counts == list(cnt) and sum(counts) >= nseg - 1


def test_world2_real_input_scattered_from_root():
    """Rank 0 alone holds five segments and the options: options broadcast, rows scattered in shard_range() blocks
    (3 + 2: unequal), every rank decodes what it received, the records come back in global order == one process."""
    _run_root_scatter(product=False)


def test_world3_real_input_scattered_from_root():
    _run_root_scatter(product=False, world=3, nseg=4)           # blocks of 2, 1, 1


@pytest.mark.gpu
def test_world2_real_input_scattered_from_root_through_the_product():
    _run_root_scatter(product=True)


def _order_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtlsdr_wsprd_amd import dist as wd

    def turn():
        if rank == 0:
            raise ValueError("decoder said no")
        return rank
    try:
        wd.in_rank_order(turn)
        q.put((rank, "no error"))
    except ValueError as e:
        q.put((rank, "own: %s" % e))
    except RuntimeError as e:
        q.put((rank, "other: %s" % e))
    dist.destroy_process_group()


def test_in_rank_order_does_not_hang_when_a_rank_fails():
    """A rank whose turn raises still reaches every barrier; it re-raises its own error afterwards and the other
    ranks learn that a turn failed (round-2 advisor finding: they used to block forever)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_order_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0].startswith("own: decoder said no") and got[1].startswith("other:")


def _cwd_worker(rank, world, port, q, dirs):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.chdir(dirs[rank])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtlsdr_wsprd_amd import dist as wd
    try:
        q.put((rank, "ran: %r" % (wd.in_rank_order(lambda: rank),)))
    except RuntimeError as e:
        q.put((rank, "refused: %s" % str(e)[:60]))
    dist.destroy_process_group()


@pytest.mark.parametrize("shared", [True, False])
def test_in_rank_order_wants_evidence_of_one_directory(tmp_path, shared):
    """Ranks in the same directory (reached through different paths: a symlink) take their turns; ranks in different
    directories would each keep their own hashtable.txt and are refused (round-3 advisor finding: the check compared
    host names and paths, which refused every multi-node job and says nothing about what the paths point at)."""
    a = tmp_path / "a"; a.mkdir()
    if shared:
        b = tmp_path / "b"; b.symlink_to(a, target_is_directory=True)
    else:
        b = tmp_path / "b"; b.mkdir()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 37500 + os.getpid() % 2000 + (1 if shared else 0)
    procs = [ctx.Process(target=_cwd_worker, args=(r, 2, port, q, [str(a), str(b)])) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if shared:
        assert got == {0: "ran: 0", 1: "ran: 1"} and not [f for f in os.listdir(a) if f.startswith(".wspr_rank_order")]
    else:
        assert got[0].startswith("refused:") and got[1].startswith("refused:")

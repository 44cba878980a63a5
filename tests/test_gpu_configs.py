"""GPU tests of the BASELINE.json configurations that round 1 left untested, and of the reference-held
golden lines on the HIP path:
  * configs[2]: 8 192 segments x 10 overlapping signals (-10..-28 dB), deep search on;
  * configs[4]: full-size raw segments (576 000 000 bytes of u8 IQ) -> K0 -> decode, incl. a clipped
    segment whose CIC integrators wrap;
  * the -t self-test signal (rtlsdr_wsprd.c:729-760; REPORT.md:198) and the -r playback chain
    reader -> wspr_decode -> spot line (REPORT.md:202) through the product only;
  * K1 against a float64 FFT over every block of several segments;
  * bench.py's RCCL path (broadcast of the options, gather of the spot records) with one rank."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NS = 45000


@pytest.fixture(scope="module")
def env():
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import rtlsdr_wsprd_amd as w
    assert w.lib().wspr_device_ready() == 1
    torch.cuda.set_device(0)
    return torch, bench, w, torch.device("cuda", 0)


def _tup(s):
    return (s.message, s.call, s.loc, s.pwr, s.cycles, s.jitter, s.drift, s.sync, s.snr, s.dt, s.freq)


def _same_as_oracle(got, ref):
    """every field equal, snr to 1e-4 dB (device vs host log10f is not involved any more, but the
    stated tolerance stays)"""
    g = [_tup(x) for x in got]
    r = [_tup(x) for x in ref]
    return [t[:8] + t[9:] for t in g] == [t[:8] + t[9:] for t in r] and \
        all(abs(a[8] - b[8]) < 1e-4 for a, b in zip(g, r))


# ------------------------------------------------------------------ configs[2]
def test_config3_8192_segments_x_10_signals(env):
    torch, bench, w, dev = env
    nseg = 8192
    I, Q, expected = bench.synth_batch_gpu(nseg, 4321, dev, 10, -10.0, -28.0, 0.3)
    dec = w.BatchDecoder(nseg, 32)
    dec.decode(I, Q)
    full = [[_tup(x) for x in dec.spots(s)] for s in range(nseg)]
    msgs = [[t[0].decode() for t in seg] for seg in full]
    n_ok = sum(len(set(expected[s]) & set(msgs[s])) for s in range(nseg))
    n_false = sum(m not in expected[s] for s in range(nseg) for m in msgs[s])
    print("configs[2]: %d/%d signals decoded, %d false" % (n_ok, 10 * nseg, n_false))
    assert n_ok >= 0.95 * 10 * nseg
    assert n_false == 0
    # segments are independent: the first half alone gives the first half's spots
    h = w.BatchDecoder(nseg // 2, 32)
    h.decode(I[: nseg // 2].contiguous(), Q[: nseg // 2].contiguous())
    assert [[_tup(x) for x in h.spots(s)] for s in range(nseg // 2)] == full[: nseg // 2]
    # exact agreement with the CPU oracle on 1 024 sampled segments (all fields of all spots, in order); a longer
    # soak: WSPR_CONFIG3_ORACLE_SEGMENTS=2048 (the oracle calls run on a thread pool: ctypes drops the GIL)
    from concurrent.futures import ThreadPoolExecutor
    nsample = int(os.environ.get("WSPR_CONFIG3_ORACLE_SEGMENTS", "1024"))
    step = max(1, nseg // nsample)
    picks = list(range(5 % step, nseg, step))          # WSPR_CONFIG3_ORACLE_SEGMENTS=8192: every segment
    Ih, Qh = I.cpu().numpy(), Q.cpu().numpy()
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
        refs = list(pool.map(lambda s: ol.decode(Ih[s], Qh[s], NS)[0], picks))
    for s, ref in zip(picks, refs):
        assert _same_as_oracle(dec.spots(s), ref), s
        assert len(ref) >= 8 or nsample > 16
    # neither the Fano budget split of the host pool nor the device search for every attempt (what such a
    # crowded batch runs by default once a pipeline has seen its time-outs) ever changes a result
    L = w.lib()
    L.wspr_set_fano_fast_budget.restype = C.c_uint
    old = L.wspr_set_fano_fast_budget(C.c_uint(300))
    old_mode = L.wspr_set_fano_device_mode(0)
    try:
        dec.decode(I, Q)
        assert [[_tup(x) for x in dec.spots(s)] for s in range(nseg)] == full
        assert w.last_timings()["fano_left_to_device"] > 1000
        L.wspr_set_fano_device_mode(1)
        dec.decode(I, Q)
        assert [[_tup(x) for x in dec.spots(s)] for s in range(nseg)] == full
        tm = w.last_timings()
        assert tm["host_fano_ms"] == 0 and tm["segments_redecoded"] == 0 and tm["fano_timeouts"] > 1000
    finally:
        L.wspr_set_fano_fast_budget(C.c_uint(old))
        L.wspr_set_fano_device_mode(old_mode)


# ------------------------------------------------------------------ configs[3], one rank's shard
def test_config4_shard_of_configs3_8192_single_signal_segments(env):
    """configs[3] = 65 536 segments over 8 GPUs: no 8-GPU box is in reach of the suite, but its per-rank workload -- the
    dist.shard_range() block of 8 192 SINGLE-SIGNAL segments ('config 4 ... as config 2'; the independence that
    makes it shardable: wsprd.c:478-479) -- had never run anywhere (verdict of round 4: the largest single-signal batch
    under test was 1 024).  Rank 3's block, generated with that rank's seed as bench.py --config 4 / --gpus 8 does:
    decode rate, no false decode, 512 sampled segments equal the oracle field for field, the two halves and a
    permutation of the block give the same spots (what makes the shard boundaries irrelevant)."""
    torch, bench, w, dev = env
    from concurrent.futures import ThreadPoolExecutor
    from rtlsdr_wsprd_amd import dist as wd
    world, rank = 8, 3
    lo, hi = wd.shard_range(65536, rank, world)
    nseg = hi - lo
    assert nseg == 8192
    I, Q, expected = bench.synth_batch_gpu(nseg, 1234 + rank, dev, 1, -20.0, -20.0, 1.0)
    dec = w.BatchDecoder(nseg, 16)
    dec.decode(I, Q)
    full = [[_tup(x) for x in dec.spots(s)] for s in range(nseg)]
    msgs = [[t[0].decode() for t in seg] for seg in full]
    n_ok = sum(expected[s][0] in msgs[s] for s in range(nseg))
    n_false = sum(m not in expected[s] for s in range(nseg) for m in msgs[s])
    print("configs[3] shard of rank %d: %d/%d decoded, %d false" % (rank, n_ok, nseg, n_false))
    assert n_ok >= 0.95 * nseg and n_false == 0
    # oracle-exact on 512 sampled segments (every 16th)
    picks = list(range(7, nseg, 16))
    Ih, Qh = I.cpu().numpy(), Q.cpu().numpy()
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
        refs = list(pool.map(lambda s: ol.decode(Ih[s], Qh[s], NS)[0], picks))
    assert len(picks) == 512
    for s, ref in zip(picks, refs):
        assert _same_as_oracle(dec.spots(s), ref), s
    # the block's halves on their own (a 16-rank split of the same job) and a permutation of its rows
    for a, b in ((0, nseg // 2), (nseg // 2, nseg)):
        h = w.BatchDecoder(b - a, 16)
        h.decode(I[a:b].contiguous(), Q[a:b].contiguous())
        assert [[_tup(x) for x in h.spots(s)] for s in range(b - a)] == full[a:b]
    perm = torch.randperm(nseg, generator=torch.Generator().manual_seed(65536)).to(dev)
    p = w.BatchDecoder(nseg, 16)
    p.decode(I[perm].contiguous(), Q[perm].contiguous())
    pl = perm.cpu().numpy()
    assert all([_tup(x) for x in p.spots(i)] == full[pl[i]] for i in range(nseg))
    # the same block through the device-Fano mode a rank with a 1/8 CPU share runs by default
    L = w.lib()
    old_mode = L.wspr_set_fano_device_mode(1)
    try:
        dec.decode(I, Q)
        assert [[_tup(x) for x in dec.spots(s)] for s in range(nseg)] == full
    finally:
        L.wspr_set_fano_device_mode(old_mode)


# ------------------------------------------------------------------ configs[4]
def test_config5_full_size_raw_segments_through_k0_and_decode(env):
    """Three complete 2-minute raw segments (576 000 000 bytes each): the config's signal level, a strong
    in-band signal whose comb outputs wrap the int32 CIC (Q8), and one with heavy clipping incl. bytes
    0x00 / 0xff (Q9).  Decimated IQ bit-exact vs the oracle front end; spots equal the oracle decoder's."""
    torch, bench, w, dev = env
    RAW = bench.RAW_BYTES
    raw0, exp0 = bench.synth_raw_gpu(1, 31, dev, -20.0)                          # the benchmarked kind
    raw1, exp1 = bench.synth_raw_gpu(1, 32, dev, noise_lsb=10.0, amp_lsb=30.0)   # strong: the int32 CIC wraps
    raw2, exp2 = bench.synth_raw_gpu(1, 33, dev, -5.0, noise_lsb=90.0)           # clipped rails
    raw = torch.cat([raw0, raw1, raw2])
    assert int((raw[2] == 0).sum()) > 1000 and int((raw[2] == 255).sum()) > 1000
    nseg = 3
    stride = int(w.lib().wspr_iq_stride())
    dI = torch.zeros(nseg, stride, device=dev)
    dQ = torch.zeros(nseg, stride, device=dev)
    w.sync_torch()                                 # raw pointers next: torch's fills and the cat have to be done
    assert w.lib().wspr_decimate_u8_batch_device(raw.data_ptr(), RAW, nseg, dI.data_ptr(), dQ.data_ptr(), 1) == 0
    dec = w.BatchDecoder(nseg, 32)
    dec.decode_ptr(dI.data_ptr(), dQ.data_ptr(), NS, stride)      # working copies are taken; dI/dQ stay
    gi, gq = dI.cpu().numpy(), dQ.cpu().numpy()
    L = ol.lib()
    peaks = []
    for s in range(nseg):
        host = raw[s].cpu().numpy()
        st = L.orc_decim_new()
        oi = np.zeros(NS, np.float32); oq = np.zeros(NS, np.float32)
        fill = L.orc_decim_feed(C.c_void_p(st), ol.ptr(host), RAW, ol.ptr(oi), ol.ptr(oq), 0, NS)
        L.orc_decim_free(C.c_void_p(st))
        assert fill == 44992                                                     # floor(288e6 / 6401)
        peaks.append(float(max(np.abs(oi[:fill]).max(), np.abs(oq[:fill]).max())))
        L.orc_normalise(ol.ptr(oi), ol.ptr(oq), C.c_int(fill), C.c_int(NS))
        assert np.array_equal(gi[s, :NS], oi) and np.array_equal(gq[s, :NS], oq), s
        ref, _, _ = ol.decode(oi, oq, NS)
        assert _same_as_oracle(dec.spots(s), ref), s
        print("raw segment %d: %d spots %s" % (s, len(ref), [x.message.decode() for x in ref]))
    assert [x.message.decode() for x in dec.spots(0)] == exp0[0]
    # the CIC passes an in-band tone with a gain of 8.4e7 per LSB: 30 LSB would be 2.5e9 > 2^31, so the
    # second segment's comb outputs wrapped (its peak stays far below the linear value)
    assert peaks[1] < 0.9 * 30.0 * 8.3e7 and 30.0 * 8.3e7 > 2.0 ** 31


def test_config5_64_distinct_raw_segments_in_waves(env):
    """configs[4] beyond three segments (verdict of round 4): 64 DISTINCT full-size raw segments (36.9 GB resident), through
    the front end in four waves of 16 into the rows of an IQ ring -- the shape bench.py --config 5 runs -- and one decoder
    call over the 64 rows.  Every row's decimated IQ bit-exact vs the oracle front end (parity UNPINNED for this function:
    oracle/orc_frontend.c has no reference-held fixture), every row's spots equal the oracle decoder's field for field."""
    torch, bench, w, dev = env
    from concurrent.futures import ThreadPoolExecutor
    RAW = bench.RAW_BYTES
    nraw, wave = 64, 16
    raw, expected = bench.synth_raw_gpu(nraw, 4242, dev, -20.0)
    L = w.lib()
    stride = int(L.wspr_iq_stride())
    dI = torch.zeros(nraw, stride, device=dev)
    dQ = torch.zeros(nraw, stride, device=dev)
    w.sync_torch()
    row = stride * 4
    for wv in range(nraw // wave):
        assert L.wspr_decimate_u8_batch_device(raw.data_ptr() + wv * wave * RAW, RAW, wave, dI.data_ptr() + wv * wave * row,
                                               dQ.data_ptr() + wv * wave * row, 1) == 0
    dec = w.BatchDecoder(nraw, 32)
    dec.decode_ptr(dI.data_ptr(), dQ.data_ptr(), NS, stride)
    gi, gq = dI.cpu().numpy(), dQ.cpu().numpy()
    O = ol.lib()

    def one(s):
        host = raw[s].cpu().numpy()
        st = O.orc_decim_new()
        oi = np.zeros(NS, np.float32); oq = np.zeros(NS, np.float32)
        fill = O.orc_decim_feed(C.c_void_p(st), ol.ptr(host), RAW, ol.ptr(oi), ol.ptr(oq), 0, NS)
        O.orc_decim_free(C.c_void_p(st))
        O.orc_normalise(ol.ptr(oi), ol.ptr(oq), C.c_int(fill), C.c_int(NS))
        ref, _, _ = ol.decode(oi, oq, NS)
        return fill, oi, oq, ref
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:        # 576 MB per task in flight
        outs = list(pool.map(one, range(nraw)))
    n_dec = 0
    for s, (fill, oi, oq, ref) in enumerate(outs):
        assert fill == 44992
        assert np.array_equal(gi[s, :NS], oi) and np.array_equal(gq[s, :NS], oq), s
        assert _same_as_oracle(dec.spots(s), ref), s
        n_dec += [x.message.decode() for x in dec.spots(s)] == expected[s]
    print("configs[4]: %d distinct raw segments in %d front-end waves, %d/%d decode their message, all equal to the oracle"
          % (nraw, nraw // wave, n_dec, nraw))
    assert n_dec >= 0.9 * nraw


# ------------------------------------------------------------------ reference-held golden lines
def test_selftest_signal_spot_line_on_the_hip_path(env):
    """decoderSelfTest(), rtlsdr_wsprd.c:729-789: glibc rand() seed 1, amplitude 1, NOT normalised."""
    torch, bench, w, dev = env
    from test_oracle_golden import _selftest_signal
    I, Q = _selftest_signal()
    spots, _, _ = w.wspr_decode(I, Q, NS)
    assert len(spots) >= 1
    s = spots[0]
    assert (s.call, s.loc, s.pwr) == (b"K1JT", b"FN20", b"20")                   # rtlsdr_wsprd.c:782-788
    line = "Spot(%i) %6.2f %6.2f %10.6f %2d %7s %6s %2s" % (
        0, s.snr, s.dt, s.freq, int(s.drift), s.call.decode(), s.loc.decode(), s.pwr.decode())
    assert line == "Spot(0)  22.80   0.01 144.490550  0    K1JT   FN20 20"       # REPORT.md:198
    ref, _, _ = ol.decode(I, Q, NS)
    assert _same_as_oracle(spots, ref)


def test_playback_chain_reader_decode_format(env):
    """-r playback (rtlsdr_wsprd.c:669-701) through the product only: file reader -> wspr_decode ->
    spot line, byte-identical to REPORT.md:202."""
    torch, bench, w, dev = env
    L = w.lib()
    I = np.zeros(NS, np.float32); Q = np.zeros(NS, np.float32)
    n = L.wspr_read_iq_file(os.path.join(ol.GOLDEN, "refSignalSnr0dB.iq").encode(), ol.ptr(I), ol.ptr(Q))
    assert n == NS
    out = (w.decoder_results * 50)()
    nres = C.c_int(0)
    assert L.wspr_decode(ol.ptr(I), ol.ptr(Q), n, w.default_options(), C.addressof(out), C.addressof(nres)) == 0
    assert nres.value == 1
    buf = C.create_string_buffer(128)
    L.wspr_format_spot(C.byref(out[0]), buf, 128)
    assert buf.value.decode() == "Spot :  -0.07   0.01 144.490550  0    K1JT   FN20 20"


# ------------------------------------------------------------------ K1 vs float64
def test_fft_bank_every_block_against_float64(env):
    """The FFT in the reference is FFTW (un-vendored); product and oracle share one float32 radix-2.
    Independent evidence: all 347 blocks of three segments (reference file, -20 dB single signal,
    ten overlapping signals) against a float64 FFT of the same float32 windowed samples."""
    torch, bench, w, dev = env
    Ir, Qr, _ = ol.read_iq_file(os.path.join(ol.GOLDEN, "refSignalSnr0dB.iq"))
    I1, Q1, _ = bench.synth_batch_gpu(1, 9, dev, 1, -20.0, -20.0, 1.0)
    I10, Q10, _ = bench.synth_batch_gpu(1, 10, dev, 10, -10.0, -28.0, 0.3)
    I = np.stack([Ir, I1[0].cpu().numpy(), I10[0].cpu().numpy()])
    Q = np.stack([Qr, Q1[0].cpu().numpy(), Q10[0].cpu().numpy()])
    blocks = 347
    out = np.zeros((3, 512, blocks), np.float32)
    assert w.lab().wspr_stage_fft_bank(ol.ptr(I), ol.ptr(Q), 3, NS, NS, ol.ptr(out)) == blocks
    win = np.sin(0.006147931 * np.arange(512)).astype(np.float32)      # sinf of a double argument, wsprd.c:512
    idx = 128 * np.arange(blocks)[:, None] + np.arange(512)[None, :]
    worst_peak, worst_rel = 0.0, 0.0
    for s in range(3):
        pad_i = np.concatenate([I[s], np.zeros(512, np.float32)])
        pad_q = np.concatenate([Q[s], np.zeros(512, np.float32)])
        x = (pad_i[idx] * win).astype(np.float64) + 1j * (pad_q[idx] * win).astype(np.float64)
        p = np.abs(np.fft.fftshift(np.fft.fft(x, axis=1), axes=1)) ** 2             # [blocks, 512]
        got = out[s, 48:465, :].T.astype(np.float64)
        ref = p[:, 48:465]
        err = np.abs(got - ref)
        worst_peak = max(worst_peak, float((err / ref.max(axis=1, keepdims=True)).max()))
        floor = np.median(ref, axis=1, keepdims=True)                               # the noise floor of a block
        above = ref >= floor
        worst_rel = max(worst_rel, float((err[above] / ref[above]).max()))
    print("K1 vs float64: max |err| / block peak = %.3g, max relative error on bins above the block's median "
          "= %.3g" % (worst_peak, worst_rel))
    assert worst_peak < 1e-6
    assert worst_rel < 1e-5           # SURVEY gate: 1e-5 relative on ps (bins the picker can act on)


# ------------------------------------------------------------------ RCCL path, one rank
def test_bench_rccl_path_world_size_one():
    """bench.py with WSPR_BENCH_FORCE_DIST=1: process group on the nccl (= RCCL) backend, options broadcast,
    spot records gathered on rank 0 -- the code path the 2/4/8-GPU runs take, on the one GPU present."""
    envv = dict(os.environ, WSPR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531",
                RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "2", "--segments", "256",
                        "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--min-seconds", "0"],
                       env=envv, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    ok, sent = map(int, d["decoded_ok"].split("/"))
    assert d["n_gpus"] == 1 and ok >= 0.95 * sent and d["false_decodes"] == 0
    assert d["spots_total"] >= ok                      # counted from the GATHERED records
    assert d["config"]["gathered_over"] == "rccl"


def _bench_line(argv, env, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def _no_launcher_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra)
    return env


def test_bench_gpus_flag_spawns_its_own_rank_over_rccl():
    """`python bench.py --gpus 1 --spawn` with NO launcher environment: bench.py re-executes itself under
    torch.distributed.run (the path `--gpus N` takes for N > 1), the rank builds the RCCL process group, the options are
    broadcast, real input is scattered from rank 0 and decoded, the spot records are gathered (round-2 verdict: --gpus
    used to be parsed and ignored)."""
    d = _bench_line(["--gpus", "1", "--spawn", "--config", "2", "--segments", "256", "--steps", "6", "--warmup", "2",
                     "--no-cpu-baseline", "--min-seconds", "0"], _no_launcher_env())
    ok, sent = map(int, d["decoded_ok"].split("/"))
    assert d["n_gpus"] == 1 and ok >= 0.95 * sent and d["false_decodes"] == 0
    assert d["config"]["gathered_over"] == "rccl" and "self-spawned" in d["config"]["launched_by"]
    f = d["fanout_check"]
    assert f["segments_scattered_from_rank0"] == 4 and f["equal_to_rank0_own_decode"] == "4/4"


@pytest.mark.parametrize("config", ["2", "3"])
def test_bench_gpus_2_spawns_two_ranks_sharing_the_one_gpu(config):
    """`python bench.py --gpus 2`: two ranks really start, see each other, split the host's CPUs, each decodes its own
    batch, rank 0 scatters eight of ITS segments over both ranks and gets the same spots back, and prints n_gpus 2
    with the aggregate rate.  On this 1-GPU box the two ranks share the device and the collectives run on gloo (RCCL
    refuses two ranks on one device) -- everything else is the code the 2/4/8-GPU runs execute."""
    # (config 3 = the driver's own default, the headline workload, at 1/32 of its size: ten signals per segment, the device
    # Fano search, 32 spots per segment in the gathered records)
    d = _bench_line(["--gpus", "2", "--config", config, "--segments", "256", "--steps", "6", "--warmup", "2",
                     "--no-cpu-baseline", "--min-seconds", "0", "--no-kernel-roofline"],
                    _no_launcher_env(WSPR_BENCH_SHARE_GPU="1", WSPR_BENCH_BACKEND="gloo", WSPR_HOST_THREADS="2"))
    ok, sent = map(int, d["decoded_ok"].split("/"))
    assert d["n_gpus"] == 2 and ok >= (0.95 if config == "2" else 0.9) * sent and d["false_decodes"] == 0
    assert sent == (256 if config == "2" else 2560) and len(d["ranks"]) == 2
    # the line says that the two ranks sat on ONE device (round-3 advisor finding), and a rank with the CPU share of an
    # 8-rank job (2 of these boxes' 16 CPUs) adds no pool threads to the lane threads that drive its batches
    assert d["distinct_devices"] == 1 and d["devices_shared"] is True
    assert d["host_threads"] == 2 and d["host_pool_workers"] == 0
    assert d["config"]["segments_per_gpu"] == 256 and d["spots_total"] >= (2 * ok - 4 if config == "2" else 1.9 * ok)   # both ranks' records arrived
    assert d["fanout_check"]["segments_scattered_from_rank0"] == 8
    assert d["fanout_check"]["equal_to_rank0_own_decode"] == "8/8"
    assert d["value"] > 0 and d["scaling"] == "weak"


def test_bench_gpus_8_rehearsal_of_configs3_on_the_one_gpu():
    """The 8-rank job configs[3] is quoted on, rehearsed on this 1-GPU box so that the first real 8-GPU run is not a
    debugging session: `bench.py --gpus 8 --config 4 --segments 1024` -- eight self-spawned ranks (sharing the device,
    collectives on gloo: RCCL refuses two ranks on one device), each with the CPU share of a rank of eight (2 host
    threads) and its own shard.  Eight distinct contiguous shard ranges of shard_range(8 x 1024, r, 8) shape, no pool
    thread on any rank, every rank's records gathered on rank 0, and rank 0's real input scattered to the seven others
    and decoded there to the spots rank 0 gets itself."""
    sys.path.insert(0, ROOT)
    from rtlsdr_wsprd_amd import dist as wd
    d = _bench_line(["--gpus", "8", "--config", "4", "--segments", "1024", "--steps", "6", "--warmup", "2", "--rotate", "2",
                     "--no-cpu-baseline", "--no-warm-extra", "--no-kernel-roofline", "--min-seconds", "0", "--inflight", "4"],
                    _no_launcher_env(WSPR_BENCH_SHARE_GPU="1", WSPR_BENCH_BACKEND="gloo", WSPR_HOST_THREADS="2"), timeout=1500)
    assert d["n_gpus"] == 8 and d["distinct_devices"] == 1 and d["devices_shared"] is True and d["scaling"] == "weak"
    ranks = d["ranks"]
    assert [r["rank"] for r in ranks] == list(range(8))
    assert [tuple(r["segments"]) for r in ranks] == [wd.shard_range(8 * 1024, r, 8) for r in range(8)]
    # (the same partition the full job uses: shard_range(65 536, r, 8) = 8 192-segment blocks, here at 1/8 of the size)
    assert [wd.shard_range(65536, r, 8) for r in range(8)] == [(8192 * r, 8192 * (r + 1)) for r in range(8)]
    assert [tuple(r["segments"]) for r in ranks] == [(1024 * r, 1024 * (r + 1)) for r in range(8)]
    assert len({tuple(r["segments"]) for r in ranks}) == 8 and ranks[-1]["segments"][1] == 8 * 1024
    assert all(r["host_threads"] == 2 and r["host_pool_workers"] == 0 for r in ranks), ranks
    for r in ranks:
        ok, sent = map(int, r["decoded_ok"].split("/"))
        assert sent == 1024 and ok >= 0.95 * sent and r["false_decodes"] == 0 and r["spots_last_step"] >= ok
    # the gathered records on rank 0 hold every rank's spots of the last step
    assert d["spots_total"] == sum(r["spots_last_step"] for r in ranks)
    f = d["fanout_check"]
    assert f["segments_scattered_from_rank0"] == 32 and f["equal_to_rank0_own_decode"] == "32/32"
    assert d["config"]["segments_per_gpu"] == 1024 and d["value"] > 0
    print("8-rank rehearsal on one GPU: %.0f segments/s aggregate, %.1f ms per step, gather %.2f ms per step" % (
        d["value"], d["ms_per_step"], d.get("gather_ms_per_step") or 0.0))


# ------------------------------------------------------------------ receiver session (f4)
def test_receiver_session_two_minute_flow(env):
    """The reference's receive loop through the session object: one full 2-minute raw segment arrives in
    librtlsdr callback buffers (65 536 bytes), the buffers roll over, the completed buffer is decoded; the next
    slot only receives a few callbacks and is skipped as too short (rtlsdr_wsprd.c:126-244, 263-328, 1179-1182).
    Decimated IQ, spots and the carried decimator state equal the oracle fed with the same callbacks."""
    torch, bench, w, dev = env
    L = w.lib()
    raw, exp = bench.synth_raw_gpu(1, 61, dev, -18.0)
    host = raw[0].cpu().numpy()
    extra = bench.synth_raw_gpu(1, 62, dev, -18.0)[0][0][: 65536 * 40].cpu().numpy()      # the start of the next slot
    L.wspr_session_create.restype = C.c_void_p
    L.wspr_session_create.argtypes = [w.decoder_options]
    L.wspr_session_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.wspr_session_rollover.argtypes = [C.c_void_p]
    L.wspr_session_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.wspr_session_fill.argtypes = [C.c_void_p, C.c_int]
    L.wspr_session_fill.restype = C.c_uint32
    L.wspr_session_samples.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.wspr_session_samples.restype = C.POINTER(C.c_float)
    L.wspr_session_destroy.argtypes = [C.c_void_p]
    s = L.wspr_session_create(w.default_options(freq=14095600))
    O = ol.lib()
    ost = O.orc_decim_new()
    oi = [np.zeros(NS, np.float32), np.zeros(NS, np.float32)]
    oq = [np.zeros(NS, np.float32), np.zeros(NS, np.float32)]
    ofill = 0
    CB = 65536
    for pos in range(0, host.size, CB):
        chunk = np.ascontiguousarray(host[pos:pos + CB])
        fill = L.wspr_session_feed(s, ol.ptr(chunk), chunk.size)
        ofill = O.orc_decim_feed(C.c_void_p(ost), ol.ptr(chunk), chunk.size, ol.ptr(oi[0]), ol.ptr(oq[0]), ofill, NS)
        assert fill == ofill, pos
    assert ofill == 44992
    done = L.wspr_session_rollover(s)
    assert done == 0 and L.wspr_session_fill(s, 1) == 0
    # the RX thread keeps feeding the other buffer while the decoder works on the completed one
    ofill1 = 0
    for pos in range(0, extra.size, CB):
        chunk = np.ascontiguousarray(extra[pos:pos + CB])
        fill = L.wspr_session_feed(s, ol.ptr(chunk), chunk.size)
        ofill1 = O.orc_decim_feed(C.c_void_p(ost), ol.ptr(chunk), chunk.size, ol.ptr(oi[1]), ol.ptr(oq[1]), ofill1, NS)
        assert fill == ofill1
    out = (w.decoder_results * 50)()
    n = C.c_int(0)
    assert L.wspr_session_decode(s, done, C.addressof(out), C.byref(n)) == 1
    O.orc_normalise(ol.ptr(oi[0]), ol.ptr(oq[0]), C.c_int(ofill), C.c_int(NS))
    ref, ri, rq = ol.decode(oi[0], oq[0], NS, ol.default_options(freq=14095600))
    assert _same_as_oracle([out[k] for k in range(n.value)], ref) and n.value >= 1
    assert [out[k].message.decode() for k in range(n.value)][0] == exp[0][0]
    gi = np.ctypeslib.as_array(L.wspr_session_samples(s, done, 0), shape=(NS,))
    gq = np.ctypeslib.as_array(L.wspr_session_samples(s, done, 1), shape=(NS,))
    assert np.array_equal(gi, ri) and np.array_equal(gq, rq)          # the residual saveSample() would write
    # the second slot is only 40 callbacks long: the carried state gave the same samples as the oracle's stream ...
    g1 = np.ctypeslib.as_array(L.wspr_session_samples(s, 1, 0), shape=(NS,))
    assert ofill1 > 100 and np.array_equal(g1[:ofill1], oi[1][:ofill1])
    # ... and the decoder skips it (fewer than 117 s of samples)
    assert L.wspr_session_rollover(s) == 1
    assert L.wspr_session_decode(s, 1, C.addressof(out), C.byref(n)) == 0 and n.value == 0
    O.orc_decim_free(C.c_void_p(ost))
    L.wspr_session_destroy(s)


def test_receiver_session_rollover_during_a_feed(env):
    """Round-2 advisor finding: feed() read `active`, ran a GPU round trip, then committed into that buffer -- a
    roll-over (and the decoder's in-place normalisation) in between corrupted the completed slot.  Now feed() and
    rollover() exclude each other.  An RX thread feeds 600 callbacks while the main thread rolls the buffers over
    four times at arbitrary moments and reads the completed buffer at once: every completed buffer must hold exactly
    the next run of the oracle's output stream (whole callbacks, nothing lost, nothing written after the hand-over),
    and a thread may not bind the sessions' lane."""
    import threading
    import time
    torch, bench, w, dev = env
    L = w.lib()
    L.wspr_session_create.restype = C.c_void_p
    L.wspr_session_create.argtypes = [w.decoder_options]
    L.wspr_session_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.wspr_session_rollover.argtypes = [C.c_void_p]
    L.wspr_session_fill.argtypes = [C.c_void_p, C.c_int]
    L.wspr_session_fill.restype = C.c_uint32
    L.wspr_session_samples.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.wspr_session_samples.restype = C.POINTER(C.c_float)
    L.wspr_session_destroy.argtypes = [C.c_void_p]
    assert L.wspr_bind_thread_lane(21) == 15 and L.wspr_bind_thread_lane(0) == 0     # lane 16 is the sessions'
    CB, NCB = 65536, 600
    rng = np.random.default_rng(99)
    host = rng.integers(0, 256, CB * NCB, dtype=np.uint8)
    O = ol.lib()
    ost = O.orc_decim_new()
    oi, oq = np.zeros(NS, np.float32), np.zeros(NS, np.float32)
    nout = O.orc_decim_feed(C.c_void_p(ost), ol.ptr(host), host.size, ol.ptr(oi), ol.ptr(oq), 0, NS)
    O.orc_decim_free(C.c_void_p(ost))
    s = L.wspr_session_create(w.default_options())
    errs = []

    def rx():
        for k in range(NCB):
            chunk = np.ascontiguousarray(host[k * CB:(k + 1) * CB])
            if L.wspr_session_feed(s, ol.ptr(chunk), chunk.size) < 0:
                errs.append(k)
    t = threading.Thread(target=rx)
    t.start()
    pieces = []
    for _ in range(4):
        time.sleep(0.04)
        done = L.wspr_session_rollover(s)
        n = L.wspr_session_fill(s, done)
        gi = np.ctypeslib.as_array(L.wspr_session_samples(s, done, 0), shape=(NS,))[:n].copy()
        gq = np.ctypeslib.as_array(L.wspr_session_samples(s, done, 1), shape=(NS,))[:n].copy()
        time.sleep(0.01)                                     # a late write into the handed-over buffer would show here
        assert L.wspr_session_fill(s, done) == n
        assert np.array_equal(np.ctypeslib.as_array(L.wspr_session_samples(s, done, 0), shape=(NS,))[:n], gi)
        pieces.append((gi, gq))
    t.join()
    done = L.wspr_session_rollover(s)
    n = L.wspr_session_fill(s, done)
    pieces.append((np.ctypeslib.as_array(L.wspr_session_samples(s, done, 0), shape=(NS,))[:n].copy(),
                   np.ctypeslib.as_array(L.wspr_session_samples(s, done, 1), shape=(NS,))[:n].copy()))
    L.wspr_session_destroy(s)
    assert not errs
    gi = np.concatenate([p[0] for p in pieces]); gq = np.concatenate([p[1] for p in pieces])
    assert gi.size == nout and nout == CB * NCB // 2 // 6401
    assert np.array_equal(gi, oi[:nout]) and np.array_equal(gq, oq[:nout])
    assert sum(1 for p in pieces if p[0].size) >= 2          # the roll-overs really fell inside the stream


def test_batch_decimator_rejects_misaligned_rows(env):
    torch, bench, w, dev = env
    raw = torch.zeros(2 * 12802 * 4 + 64, device=dev, dtype=torch.uint8)
    I = torch.zeros(2, int(w.lib().wspr_iq_stride()), device=dev); Q = torch.zeros_like(I)
    w.sync_torch()
    L = w.lib()
    assert L.wspr_decimate_u8_batch_device(raw.data_ptr(), 12802 * 4, 2, I.data_ptr(), Q.data_ptr(), 0) == -1     # 51208 % 16 = 8
    assert L.wspr_decimate_u8_batch_device(raw.data_ptr() + 8, 12800 * 4, 2, I.data_ptr(), Q.data_ptr(), 0) == -1
    assert L.wspr_decimate_u8_batch_device(raw.data_ptr(), 12800 * 4, 2, I.data_ptr(), Q.data_ptr(), 0) == 0


# ------------------------------------------------------------------ calibration hooks behind the rooflines
def test_calibration_hooks_report_plausible_ceilings(env):
    """The three ceilings bench.py prints beside the rooflines come from kernels of the library itself: a stream copy
    (4 and 16 bytes per lane), a read-only stream with K0's access pattern, and register-only chains of separately
    rounded packed multiplies and adds.  They must run and land where an MI355X can be: between a fifth of the
    peak and the peak."""
    import time
    torch, bench, w, dev = env
    L = w.lab()                    # include/wspr_mi355x_bench.h: the calibration kernels are not in the product
    assert not hasattr(w.lib(), "wspr_calib_copy") or os.environ.get("WSPR_USE_LAB") == "1"
    n = 1 << 26
    src = torch.empty(n, device=dev, dtype=torch.float32).normal_(); dst = torch.empty_like(src)
    w.sync_torch()
    L.wspr_calib_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert L.wspr_calib_copy(src.data_ptr(), dst.data_ptr(), n, 2) == 0
    t0 = time.perf_counter()
    assert L.wspr_calib_copy(src.data_ptr(), dst.data_ptr(), n, 10) == 0
    gbs = 10 * 8.0 * n / (time.perf_counter() - t0) / 1e9
    assert 1500.0 < gbs < 8000.0, gbs
    assert torch.equal(src, dst)
    L.wspr_calib_copy16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    cms = C.c_double(0.0)
    for variant in (0, 1, 2):                                  # the tuned copy under its three cache policies
        dst.zero_()
        w.sync_torch()
        assert L.wspr_calib_copy16(src.data_ptr(), dst.data_ptr(), n, 5, variant, C.addressof(cms)) == 0
        gbs = 8.0 * n / (cms.value * 1e-3) / 1e9
        assert 2500.0 < gbs < 8000.0, (variant, gbs)
        assert torch.equal(src, dst)
    odd = n - 4 * 37                                           # a length the unrolled loop does not divide
    dst.zero_(); w.sync_torch()
    assert L.wspr_calib_copy16(src.data_ptr(), dst.data_ptr(), odd, 1, 0, None) == 0
    assert torch.equal(src[:odd], dst[:odd]) and not dst[odd:].any()
    raw = torch.randint(1, 256, (4, bench.RAW_BYTES), device=dev, dtype=torch.uint8)
    w.sync_torch()
    ms = (C.c_double * 1)()
    assert L.wspr_calib_read(raw.data_ptr(), bench.RAW_BYTES, 4, 5, C.addressof(ms)) >= 0
    read_gbs = 4 * bench.RAW_BYTES / (ms[0] * 1e-3) / 1e9
    assert 2000.0 < read_gbs < 8000.0, read_gbs
    tf = C.c_double(0.0)
    L.wspr_calib_valu.argtypes = [C.c_int, C.c_void_p]
    assert L.wspr_calib_valu(10, C.addressof(tf)) == 0
    assert 30.0 < tf.value <= 78.7, tf.value                  # the no-FMA vector bound is 78.6 TF/s


@pytest.mark.gpu
def test_many_receivers_one_slot_decoded_together(env):
    """SURVEY 8 f4, "a many-receiver service": the completed buffers of several receiver sessions go through
    wspr_session_decode_many() -- one batch call per distinct set of decoder options -- and every receiver gets the
    spots, the return flag and the residual buffer wspr_session_decode() gives it alone (rtlsdr_wsprd.c:263-328 per
    receiver).  Four receivers: two on one band, one on another, one that has only just started (too short)."""
    torch, bench, w, dev = env
    L = w.lib()
    L.wspr_session_create.restype = C.c_void_p
    L.wspr_session_create.argtypes = [w.decoder_options]
    L.wspr_session_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.wspr_session_rollover.argtypes = [C.c_void_p]
    L.wspr_session_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.wspr_session_decode_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.wspr_session_samples.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.wspr_session_samples.restype = C.POINTER(C.c_float)
    L.wspr_session_destroy.argtypes = [C.c_void_p]
    raw, exp = bench.synth_raw_gpu(3, 24680, dev, -17.0)
    streams = [raw[k].cpu().numpy() for k in range(3)] + [raw[0][: 65536 * 64].cpu().numpy()]
    del raw
    torch.cuda.empty_cache()
    dials = [14095600, 14095600, 7038600, 14095600]
    K = 50

    def receivers():
        out = []
        for host, dial in zip(streams, dials):
            s = L.wspr_session_create(w.default_options(freq=dial))
            for pos in range(0, host.size, 1 << 26):                    # any chunking gives the same stream (tested elsewhere)
                chunk = np.ascontiguousarray(host[pos:pos + (1 << 26)])
                assert L.wspr_session_feed(s, ol.ptr(chunk), chunk.size) >= 0
            assert L.wspr_session_rollover(s) == 0
            out.append(s)
        return out

    def buffer_of(s):
        return [np.ctypeslib.as_array(L.wspr_session_samples(s, 0, rail), shape=(NS,)).copy() for rail in (0, 1)]

    alone, together = receivers(), receivers()
    want = []
    for s in alone:
        res = (w.decoder_results * K)()
        n = C.c_int(0)
        rc = L.wspr_session_decode(s, 0, res, C.byref(n))
        want.append((rc, [bytes(res[i].message) for i in range(n.value)], [(res[i].freq, res[i].snr, res[i].dt, res[i].cycles) for i in range(n.value)], buffer_of(s)))
    arr = (C.c_void_p * 4)(*together)
    bufs = (C.c_int * 4)(0, 0, 0, 0)
    res = (w.decoder_results * (4 * K))()
    nres = (C.c_int * 4)()
    flags = (C.c_int * 4)()
    assert L.wspr_session_decode_many(arr, bufs, 4, res, K, nres, flags) == 3
    for k, s in enumerate(together):
        rc, msgs, nums, buf = want[k]
        assert flags[k] == rc and nres[k] == len(msgs)
        assert [bytes(res[k * K + i].message) for i in range(nres[k])] == msgs
        assert [(res[k * K + i].freq, res[k * K + i].snr, res[k * K + i].dt, res[k * K + i].cycles) for i in range(nres[k])] == nums
        got = buffer_of(s)
        assert np.array_equal(got[0], buf[0]) and np.array_equal(got[1], buf[1]), k
    assert [want[k][0] for k in range(4)] == [1, 1, 1, 0]
    for k in range(3):
        assert any(exp[k][0].encode() in m for m in want[k][1]), (k, exp[k], want[k][1])
    assert abs(want[2][2][0][0] - want[0][2][0][0]) > 5.0               # 40 m against 20 m: the dial enters the reported MHz
    for s in alone + together:
        L.wspr_session_destroy(s)


@pytest.mark.gpu
def test_many_receivers_with_hashtable_keep_index_order_across_option_groups(env, tmp_path):
    """Three receivers with usehashtable and options A, B, A: receiver 1 (the other band) sends the type-2 message whose
    call receiver 2's type-3 "<call>" needs.  The hash memory must see the receivers in index order 0, 1, 2 -- as three
    wspr_session_decode() calls would -- although 0 and 2 share their options (folding them into one batch call ahead of
    receiver 1 would leave "<...>" in receiver 2's spot)."""
    torch, bench, w, dev = env
    L = w.lib()
    L.wspr_session_create.restype = C.c_void_p
    L.wspr_session_create.argtypes = [w.decoder_options]
    L.wspr_session_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.wspr_session_rollover.argtypes = [C.c_void_p]
    L.wspr_session_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.wspr_session_decode_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.wspr_session_destroy.argtypes = [C.c_void_p]
    texts = ["K1JT FN20 20", "PJ4/K1ABC 37", "<PJ4/K1ABC> FK52UD 37"]
    raw, _ = bench.synth_raw_gpu(3, 13579, dev, -12.0, messages=texts)
    streams = [raw[k].cpu().numpy() for k in range(3)]
    del raw
    torch.cuda.empty_cache()
    dials = [14095600, 7038600, 14095600]
    K = 50

    def receivers():
        out = []
        for host, dial in zip(streams, dials):
            o = w.default_options(freq=dial)
            o.usehashtable = 1
            s = L.wspr_session_create(o)
            for pos in range(0, host.size, 1 << 26):
                chunk = np.ascontiguousarray(host[pos:pos + (1 << 26)])
                assert L.wspr_session_feed(s, ol.ptr(chunk), chunk.size) >= 0
            assert L.wspr_session_rollover(s) == 0
            out.append(s)
        return out

    def in_dir(d, fn):
        cwd = os.getcwd()
        d.mkdir()
        os.chdir(d)
        try:
            return fn(), open("hashtable.txt").read()
        finally:
            os.chdir(cwd)

    def one_by_one():
        got = []
        for s in receivers():
            res = (w.decoder_results * K)(); n = C.c_int(0)
            assert L.wspr_session_decode(s, 0, res, C.byref(n)) == 1
            got.append([bytes(res[i].message).split(b"\0")[0].decode() for i in range(n.value)])
            L.wspr_session_destroy(s)
        return got

    def together():
        ss = receivers()
        arr = (C.c_void_p * 3)(*ss); bufs = (C.c_int * 3)(0, 0, 0)
        res = (w.decoder_results * (3 * K))(); nres = (C.c_int * 3)()
        assert L.wspr_session_decode_many(arr, bufs, 3, res, K, nres, None) == 3
        got = [[bytes(res[k * K + i].message).split(b"\0")[0].decode() for i in range(nres[k])] for k in range(3)]
        for s in ss:
            L.wspr_session_destroy(s)
        return got
    want, wf = in_dir(tmp_path / "serial", one_by_one)
    got, gf = in_dir(tmp_path / "many", together)
    assert want[0] == ["K1JT FN20 20"] and want[1] == ["PJ4/K1ABC 37"] and want[2] == ["<PJ4/K1ABC> FK52UD 37"], want
    assert got == want and gf == wf


@pytest.mark.gpu
def test_two_receivers_fed_from_two_threads(env):
    """A service has one RX thread per receiver (rtlsdr_wsprd.c:1136-1151 per dongle).  Every session's front end runs on
    the library's one reserved lane, so two threads feeding two sessions at once must take turns at it: each session's
    buffer must hold exactly the oracle's stream for ITS bytes (round 5: the lane's context was shared unguarded)."""
    import threading
    torch, bench, w, dev = env
    L = w.lib()
    L.wspr_session_create.restype = C.c_void_p
    L.wspr_session_create.argtypes = [w.decoder_options]
    L.wspr_session_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.wspr_session_fill.argtypes = [C.c_void_p, C.c_int]
    L.wspr_session_fill.restype = C.c_uint32
    L.wspr_session_samples.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.wspr_session_samples.restype = C.POINTER(C.c_float)
    L.wspr_session_destroy.argtypes = [C.c_void_p]
    CB, NCB = 65536, 400
    O = ol.lib()
    hosts, want = [], []
    for seed in (7, 8, 9):
        host = np.random.default_rng(seed).integers(0, 256, CB * NCB, dtype=np.uint8)
        ost = O.orc_decim_new()
        oi, oq = np.zeros(NS, np.float32), np.zeros(NS, np.float32)
        nout = O.orc_decim_feed(C.c_void_p(ost), ol.ptr(host), host.size, ol.ptr(oi), ol.ptr(oq), 0, NS)
        O.orc_decim_free(C.c_void_p(ost))
        hosts.append(host)
        want.append((nout, oi, oq))
    sessions = [L.wspr_session_create(w.default_options()) for _ in hosts]
    errs = []

    def rx(k):
        for c in range(NCB):
            chunk = np.ascontiguousarray(hosts[k][c * CB:(c + 1) * CB])
            if L.wspr_session_feed(sessions[k], ol.ptr(chunk), chunk.size) < 0:
                errs.append((k, c))
    threads = [threading.Thread(target=rx, args=(k,)) for k in range(len(hosts))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs
    for k, s in enumerate(sessions):
        nout, oi, oq = want[k]
        assert L.wspr_session_fill(s, 0) == nout
        gi = np.ctypeslib.as_array(L.wspr_session_samples(s, 0, 0), shape=(NS,))
        gq = np.ctypeslib.as_array(L.wspr_session_samples(s, 0, 1), shape=(NS,))
        assert np.array_equal(gi[:nout], oi[:nout]) and np.array_equal(gq[:nout], oq[:nout]), k
        L.wspr_session_destroy(s)


@pytest.mark.gpu
def test_threads_that_never_bound_a_lane_take_turns(env):
    """The library is not re-entrant within a lane (neither is the reference, wsprd.c:81, :133) and every thread sits on
    lane 0 until it binds another.  Two such threads calling at once used to share lane 0's context silently; since round 5
    their calls take turns.  Three unbound threads decode three different batches four times each, concurrently: every
    result equals the same batch decoded alone."""
    import threading
    torch, bench, w, dev = env
    L = w.lib()
    K, nseg = 16, 96
    opt = w.default_options()
    batches = []
    for seed in (11, 12, 13):
        I, Q, _ = bench.synth_batch_gpu(nseg, seed, dev, 3, -12.0, -24.0, 0.5)
        batches.append((I.cpu().numpy()[:, :NS].copy(), Q.cpu().numpy()[:, :NS].copy()))
    torch.cuda.synchronize()

    def decode(k):
        I, Q = batches[k]
        out = (w.decoder_results * (nseg * K))()
        n = (C.c_int * nseg)()
        assert L.wspr_decode_batch(ol.ptr(I), ol.ptr(Q), nseg, NS, NS, opt, C.addressof(out), K, C.addressof(n), 0) == 0
        return [[(bytes(out[s * K + i].message), out[s * K + i].freq, out[s * K + i].snr, out[s * K + i].cycles) for i in range(n[s])]
                for s in range(nseg)]
    alone = [decode(k) for k in range(3)]
    assert sum(len(x) for x in alone[0]) > nseg
    got = [[] for _ in range(3)]

    def worker(k):                                   # a fresh thread: bound to lane 0 like every other
        for _ in range(4):
            got[k].append(decode(k))
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(3):
        assert len(got[k]) == 4 and all(g == alone[k] for g in got[k]), k


@pytest.mark.gpu
def test_release_buffers_waits_for_calls_in_flight(env):
    """wspr_release_buffers() frees the work buffers of every lane of the device.  It used to require that no call be in
    flight; since round 5 it takes every lane's turn first.  Two lanes decode in a loop while the main thread releases the
    buffers eight times in between: every decode still returns the spots of the quiet decode."""
    import threading
    import time
    torch, bench, w, dev = env
    L = w.lib()
    L.wspr_release_buffers.restype = C.c_size_t
    K, nseg = 16, 160
    opt = w.default_options()
    I, Q, _ = bench.synth_batch_gpu(nseg, 21, dev, 2, -14.0, -22.0, 0.5)
    Ih, Qh = I.cpu().numpy()[:, :NS].copy(), Q.cpu().numpy()[:, :NS].copy()
    torch.cuda.synchronize()

    def decode():
        out = (w.decoder_results * (nseg * K))()
        n = (C.c_int * nseg)()
        assert L.wspr_decode_batch(ol.ptr(Ih), ol.ptr(Qh), nseg, NS, NS, opt, C.addressof(out), K, C.addressof(n), 0) == 0
        return [[(bytes(out[s * K + i].message), out[s * K + i].cycles) for i in range(n[s])] for s in range(nseg)]
    quiet = decode()
    stop = threading.Event()
    bad = []

    def worker(lane):
        assert L.wspr_bind_thread_lane(lane) == lane
        while not stop.is_set():
            if decode() != quiet:
                bad.append(lane)
    threads = [threading.Thread(target=worker, args=(lane,)) for lane in (1, 2)]
    for t in threads:
        t.start()
    freed = 0
    for _ in range(8):
        time.sleep(0.05)
        freed += L.wspr_release_buffers()
    stop.set()
    for t in threads:
        t.join()
    assert not bad and freed > 0


@pytest.mark.gpu
def test_callbacks_of_many_receivers_fed_together(env):
    """wspr_session_feed_many(): one callback of each of n receivers as one transfer and one launch set.  Five receivers
    with different bytes, 260 callbacks each (the last ones 4 096 bytes long, as at a slot's end), one of them already
    part-way into its buffer through single feeds: every buffer holds exactly the oracle's stream for its bytes, the
    fills agree, duplicates and ragged lengths are refused."""
    torch, bench, w, dev = env
    L = w.lib()
    L.wspr_session_create.restype = C.c_void_p
    L.wspr_session_create.argtypes = [w.decoder_options]
    L.wspr_session_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.wspr_session_feed_many.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    L.wspr_session_fill.argtypes = [C.c_void_p, C.c_int]
    L.wspr_session_fill.restype = C.c_uint32
    L.wspr_session_samples.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.wspr_session_samples.restype = C.POINTER(C.c_float)
    L.wspr_session_destroy.argtypes = [C.c_void_p]
    CB, NCB, NR, LEAD = 65536, 260, 5, 7
    O = ol.lib()
    rng = np.random.default_rng(4242)
    streams = [rng.integers(0, 256, CB * (NCB - 2) + 2 * 4096 + (LEAD * CB if k == 2 else 0), dtype=np.uint8) for k in range(NR)]
    want = []
    for host in streams:
        ost = O.orc_decim_new()
        oi, oq = np.zeros(NS, np.float32), np.zeros(NS, np.float32)
        nout, pos = 0, 0
        sizes = ([CB] * LEAD if host.size > CB * (NCB - 2) + 2 * 4096 else []) + [CB] * (NCB - 2) + [4096, 4096]
        for sz in sizes:                                       # the oracle, callback by callback like the receivers
            chunk = np.ascontiguousarray(host[pos:pos + sz])
            nout = O.orc_decim_feed(C.c_void_p(ost), ol.ptr(chunk), sz, ol.ptr(oi), ol.ptr(oq), nout, NS)
            pos += sz
        O.orc_decim_free(C.c_void_p(ost))
        want.append((nout, oi, oq))
    sessions = [L.wspr_session_create(w.default_options()) for _ in range(NR)]
    offs = [0] * NR
    for _ in range(LEAD):                                      # receiver 2 has been running for a while
        chunk = np.ascontiguousarray(streams[2][offs[2]:offs[2] + CB])
        assert L.wspr_session_feed(sessions[2], ol.ptr(chunk), CB) >= 0
        offs[2] += CB
    arr = (C.c_void_p * NR)(*sessions)
    fills = (C.c_int * NR)()
    for c in range(NCB):
        sz = CB if c < NCB - 2 else 4096
        chunks = [np.ascontiguousarray(streams[k][offs[k]:offs[k] + sz]) for k in range(NR)]
        ptrs = (C.c_void_p * NR)(*[ch.ctypes.data for ch in chunks])
        assert L.wspr_session_feed_many(arr, ptrs, sz, NR, fills) == 0
        for k in range(NR):
            offs[k] += sz
    for k, s in enumerate(sessions):
        nout, oi, oq = want[k]
        assert fills[k] == nout == L.wspr_session_fill(s, 0), k
        gi = np.ctypeslib.as_array(L.wspr_session_samples(s, 0, 0), shape=(NS,))
        gq = np.ctypeslib.as_array(L.wspr_session_samples(s, 0, 1), shape=(NS,))
        assert np.array_equal(gi[:nout], oi[:nout]) and np.array_equal(gq[:nout], oq[:nout]), k
    twice = (C.c_void_p * 2)(sessions[0], sessions[0])
    two = (C.c_void_p * 2)(chunks[0].ctypes.data, chunks[1].ctypes.data)
    assert L.wspr_session_feed_many(twice, two, 4096, 2, None) == -1           # the same receiver twice
    assert L.wspr_session_feed_many(arr, ptrs, 4100, NR, None) == -1           # not a multiple of 16
    for s in sessions:
        L.wspr_session_destroy(s)


@pytest.mark.gpu
def test_a_batch_too_large_for_the_device_fails_cleanly_and_the_next_call_works(env):
    """Thirty million segments cannot be held (5.4 TB of working rows): the very first allocation fails, the call returns
    -1 with every n_results zero and nothing launched -- and the context is usable afterwards (a failed allocation used to
    leave a buffer that remembered its old size with no memory behind it)."""
    torch, bench, w, dev = env
    L = w.lib()
    L.wspr_set_thread_slots(1)
    K, nseg = 8, 48
    I, Q, _ = bench.synth_batch_gpu(nseg, 31, dev, 1, -16.0, -16.0, 0.5)
    torch.cuda.synchronize()
    out = (w.decoder_results * (nseg * K))()
    n = (C.c_int * nseg)()
    opt = w.default_options()
    assert L.wspr_decode_batch_device(I.data_ptr(), Q.data_ptr(), nseg, NS, I.stride(0), opt, C.addressof(out), K, C.addressof(n)) == 0
    good = [[bytes(out[s * K + i].message) for i in range(n[s])] for s in range(nseg)]
    assert sum(len(g) for g in good) >= nseg - 2
    huge = 30_000_000
    nh = np.ones(huge, np.int32)
    assert L.wspr_decode_batch_device(I.data_ptr(), Q.data_ptr(), huge, NS, I.stride(0), opt, C.addressof(out), K, ol.ptr(nh)) == -1
    assert not nh.any()
    for _ in range(2):
        assert L.wspr_decode_batch_device(I.data_ptr(), Q.data_ptr(), nseg, NS, I.stride(0), opt, C.addressof(out), K, C.addressof(n)) == 0
        assert [[bytes(out[s * K + i].message) for i in range(n[s])] for s in range(nseg)] == good
    L.wspr_set_thread_slots(0)

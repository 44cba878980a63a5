#!/usr/bin/env python3
"""bench.py -- 2-minute WSPR segments decoded per second on MI355X.

Headline workload (BASELINE.json configs[2], the largest single-GPU configuration): 8 192 synthetic
"wsprsim" segments per GPU, ten overlapping type-1 signals each at SNR -10..-28 dB (2500 Hz reference
bandwidth), deep search on (reference defaults), 45 000 complex f32 samples per segment at 375 sps,
resident in HBM when the timed region starts.  A step = one full decode of the batch: FFT bank, peak
pick, coarse sync, fine sync, soft demodulation (HIP kernels), host Fano decode with the device Fano
tail, coherent subtraction (HIP), second pass, spot records on the host.  With --gpus N every rank
decodes its own 8 192 segments (weak scaling) and the spot records are gathered on rank 0 over RCCL.
--config 2 (configs[1]: 1 024 segments x 1 signal at -20 dB) and --config 5 (configs[4]: raw 2.4 Msps
u8 IQ through the on-GPU decimator) select the other single-GPU configurations; at N=1 the line also
carries configs[1] as the block "secondary".

The timed region is never shorter than --min-seconds (default 3 s): if the requested --steps finish
sooner, the measurement is repeated with proportionally more steps and both numbers are reported.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# The HIP runtime multiplexes a process' streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share
# a queue serialise.  With the GPU to itself this benchmark gains 1-3 % from sixteen (GPU_MAX_HW_QUEUES=16 python
# bench.py: 221 -> 215 ms per configs[2] step in two calls, 220 -> 218 in a third), but more than sixteen queues on one
# device IN TOTAL take turns (24 in one process: 290-357 ms; this process' sixteen plus the sixteen of the
# per_rank_share_of_8 child: 289 ms for the child), so the default is left alone (docs/HISTORY.md section 4).
import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import rtlsdr_wsprd_amd as w            # noqa: E402
from rtlsdr_wsprd_amd import dist as wd  # noqa: E402  (package submodule via the shim's __path__)

NS = 45000
K1_BYTES = 360000 + 4 * 417 * 347            # IQ read once + ps rows 48..464 written
K23_BYTES = 4 * 417 * 347                    # ps read once (the <= 4 000 B of candidates are not counted)
STAGE_BYTES = K1_BYTES + K23_BYTES           # 1 517 592 B per segment per pass
HBM_PEAK_GBS = 8000.0


def synth_batch_gpu(nseg, seed, dev, n_signals=1, snr_hi=-20.0, snr_lo=-20.0, t_jitter=1.0, chunk=256, frac23=0.0, wide=False):
    """Synthetic segments generated on the GPU (tests/synth.py is the numpy twin).
    n_signals = 1: SURVEY config 2 (f0 ~ U(-100,100) Hz, t0 = 2 s +- 1 s).
    n_signals > 1: config 3 (frequency slots across +-100 Hz, SNR linearly snr_hi..snr_lo, t0 +- 0.3 s).
    frac23 > 0: that share of the signals comes from compound-call stations alternating between their type-2 and
    type-3 messages from segment to segment (-H traffic, synth.station_message).
    wide: every signal carries its own message out of the whole type-1 space (synth.message_wide) instead of one of the
    7 600 combinations of synth.message_for -- what the timed batches use, so that no step finds its messages in a cache."""
    import synth
    draw = (lambda r: synth.message_wide(int(r.integers(0, 1 << 62)))) if wide else \
           (lambda r: synth.message_for(int(r.integers(0, 1 << 20))))
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    sigma = float(np.sqrt((375.0 / 2500.0) / 2.0))
    df, dt = 375.0 / 256.0, 1.0 / 375.0
    I = torch.empty(nseg, NS, device=dev, dtype=torch.float32)
    Q = torch.empty(nseg, NS, device=dev, dtype=torch.float32)
    expected = []
    sym_cache = {}
    ar = torch.arange(162 * 256, device=dev)
    for c0 in range(0, nseg, chunk):
        n = min(chunk, nseg - c0)
        if frac23 > 0.0:
            msgs = [[synth.station_message(int(rng.integers(0, 8)), c0 + r) if rng.random() < frac23
                     else draw(rng) for _ in range(n_signals)] for r in range(n)]
        else:
            msgs = [[draw(rng) for _ in range(n_signals)] for _ in range(n)]
        for row in msgs:
            for m in row:
                if m not in sym_cache:
                    sym_cache[m] = w.get_wspr_channel_symbols(m)[1].astype(np.float64)
        sym = np.stack([np.stack([sym_cache[m] for m in row]) for row in msgs])            # [n,S,162]
        if n_signals == 1:
            f0 = rng.uniform(-100.0, 100.0, (n, 1))
            snr = np.full((n, 1), snr_hi)
        else:
            f0 = np.linspace(-100.0, 100.0, n_signals)[None, :] + rng.uniform(-2.0, 2.0, (n, n_signals))
            snr = np.linspace(snr_hi, snr_lo, n_signals)[None, :].repeat(n, 0)
        t0 = 2.0 + rng.uniform(-t_jitter, t_jitter, (n, n_signals))
        amp = torch.from_numpy(10.0 ** (snr / 20.0)).to(dev)
        dphi = 2.0 * np.pi * dt * (f0[:, :, None] + (sym - 1.5) * df)                       # [n,S,162]
        Ic = torch.randn(n, NS, device=dev, generator=g, dtype=torch.float32) * sigma
        Qc = torch.randn(n, NS, device=dev, generator=g, dtype=torch.float32) * sigma
        for k in range(n_signals):
            dphi_t = torch.from_numpy(dphi[:, k, :]).to(dev).repeat_interleave(256, dim=1)   # [n,41472] f64
            phi = torch.cumsum(dphi_t, dim=1) - dphi_t
            start = torch.from_numpy(np.round(t0[:, k] / dt).astype(np.int64)).to(dev)
            idx = start[:, None] + ar[None, :]
            ok = (idx >= 0) & (idx < NS)
            idx = idx.clamp(0, NS - 1)
            a = amp[:, k:k + 1]
            Ic.scatter_add_(1, idx, (a * torch.cos(phi)).float() * ok)
            Qc.scatter_add_(1, idx, (a * torch.sin(phi)).float() * ok)
        peak = torch.maximum(Ic.abs().amax(dim=1), Qc.abs().amax(dim=1)).clamp_min(1e-24)
        scale = (0.5 / peak.double()).float()[:, None]
        I[c0:c0 + n] = Ic * scale
        Q[c0:c0 + n] = Qc * scale
        expected += [[synth.expected_text(m) for m in row] for row in msgs]
    return I, Q, expected


RAW_BYTES = 576_000_000                      # 120 s x 2.4 Msps x 2 bytes


def synth_raw_gpu(nseg, seed, dev, snr_db=-20.0, noise_lsb=10.0, amp_lsb=None, messages=None):
    """Config-5 raw segments: unsigned 8-bit interleaved I/Q at 2.4 Msps.  The config-2 baseband
    frame (375 sps) is held for 6400 samples, moved to -600 kHz (the tuner sits fs/4 above the band:
    rtlsdr_wsprd.c:1112; the receiver's (1, j, -1, -j) mixer brings it back), buried in wide-band
    noise of `noise_lsb` LSB rms per rail and quantised around 127.5."""
    import synth
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    nsamp = RAW_BYTES // 2
    raw = torch.empty(nseg, RAW_BYTES, device=dev, dtype=torch.uint8)
    df, dt = 375.0 / 256.0, 1.0 / 375.0
    # in-band noise power in 2500 Hz of complex noise with sigma per rail at 2.4 Msps
    n2500 = 2.0 * noise_lsb ** 2 * 2500.0 / 2.4e6
    amp = float(np.sqrt(n2500 * 10.0 ** (snr_db / 10.0))) if amp_lsb is None else float(amp_lsb)
    expected = []
    chunk = 6400 * 1500                                   # 9.6 M samples per chunk
    for s in range(nseg):
        msg = synth.message_for(int(rng.integers(0, 1 << 20)))
        if messages is not None:                            # the caller's texts (any type) instead of the drawn type-1 ones
            msg = messages[s]
        sym = w.get_wspr_channel_symbols(msg)[1].astype(np.float64)
        f0 = rng.uniform(-100.0, 100.0)
        t0 = 2.0 + rng.uniform(-1.0, 1.0)
        dphi = np.repeat(2.0 * np.pi * dt * (f0 + (sym - 1.5) * df), 256)
        phi = np.concatenate(([0.0], np.cumsum(dphi)[:-1]))
        bb = np.zeros(45000, np.complex128)
        start = int(round(t0 / dt))
        bb[start:start + 41472] = amp * np.exp(1j * phi)[:max(0, min(41472, 45000 - start))]
        bbr = torch.from_numpy(np.ascontiguousarray(bb.real)).float().to(dev)
        bbi = torch.from_numpy(np.ascontiguousarray(bb.imag)).float().to(dev)
        out = raw[s].view(nsamp, 2)
        for c0 in range(0, nsamp, chunk):
            n = torch.arange(c0, min(nsamp, c0 + chunk), device=dev)
            m = torch.div(n, 6400, rounding_mode="floor")
            xr, xi = bbr[m], bbi[m]
            ph = n & 3                                      # x * (-j)^n
            zr = torch.where(ph == 0, xr, torch.where(ph == 1, xi, torch.where(ph == 2, -xr, -xi)))
            zi = torch.where(ph == 0, xi, torch.where(ph == 1, -xr, torch.where(ph == 2, -xi, xr)))
            nz = torch.randn(n.numel(), 2, device=dev, generator=g) * noise_lsb
            out[c0:c0 + n.numel(), 0] = (127.5 + zr + nz[:, 0]).round().clamp(0, 255).to(torch.uint8)
            out[c0:c0 + n.numel(), 1] = (127.5 + zi + nz[:, 1]).round().clamp(0, 255).to(torch.uint8)
        expected.append([synth.expected_text(msg)])
    return raw, expected


def usable_cpus():
    """Hardware threads capped by the cgroup CPU quota (the GPU box gives each slot a slice)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        pass
    return n


K4_FLOP = 33 * 162 * 256 * 32                 # mode-0 lag scan per candidate: 43.8 MFLOP
K41_FLOP = 4 * 162 * 256 * 32                 # mode-1 frequency scan per candidate: the reference sums five hypotheses
                                              # (6.6 MFLOP); the kernel sums FOUR (5.3 MFLOP) and copies the centre one from
                                              # the lag scan, so only four are credited
K7_FLOP = 64.3e6                              # subtract_signal2 per decoded signal
VALU_PEAK_TF = 157.3                          # fp32 vector peak (FMA counted as 2), MI355X_MICROARCH.md
VALU_NOFMA_TF = 78.6                          # the same pipes issuing separately rounded mul / add


def cpu_baseline(I_host, Q_host, expected, budget_s=25.0):
    """Oracle (CPU restatement, oracle/liboracle.so) on a bounded sample of the SAME segments."""
    import oracle_lib as ol
    from concurrent.futures import ThreadPoolExecutor
    ol.lib()
    cores = usable_cpus()
    t = time.perf_counter()
    ol.decode(I_host[0], Q_host[0], NS)
    one = time.perf_counter() - t                       # single-core seconds per segment
    n = int(min(len(I_host), max(cores, budget_s * cores / max(one, 1e-3) * 0.5)))
    n = max(1, min(n, len(I_host)))

    def run(s):
        spots, _, _ = ol.decode(I_host[s], Q_host[s], NS)
        return [x.message.decode() for x in spots]

    t = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:               # ctypes releases the GIL
        res = list(ex.map(run, range(n)))
    wall = time.perf_counter() - t
    return {"value": n / wall, "unit": "segments/s", "cores": cores, "kind": "port",
            "sample": "%d of the benchmarked segments, oracle/liboracle.so (gcc -O3, scalar C), %d threads = the "
                      "CPUs this container may use (cgroup quota; host has %d hw threads); single core %.1f "
                      "segments/s" % (n, cores, os.cpu_count() or 0, 1.0 / one)}, res


def measure_k1_traffic(nseg, nsig, timeout_s=150):
    """HBM bytes of the FFT+sync stage's kernels from the PMC counters, measured NOW: two rocprofv3 passes (FETCH_SIZE,
    then WRITE_SIZE -- separate passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes) of tools/pmc_k1.py (a
    1 GiB calibration copy + five launches of K1/K2/K3 on nseg resident segments) in child processes, summarised by
    tools/pmc_summarise.py (gfx950: FETCH_SIZE counts half of the bytes read; calibrated on the copy).  Returns the
    summary dict or None (no rocprofv3, a pass failed or timed out: the caller falls back to the round's profile)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None
    tmp = tempfile.mkdtemp(prefix="wspr_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    t_end = time.time() + timeout_s
    try:
        csvs = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            r = subprocess.run([rp, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable,
                                os.path.join(ROOT, "tools", "pmc_k1.py"), str(nseg), str(nsig)], cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=max(10.0, t_end - time.time()))
            found = glob.glob(os.path.join(out, "*", "*counter_collection.csv"))
            if r.returncode != 0 or not found:
                return None
            csvs[counter] = found[0]
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summarise.py"), csvs["FETCH_SIZE"], csvs["WRITE_SIZE"],
                            str(nseg)], capture_output=True, text=True, timeout=60)
        return json.loads(r.stdout) if r.returncode == 0 else None
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def k0_report(L, m, with_cpu=True):
    """(L: the LAB library -- the timing and calibration entry points are not in the product.)  Roofline of the front end (K0, the HBM-bound kernel of configs[4]) on the resident raw segments of measurement
    `m`, HIP events on the launch stream, and the decimator's CPU baseline (the oracle's restatement of
    rtlsdr_callback(), rtlsdr_wsprd.c:126-244, over one whole 576 MB segment per host thread)."""
    raw, nraw = m["raw"], m["raw"].shape[0]
    I, Q = m["I"], m["Q"]
    kms = (C.c_double * 1)()
    # 5 launches to settle, then three sets of 10 (all reported, the best one is the figure)
    L.wspr_bench_decimate(raw.data_ptr(), RAW_BYTES, nraw, I.data_ptr(), Q.data_ptr(), 5, C.addressof(kms))
    k0_sets = []
    for _ in range(3):
        L.wspr_bench_decimate(raw.data_ptr(), RAW_BYTES, nraw, I.data_ptr(), Q.data_ptr(), 10, C.addressof(kms))
        k0_sets.append(kms[0])
    best = min(k0_sets)
    k0_bytes = (RAW_BYTES + 360000) * nraw
    rms = (C.c_double * 1)()
    L.wspr_calib_read(raw.data_ptr(), RAW_BYTES, nraw, 10, C.addressof(rms))
    out = {"bound": "hbm", "kernel": "cic_block_sums_mfma_kernel + cic_comb_fir_kernel + normalise_kernel",
           "segments_per_launch": nraw, "avg_launch_ms": best, "bytes_per_launch": k0_bytes,
           "achieved_GBs": k0_bytes / (best * 1e-3) / 1e9, "peak_GBs": HBM_PEAK_GBS,
           "frac": k0_bytes / (best * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "segments_per_second_bound_at_this_rate": nraw / (best * 1e-3),
           "avg_launch_ms_of_each_set_of_10": k0_sets,
           # the same rows read by a kernel with K0's access pattern and no arithmetic
           "measured_read_GBs": RAW_BYTES * nraw / (rms[0] * 1e-3) / 1e9}
    if with_cpu:
        import oracle_lib as ol
        from concurrent.futures import ThreadPoolExecutor
        O = ol.lib()
        cores = usable_cpus()
        host = raw[0].cpu().numpy()                              # one raw segment; every thread decimates it (read-only)

        def one(full):
            st = O.orc_decim_new()
            oi = np.zeros(NS, np.float32); oq = np.zeros(NS, np.float32)
            n = O.orc_decim_feed(C.c_void_p(st), ol.ptr(host), host.size, ol.ptr(oi), ol.ptr(oq), 0, NS)
            O.orc_decim_free(C.c_void_p(st))
            if full:                                             # ... and the decoder thread's share (:284-317)
                O.orc_normalise(ol.ptr(oi), ol.ptr(oq), C.c_int(n), C.c_int(NS))
                ol.decode(oi, oq, NS)
            return n
        t0 = time.perf_counter()
        one(False)
        single = time.perf_counter() - t0
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            outs = list(ex.map(one, [False] * cores))
        wall = time.perf_counter() - t0
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(one, [True] * cores))
        wall_full = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": cores / wall_full, "unit": "segments/s", "cores": cores, "kind": "port",
                               "decimator_only_segments_per_s": cores / wall,
                               "decimator_MBs_per_core_alone": RAW_BYTES / single / 1e6, "outputs_per_segment": int(outs[0]),
                               "sample": "one 576 MB raw segment per host thread (%d threads) through oracle/liboracle.so: "
                                         "orc_decim_feed (scalar C restatement of rtlsdr_callback, rtlsdr_wsprd.c:126-244; the "
                                         "reference's README quotes this stage as its 'RX load'), then normalise + decode "
                                         "(:284-317)" % cores}
    return out


def reference_case_block(with_cpu):
    """BASELINE configs[0] / the reference's only published metric (README.md:138-151: the decode burst per 2-minute
    segment, 0.5 s on an i7-5820K): ONE wspr_decode() call on the reference's own signal file through the host-buffer
    entry point -- H2D, decode, residual and spots back -- as rtlsdr_wsprd.c:316 calls it once every 120 s."""
    # the product's own reader and print format (rtlsdr_wsprd.c:555-592, 691-701); the oracle is touched by the CPU leg only
    L = w.lib()
    I = np.zeros(NS, np.float32)
    Q = np.zeros(NS, np.float32)
    n = int(L.wspr_read_iq_file(os.path.join(ROOT, "tests", "golden", "refSignalSnr0dB.iq").encode(),
                                I.ctypes.data_as(C.c_void_p), Q.ctypes.data_as(C.c_void_p)))
    assert n == NS, "reference signal file not readable"

    def line(spot):
        buf = C.create_string_buffer(128)
        L.wspr_format_spot(C.byref(spot), buf, C.c_size_t(128))
        return buf.value.decode().rstrip("\n")
    opt = w.default_options()
    for _ in range(3):
        spots, _, _ = w.wspr_decode(I, Q, n, opt)
    times = []
    for _ in range(20):
        t0 = time.perf_counter()
        spots, _, _ = w.wspr_decode(I, Q, n, opt)
        times.append(time.perf_counter() - t0)
    times.sort()
    blk = {"workload": "configs[0]: signals/refSignalSnr0dB.iq, one wspr_decode() call (host buffers in, residual and "
                       "spots out), warm, median of 20",
           "ms_per_call": 1e3 * times[len(times) // 2], "ms_min": 1e3 * times[0], "ms_max": 1e3 * times[-1],
           "spots": [line(x) for x in spots],
           "published_reference": {"ms_per_call": 500.0, "hardware": "i7-5820K, one core, FFTW (reference README.md:138-151)",
                                   "note": "context only: other hardware, never a vs_baseline"}}
    if with_cpu:
        import oracle_lib as ol
        t0 = time.perf_counter()
        for _ in range(5):
            ref, _, _ = ol.decode(I, Q, n)
        blk["cpu_oracle_one_core_ms_per_call"] = 1e3 * (time.perf_counter() - t0) / 5
        blk["equal_to_oracle"] = [ol.spot_line(x) for x in ref] == blk["spots"]
    return blk


def child_bench(args, config, steps, warmup, inflight=None, cpu_share=None, spawn=False, timeout=600):
    """One more bench.py run in a child process (slim: no baselines, no extra blocks) and its parsed JSON line:
    with cpu_share the child is pinned to that many CPUs (--cpu-share), with spawn it runs as one rank under
    torch.distributed.run with the RCCL process group (--spawn)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "WSPR_HOST_THREADS",
                                                             "OMP_NUM_THREADS")}
    cmd = [sys.executable, os.path.abspath(__file__), "--config", str(config), "--steps", str(steps), "--warmup", str(warmup),
           "--no-cpu-baseline", "--no-secondary", "--no-tertiary", "--no-pmc", "--no-share-block", "--no-reference-case",
           "--no-ceilings", "--no-shard-block", "--no-host-entry", "--no-hashtable-block", "--no-kernel-roofline",
           "--no-warm-extra"] + (["--rotate", str(args.rotate)] if args.rotate else [])
    if inflight:
        cmd += ["--inflight", str(inflight)]
    if cpu_share:
        cmd += ["--cpu-share", str(cpu_share)]
    if spawn:
        cmd += ["--spawn"]
    if args.segments:
        cmd += ["--segments", str(args.segments)]
    if args.slots:
        cmd += ["--slots", str(args.slots)]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "child bench failed (rc %d): %s" % (r.returncode, r.stderr[-300:])}
    d = json.loads(lines[-1])
    d["child_wall_s"] = time.perf_counter() - t0
    return d


def slim(d):
    """The figures of a child line that the blocks of the parent line quote."""
    if "error" in d:
        return d
    return {"value": d["value"], "unit": "segments/s", "ms_per_step": d["ms_per_step"], "steps": d["steps"],
            "batches_in_flight": d["config"]["batches_in_flight"], "slots_per_batch": d["config"]["slots_per_batch"],
            "host_threads": d["host_threads"], "cpus_pinned_to": d.get("cpus_pinned_to"),
            "host_pool_workers": d.get("host_pool_workers"), "decoded_ok": d["decoded_ok"],
            "false_decodes": d["false_decodes"], "gathered_over": d["config"]["gathered_over"],
            "gather_ms_per_step": d.get("gather_ms_per_step"), "gather_cpu_ms_per_step": d.get("gather_cpu_ms_per_step"),
            "message_cache_hit_rate_last_step": d.get("stage_ms_last_step", {}).get("message_cache_hit_rate"),
            "distinct_batches_rotated": d["config"].get("distinct_batches_rotated"),
            "child_wall_s": d["child_wall_s"]}


def share_of_8_block(args, inflight):
    """What ONE rank of an 8-rank job on this host gets: configs[2] again in a child process PINNED to 1/8 of the usable
    CPUs (sched_setaffinity: lane threads, HIP runtime threads and the library's pools all live there), same batches in
    flight.  A SCALE value at N = 8 can be read against 8 x this figure: below it the curve is limited by something
    other than the ranks' CPU shares."""
    share = max(1, usable_cpus() // 8)
    d = child_bench(args, 3, 8, 3, inflight=inflight, cpu_share=share)
    if "error" in d:
        return d
    out = slim(d)
    out.update({"workload": d["config"]["workload"], "of_usable_cpus": usable_cpus(), "expected_8_gpu_aggregate": 8 * d["value"],
                "note": "same GPU, same kernels; the whole process (lane threads, bookkeeping, copies, runtime) pinned to %d CPU(s)" % share})
    return out


def shard_block(args, full_host):
    """configs[3] = 65 536 segments over 8 GPUs has never run (no 8-GPU box in reach of this session): what CAN be measured
    on one GPU is its per-rank shard -- 8 192 single-signal segments -- through the code an 8-rank run takes:
      full_host      the shard with the whole host's CPUs (measured in this process by the caller),
      share_of_8     the same in a child pinned to usable_cpus // 8 CPUs: the host-CPU limit of a rank,
      rccl_world_1_share_of_8   the same as ONE RANK under torch.distributed.run with the RCCL process group: options
                     broadcast, per-step SpotGatherer (staging copy, H2D, gather, D2H on rank 0): gather_ms_per_step
                     (bench.py --config 4 --spawn without the pinning gives the full-host figure: 349-358 k in round 5).
    8 x min(...) is what the sharded run can reach on this host if nothing else limits."""
    share = max(1, usable_cpus() // 8)
    blk = {"workload": full_host["workload"], "full_host": full_host,
           "share_of_8": slim(child_bench(args, 4, 20, 4, cpu_share=share)),
           "rccl_world_1_share_of_8": slim(child_bench(args, 4, 20, 4, cpu_share=share, spawn=True))}
    vals = [v["value"] for v in (blk["full_host"], blk["share_of_8"], blk["rccl_world_1_share_of_8"]) if "value" in v]
    if len(vals) == 3:
        blk["share_of_8_over_full_host"] = blk["share_of_8"]["value"] / blk["full_host"]["value"]
        # what the fan-in costs: in the rank's throughput (with it / without it), in CPU time of the driving thread per
        # step, and -- not a cost but a latency: the thread sleeps while its copies and the collective wait their turn
        # in the GPU's queues -- in that thread's wall time
        blk["with_gather_over_without"] = blk["rccl_world_1_share_of_8"]["value"] / blk["share_of_8"]["value"]
        g = blk["rccl_world_1_share_of_8"]
        if g.get("gather_cpu_ms_per_step") is not None:
            blk["gather_cpu_fraction_of_a_step"] = g["gather_cpu_ms_per_step"] / g["ms_per_step"]
        if g.get("gather_ms_per_step") is not None:
            blk["gather_wall_fraction_of_a_step_of_the_driving_thread"] = g["gather_ms_per_step"] / g["ms_per_step"]
        blk["expected_8_gpu_aggregate_configs3"] = 8 * min(vals)
    return blk


def host_entry_block(dev, lanes, resident):
    """The reference's OWN calling convention at speed: wspr_decode() takes host buffers (wsprd.h:106-111; call sites
    rtlsdr_wsprd.c:316, :689), so does wspr_decode_batch().  configs[1] and configs[2] again with the IQ in CALLER HOST
    MEMORY -- pageable (the library gathers the rows through its pinned chunk ring) and pinned with
    wspr_pin_host_buffer() (plain DMA) -- twelve calls in flight on twelve lanes, next to the resident figure of the same
    process and the box's host-to-device rate (what bounds configs[1]: 360 000 bytes per segment).  Never `value`."""
    L = w.lib()
    out = {"bytes_per_segment": 2 * 4 * NS}
    # the link: one linear pinned -> device copy of 1 GiB, five times
    src = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
    dst = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    out["h2d_GBs"] = 5 * (1 << 30) / (time.perf_counter() - t0) / 1e9
    out["pcie_bound_segments_per_s"] = out["h2d_GBs"] * 1e9 / (2 * 4 * NS)
    del src, dst
    torch.cuda.empty_cache()
    opt = w.default_options()
    inflight = len(lanes)
    for ex in lanes:
        ex.submit(L.wspr_set_thread_slots, 1).result()
    # (configs[2] with HALF its segments per call: the ratio to the resident call is what the block is for, and 2 x 1.5 GB of
    # host rows per variant keep the default command short)
    for name, nseg, nsig, steps, K in (("configs1", 1024, 1, 96, 16), ("configs2", 4096, 10, 12, 32)):
        if nsig == 1:
            I, Q, _ = synth_batch_gpu(nseg, 1234, dev, 1, -20.0, -20.0, 1.0, wide=True)
        else:
            I, Q, _ = synth_batch_gpu(nseg, 4321, dev, 10, -10.0, -28.0, 0.3, wide=True)
        torch.cuda.synchronize()
        Ih, Qh = I.cpu().numpy(), Q.cpu().numpy()
        outs = [((w.decoder_results * (nseg * K))(), (C.c_int * nseg)()) for _ in range(inflight)]

        def run(fn, n):
            pend = []
            for s in range(n):
                if len(pend) >= inflight:
                    pend.pop(0).result()
                pend.append(lanes[s % inflight].submit(fn, s % inflight))
            for f in pend:
                f.result()

        def timed(fn):
            run(fn, inflight)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(fn, steps)
            el = time.perf_counter() - t0
            return {"value": nseg * steps / el, "unit": "segments/s", "ms_per_step": 1e3 * el / steps, "steps": steps,
                    "iq_GBs": 2 * 4 * NS * nseg * steps / el / 1e9}

        def resident_call(k):
            o, n = outs[k]
            assert L.wspr_decode_batch_device(I.data_ptr(), Q.data_ptr(), nseg, NS, I.stride(0), opt, C.addressof(o), K, C.addressof(n)) == 0

        def host_call(k):
            o, n = outs[k]
            assert L.wspr_decode_batch(Ih.ctypes.data_as(C.c_void_p), Qh.ctypes.data_as(C.c_void_p), nseg, NS, NS, opt,
                                       C.addressof(o), K, C.addressof(n), 0) == 0
        blk = {"segments": nseg, "signals_per_segment": nsig, "batches_in_flight": inflight}
        blk["resident"] = timed(resident_call)
        want = list(outs[0][1])
        del I, Q
        torch.cuda.empty_cache()
        blk["host_pageable"] = timed(host_call)
        same = list(outs[0][1]) == want
        assert L.wspr_pin_host_buffer(Ih.ctypes.data_as(C.c_void_p), Ih.nbytes) == 0
        assert L.wspr_pin_host_buffer(Qh.ctypes.data_as(C.c_void_p), Qh.nbytes) == 0
        blk["host_pinned"] = timed(host_call)
        same = same and list(outs[0][1]) == want
        L.wspr_unpin_host_buffer(Ih.ctypes.data_as(C.c_void_p))
        L.wspr_unpin_host_buffer(Qh.ctypes.data_as(C.c_void_p))
        blk["spot_counts_equal_resident"] = same
        for kind in ("host_pageable", "host_pinned"):
            blk[kind]["of_resident"] = blk[kind]["value"] / blk["resident"]["value"]
            blk[kind]["of_pcie_bound"] = blk[kind]["value"] / out["pcie_bound_segments_per_s"]
        out[name] = blk
        del Ih, Qh, outs
        L.wspr_release_buffers()
    return out


def hashtable_block(dev, lanes):
    """usehashtable (-H, what real traffic needs: SURVEY 8 f3) on a BATCH.  Rounds 2-4 decoded such a batch one segment at
    a time; since round 5 it is decoded in parallel against a logged view of the ordered hash memory and only the
    segments whose look-ups would have seen something else are decoded again.  configs[2]-shaped
    traffic, 5 % of the signals type 2 / type 3: the same batch without the option, with it from an EMPTY
    hashtable.txt (every hashed call of the batch resolves through an earlier segment of the batch: the worst case) and
    again with the file the first call wrote (a service's steady state).  One call at a time: calls with the option
    take turns by definition (each reads the file the previous one wrote)."""
    import shutil
    import tempfile
    L = w.lib()
    nseg, K = 2048, 32                       # a quarter of configs[2]'s batch: same traffic, a short block
    I, Q, _ = synth_batch_gpu(nseg, 2468, dev, 10, -10.0, -28.0, 0.3, frac23=0.05)
    torch.cuda.synchronize()
    out = (w.decoder_results * (nseg * K))()
    nres = (C.c_int * nseg)()
    lane = lanes[0]
    lane.submit(L.wspr_set_thread_slots, 0).result()            # a lone call: the library's own three pipelines

    def call(use):
        o = w.default_options()
        o.usehashtable = use
        t0 = time.perf_counter()
        assert L.wspr_decode_batch_device(I.data_ptr(), Q.data_ptr(), nseg, NS, I.stride(0), o, C.addressof(out), K, C.addressof(nres)) == 0
        return time.perf_counter() - t0
    cwd, tmp = os.getcwd(), tempfile.mkdtemp(prefix="wspr_hash_", dir="/tmp")
    try:
        os.chdir(tmp)
        for _ in range(3):
            lane.submit(call, 0).result()
        plain = min(lane.submit(call, 0).result() for _ in range(3))
        n_plain = int(sum(nres))
        unresolved_plain = sum(1 for s in range(nseg) for i in range(nres[s]) if out[s * K + i].message.startswith(b"<...>"))
        cold = lane.submit(call, 1).result()
        n_cold = int(sum(nres))
        warm = min(lane.submit(call, 1).result() for _ in range(3))
        unresolved_warm = sum(1 for s in range(nseg) for i in range(nres[s]) if out[s * K + i].message.startswith(b"<...>"))
        entries = sum(1 for _ in open("hashtable.txt"))
        # ... and several such calls IN FLIGHT (twelve lanes, one pipeline each): a call runs its first round ahead of its
        # turn, then takes the file its predecessors wrote as its base and decodes again what that changes
        nfl = min(12, len(lanes))
        outs = [((w.decoder_results * (nseg * K))(), (C.c_int * nseg)()) for _ in range(nfl)]
        for ex in lanes[:nfl]:
            ex.submit(L.wspr_set_thread_slots, 1).result()

        def call_on(k, use):
            o = w.default_options()
            o.usehashtable = use
            oo, nn = outs[k]
            assert L.wspr_decode_batch_device(I.data_ptr(), Q.data_ptr(), nseg, NS, I.stride(0), o, C.addressof(oo), K, C.addressof(nn)) == 0

        def flight(use, n):
            pend = []
            for s in range(n):
                if len(pend) >= nfl:
                    pend.pop(0).result()
                pend.append(lanes[s % nfl].submit(call_on, s % nfl, use))
            for f in pend:
                f.result()
        flight(0, 2 * nfl)                                      # the lanes' contexts and buffers exist, clocks settled
        rates = {}
        for use in (1, 0):
            t0 = time.perf_counter()
            flight(use, 4 * nfl)
            rates[use] = nseg * 4 * nfl / (time.perf_counter() - t0)
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)
    del I, Q
    L.wspr_release_buffers()
    return {"workload": "configs[2]-shaped: %d segments x 10 signals, 5 %% of them type 2 / type 3 (compound-call stations "
                        "alternating between both), one call at a time, three pipelines per call" % nseg,
            "without_option": {"value": nseg / plain, "unit": "segments/s", "ms_per_call": 1e3 * plain, "spots": n_plain,
                               "unresolved_hashed_calls": unresolved_plain},
            "with_option_empty_file": {"value": nseg / cold, "unit": "segments/s", "ms_per_call": 1e3 * cold, "spots": n_cold,
                                       "of_without_option": plain / cold},
            "with_option_file_of_the_previous_call": {"value": nseg / warm, "unit": "segments/s", "ms_per_call": 1e3 * warm,
                                                      "of_without_option": plain / warm,
                                                      "unresolved_hashed_calls": unresolved_warm},
            "calls_in_flight": {"lanes": nfl, "without_option": {"value": rates[0], "unit": "segments/s"},
                                "with_option": {"value": rates[1], "unit": "segments/s", "of_without_option": rates[1] / rates[0]},
                                "note": "the file of the earlier calls is in place (steady state); calls enter in "
                                        "submission order and commit in that order"},
            "hashtable_entries": entries}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: re-executes this script as N ranks of ONE node under
    torch.distributed.run (one rank per GPU, RCCL over xGMI, rendezvous on 127.0.0.1), each rank with its share of
    the host's CPUs.  Fails loudly if the node has fewer than N GPUs."""
    have = torch.cuda.device_count()
    if have < n and not (os.environ.get("WSPR_BENCH_SHARE_GPU") == "1" and have >= 1):
        sys.exit("bench.py: --gpus %d but only %d HIP device(s) visible on this node" % (n, have))
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cpus() // n)))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: what RCCL needs on these hosts
    env["WSPR_BENCH_SPAWNED"] = "1"
    if n == 1:
        env["WSPR_BENCH_FORCE_DIST"] = "1"               # --spawn on a 1-GPU box: still the RCCL code path
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--segments", type=int, default=None, help="segments per GPU (default: the config's)")
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5],
                    help="BASELINE.json configs index: 3 = configs[2] (8192 seg x 10 signals, -10..-28 dB, deep search "
                         "on; the headline: largest single-GPU configuration), 2 = configs[1] (1024 seg x 1 signal, "
                         "-20 dB), 4 = ONE RANK'S SHARD of configs[3] (65 536 segments over 8 GPUs = 8 192 single-signal "
                         "segments per GPU, SURVEY 8d 'config 4 ... as config 2'; with --gpus 8 this IS configs[3]), "
                         "5 = configs[4] (raw 2.4 Msps u8 IQ through the on-GPU decimator; --segments = segments "
                         "per decoder call, default 1024, fed by front-end waves of --raw-segments)")
    ap.add_argument("--raw-segments", type=int, default=64,
                    help="--config 5: distinct raw segments resident in HBM (576 MB each) = one front-end wave")
    ap.add_argument("--cpu-share", type=int, default=None,
                    help="run as ONE rank of a job that gives each rank this many CPUs: the process is pinned to that many "
                         "CPUs (sched_setaffinity) and the library sizes its host side for them (WSPR_HOST_THREADS); "
                         "what the per-rank blocks of the N=1 line use")
    ap.add_argument("--snr", type=float, default=-20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] block of the N=1 line")
    ap.add_argument("--no-tertiary", action="store_true", help="skip the configs[4] block of the N=1 line")
    ap.add_argument("--no-ceilings", action="store_true", help="skip the copy / read / write calibration kernels")
    ap.add_argument("--no-share-block", action="store_true",
                    help="skip the per_rank_share_of_8 block (configs[2] again in a child process with 1/8 of the CPUs)")
    ap.add_argument("--no-reference-case", action="store_true", help="skip the configs[0] block (one wspr_decode() call)")
    ap.add_argument("--no-shard-block", action="store_true",
                    help="skip the configs3_shard block (the per-rank shard of configs[3]: full host, 1/8 of the CPUs, world-1 RCCL)")
    ap.add_argument("--no-host-entry", action="store_true",
                    help="skip the host_entry block (the reference's own calling convention: host buffers in, wsprd.h:106-111)")
    ap.add_argument("--no-hashtable-block", action="store_true", help="skip the usehashtable_batch block (-H on a batch)")
    ap.add_argument("--no-kernel-roofline", action="store_true",
                    help="child runs: skip the kernel-level timing sets behind the roofline block (the line then carries no roofline)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure roofline.traffic with two rocprofv3 --pmc passes (about 40 s); read it from profiles/")
    ap.add_argument("--min-seconds", type=float, default=3.0, help="minimum length of the timed region")
    ap.add_argument("--rotate", type=int, default=None,
                    help="distinct resident batches (other seeds, every signal its own message) a lane rotates through from step "
                         "to step, so that no step finds the previous one's messages in the host's message cache (default: 3, "
                         "or as many as it takes for 24 576 distinct messages -- the cache holds 20 000 -- i.e. 24 batches of "
                         "1 024 single-signal segments); 1 = the round-5 shape (ONE batch of 7 600 possible messages "
                         "decoded on every step: a warm cache)")
    ap.add_argument("--no-warm-extra", action="store_true",
                    help="skip the labelled extra 'warm_cache' (the round-5 shape measured beside the cold figure)")
    ap.add_argument("--fano-fast", type=int, default=None,
                    help="host Fano budget in cycles/bit before an attempt is left to the device tail (configs[2]; "
                         "default 200 with >= 8 CPUs per rank, 60 with 4-7, 25 below; 10000 = no split)")
    ap.add_argument("--inflight", type=int, default=None,
                    help="batches in flight (default: 12; 6 for --config 5): step k+1 starts "
                         "under the tail of step k, each on its own lane of the library")
    ap.add_argument("--inflight-light", type=int, default=None,
                    help="batches in flight for the single-signal workloads (--config 2 / 4) when --inflight is not given")
    ap.add_argument("--slots", type=int, default=None,
                    help="concurrent pipelines per batch inside the library (wspr_set_thread_slots; default: 1 with four or more "
                         "batches in flight -- they already overlap each other -- else the library's own 3)")
    ap.add_argument("--spawn", action="store_true",
                    help="take the launcher path even for --gpus 1 (one rank under torch.distributed.run with the RCCL "
                         "process group, broadcast and gather): how the multi-GPU entry is exercised on a 1-GPU box")
    args = ap.parse_args()
    if args.cpu_share:
        # an honest emulation of a rank's CPU share: the lane threads, the HIP runtime's threads and torch's all
        # live on these CPUs (WSPR_HOST_THREADS alone only sizes the library's pools)
        cpus = sorted(os.sched_getaffinity(0))[:max(1, args.cpu_share)]
        os.sched_setaffinity(0, cpus)
        os.environ["WSPR_HOST_THREADS"] = str(len(cpus))
        os.environ["OMP_NUM_THREADS"] = str(len(cpus))
        torch.set_num_threads(len(cpus))

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        launch_ranks(args.gpus)                          # does not return: re-executes under torch.distributed.run
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python bench.py --gpus N spawns them "
                 "itself; under torch.distributed.run pass --nproc-per-node N)" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    # host Fano pool: share the host's cores between the ranks of this node
    os.environ.setdefault("WSPR_HOST_THREADS", str(max(2, usable_cpus() // max(1, world))))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook for 1-GPU boxes (tests/test_gpu_configs.py): WSPR_BENCH_SHARE_GPU=1 lets the ranks share the devices
    # there are, WSPR_BENCH_BACKEND=gloo carries the collectives (RCCL refuses two ranks on one device)
    if os.environ.get("WSPR_BENCH_SHARE_GPU") == "1":
        local %= max(1, torch.cuda.device_count())
    backend = os.environ.get("WSPR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("WSPR_BENCH_FORCE_DIST") == "1"   # the latter: 1-rank RCCL smoke test
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    assert w.lib().wspr_device_ready() == 1, "HIP extension / device not usable"
    # how many physical devices the ranks really sit on (advisor, round 3: with WSPR_BENCH_SHARE_GPU a two-rank line
    # said n_gpus 2 while one GPU did the work)
    import socket
    try:
        my_dev = (socket.gethostname(), str(torch.cuda.get_device_properties(local).uuid))
    except Exception:
        my_dev = (socket.gethostname(), local)
    devs = [my_dev]
    if use_dist:
        devs = [None] * world
        dist.all_gather_object(devs, my_dev)
    distinct_devices = len(set(devs))
    L = w.lib()

    opt = w.default_options()
    if use_dist:
        opt = wd.broadcast_options(opt, src=0)          # fan-out of the (tiny) job description
    cpus_here = int(os.environ["WSPR_HOST_THREADS"])
    # crowded band (configs[2]): the Fano attempts run on the device (library default for such batches), the host
    # only keeps books, and several batches in flight cover the device round trips of a wave (round 2: 25.5 / 25.9 /
    # 27.2 k segments/s with 2 / 3 / 4 in flight; round 3: 30.7 / 31.8 / 30.9 k with 4 / 6 / 8, three slots each; round 4:
    # ONE slot per batch -- every kernel launch covers the whole batch -- and twelve batches in flight: 231-233 ms per
    # step with 3 slots x 6, 226-230 with 1 x 8, 220-224 with 1 x 12, 221-225 with 1 x 16, one box).  configs[1] the
    # same way: 3.93 ms per 1 024 segments with 3 slots x 3, 3.55-3.65 with 2 x 4, 3.26-3.31 with 1 x 6 or 8,
    # 3.19-3.24 with 1 x 12.  configs[4] does not care (197-201 ms with 3 x 6 or 1 x 6) and holds 147 GB of raw data
    # resident, so it stays at six.  With few CPUs per rank: two in flight, or one.
    def default_inflight(config):
        # twelve calls in flight whatever the rank's CPU share: since round 5 a waiting lane holds no CPU (the library
        # polls briefly, then sleeps), so twelve lane threads cost a 2-CPU rank 20 ms of CPU per 25 ms step of the
        # single-signal shard instead of keeping both CPUs busy doing nothing (rounds 2-4 fell back to 2 or 1 in flight
        # below 8 / 6 CPUs: the cliff the verdict of round 4 pointed at).  configs[4] holds 147 GB of raw data resident.
        if args.inflight_light and config in (2, 4):
            return args.inflight_light
        return 6 if config == 5 else 12
    inflight = max(1, min(args.inflight if args.inflight else default_inflight(args.config), 16))
    from concurrent.futures import ThreadPoolExecutor
    lanes = [ThreadPoolExecutor(1) for _ in range(16 if not args.inflight else inflight)]

    def bind(lane):
        torch.cuda.set_device(local)
        return L.wspr_bind_thread_lane(lane)
    for k, ex in enumerate(lanes):
        assert ex.submit(bind, k).result() == k
    rec = C.sizeof(w.decoder_results)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    lanes_primary = inflight

    def measure(config, nseg, steps, warmup, seed, warm=False):
        """Builds the workload of one configuration and times `steps` steps of it (>= --min-seconds).
        Cold by construction (round 6): --rotate distinct batches, every signal its own message; a lane's consecutive steps
        decode different batches.  warm=True: ONE batch of synth.message_for texts on every step (the round-5 shape)."""
        fast_old = None
        inflight = lanes_primary if config == args.config else (min(args.inflight, 6) if args.inflight else default_inflight(config))
        nsig_of = {2: 1, 3: 10, 4: 1, 5: 1}[config]
        auto_rot = min(32, max(3, -(-24576 // max(1, nseg * nsig_of))))          # more distinct messages than the cache holds
        nrot = 1 if (config == 5 or warm) else max(1, args.rotate if args.rotate else auto_rot)
        wide = nrot > 1
        batches = []                                     # [(I, Q, expected)]: the distinct batches a lane rotates through
        if config in (2, 4):
            for b in range(nrot):
                batches.append(synth_batch_gpu(nseg, 1234 + seed + 7919 * b, dev, 1, args.snr, args.snr, 1.0, wide=wide))
            I, Q, expected = batches[0]
            workload = "configs[1]: %d synthetic wsprsim segments per GPU, 1 signal each, SNR %g dB" % (nseg, args.snr)
            if config == 4:
                workload = ("configs[3], one rank's shard: %d of the 65 536 single-signal segments (8 GPUs x 8 192; SURVEY 8d: "
                            "'config 4 ... as config 2'), SNR %g dB, generated with the rank's seed" % (nseg, args.snr))
            raw = None
        elif config == 3:
            for b in range(nrot):
                batches.append(synth_batch_gpu(nseg, 4321 + seed + 7919 * b, dev, 10, -10.0, -28.0, 0.3, wide=wide))
            I, Q, expected = batches[0]
            workload = "configs[2]: %d segments per GPU x 10 overlapping signals, SNR -10..-28 dB, deep search on" % nseg
            raw = None
            if nseg >= 1024 and "WSPR_FANO_FAST" not in os.environ:
                # crowded band, thousands of Fano time-outs per step: short host budget + device tail (K6w);
                # results are those of the full budget by construction
                # the fewer CPUs a rank has, the earlier an attempt is handed to the device tail (measured with
                # 2 host threads: 5.2k / 6.2k / 6.7k segments/s at 200 / 60 / 25 cycles per bit)
                fast = args.fano_fast if args.fano_fast else (200 if cpus_here >= 8 else (60 if cpus_here >= 4 else 25))
                fast_old = L.wspr_set_fano_fast_budget(C.c_uint(fast))
                workload += ("; Fano attempts on the device (exact wave-parallel search) once the pipeline has seen the "
                             "band is crowded; before that the host pool with a budget of %d cycles/bit + device tail" % fast)
        else:
            # configs[4]: `nraw` distinct raw segments stay resident (576 MB each); a step takes nseg / nraw front-end
            # waves over them into the rows of an IQ ring (360 KB per segment) and ONE decoder call over the nseg rows,
            # so the decoder works on configs[1]-sized batches while the raw data streams through wave after wave
            nraw = min(args.raw_segments, nseg)
            assert nseg % nraw == 0, "--segments must be a multiple of --raw-segments"
            raw, exp_raw = synth_raw_gpu(nraw, 777 + seed, dev, args.snr)
            expected = [exp_raw[i % nraw] for i in range(nseg)]
            stride = int(L.wspr_iq_stride())
            I = torch.zeros(nseg, stride, device=dev, dtype=torch.float32)
            Q = torch.zeros(nseg, stride, device=dev, dtype=torch.float32)
            workload = ("configs[4]: raw 2.4 Msps u8 IQ (576 MB per 2-minute segment, %d distinct segments resident in HBM) "
                        "through the on-GPU decimator (K0) in waves of %d into an IQ ring, decoded %d segments per decoder "
                        "call; 1 signal each, SNR %g dB, 10 LSB rms noise; the config's 4096 segments = %d such steps"
                        % (nraw, nraw, nseg, args.snr, max(1, 4096 // nseg)))
        if config != 5:
            workload += ("; %d distinct batches resident (seeds apart, every signal its own message out of the whole type-1 "
                         "space), a lane's consecutive steps decode different ones" % nrot) if nrot > 1 else \
                        "; ONE batch of 7 600 possible messages decoded on every step (warm message cache, the round-5 shape)"
        torch.cuda.synchronize()
        slots = args.slots if args.slots else (1 if inflight >= 4 else 0)         # few in flight: the library's own split
        slots_used = [ex.submit(L.wspr_set_thread_slots, slots).result() for ex in lanes][0]
        decs = [w.BatchDecoder(nseg, max_results=16 if config != 3 else 32, options=opt) for _ in range(inflight)]
        gatherers = [wd.SpotGatherer(d.out, d.nres, nseg, d.max_results, rec, dst=0) for d in decs] if use_dist else None
        if config == 5:                                  # the decimator's output rows, one set per lane
            IQs = [(I, Q)] + [(torch.zeros_like(I), torch.zeros_like(Q)) for _ in range(inflight - 1)]
            torch.cuda.synchronize()                     # raw pointers from here on (stream contract of the library)

        def decode_on(k, b=0):
            if config == 5:
                Ik, Qk = IQs[k]
                row = Ik.stride(0) * 4
                for wv in range(nseg // nraw):
                    rc = L.wspr_decimate_u8_batch_device(raw.data_ptr(), RAW_BYTES, nraw, Ik.data_ptr() + wv * nraw * row,
                                                         Qk.data_ptr() + wv * nraw * row, 1)
                    assert rc == 0
                decs[k].decode_ptr(Ik.data_ptr(), Qk.data_ptr(), NS, Ik.stride(0))
            else:
                decs[k].decode(batches[b][0], batches[b][1])
            return (k, b), w.last_timings()              # timings of THIS step, read on the lane that ran it

        gather_s = [0.0, 0, 0.0]                             # seconds in the fan-in (wall), gathers, CPU seconds of the driving thread

        def run_steps(n):
            """n steps, at most `inflight` of them running; spot records are gathered in step order."""
            pending, last, tim, lastk, lastb = [], None, None, 0, 0
            for s in range(n):
                done = None
                if len(pending) >= inflight:
                    (done, _), tim = pending.pop(0).result()
                    if use_dist:
                        t_g, c_g = time.perf_counter(), time.thread_time()
                        gatherers[done].stage()               # results copied out: the lane is free again
                        gather_s[0] += time.perf_counter() - t_g
                        gather_s[2] += time.thread_time() - c_g
                # lane k = s mod inflight runs its j-th step (j = s div inflight) on batch (k + j) mod nrot
                pending.append(lanes[s % inflight].submit(decode_on, s % inflight, (s % inflight + s // inflight) % nrot))
                if done is not None and use_dist:
                    t_g, c_g = time.perf_counter(), time.thread_time()
                    last = gatherers[done].exchange()         # every rank's records land on rank 0 (RCCL)
                    gather_s[0] += time.perf_counter() - t_g
                    gather_s[2] += time.thread_time() - c_g
                    gather_s[1] += 1
            for fut in pending:
                (lastk, lastb), tim = fut.result()
                if use_dist:
                    t_g, c_g = time.perf_counter(), time.thread_time()
                    last = gatherers[lastk].gather()
                    gather_s[0] += time.perf_counter() - t_g
                    gather_s[2] += time.thread_time() - c_g
                    gather_s[1] += 1
            return last, tim, (lastk, lastb)

        # a lane needs about four untimed steps before its contexts, buffers, host pools and the clocks are
        # settled (tools/pipelined_trace.py: steps 0-1 create the contexts, 2-6 still run 11-22 ms)
        untimed = max(warmup, {2: 4, 3: 2, 4: 2, 5: 1}[config] * inflight)
        run_steps(untimed)

        def timed(n):
            fence()
            gather_s[0], gather_s[1], gather_s[2] = 0.0, 0, 0.0
            t0 = time.perf_counter()
            out = run_steps(n)
            fence()
            el = time.perf_counter() - t0
            if use_dist:
                tmax = torch.tensor([el], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                el = float(tmax.item())
            return el, out

        elapsed, (gathered, timings, lastk) = timed(steps)
        first = {"steps": steps, "seconds": elapsed}
        n_timed = steps
        if elapsed < args.min_seconds:
            # too short to trust: repeat with enough steps for the minimum region (every rank takes the same
            # decision: `elapsed` is the maximum over ranks)
            n_timed = int(np.ceil(steps * args.min_seconds / max(elapsed, 1e-6) * 1.1))
            elapsed, (gathered, timings, lastk) = timed(n_timed)
        # correctness of what was timed: every segment's message must be the transmitted one
        lastk, lastb = lastk
        dec = decs[lastk]                                                        # the last step's results ...
        I, Q, expected = batches[lastb] if batches else (I, Q, expected)         # ... and the batch it decoded
        del batches
        got = [[s.message.decode() for s in dec.spots(i)] for i in range(nseg)]
        n_sent = sum(len(e) for e in expected)
        n_ok = sum(len(set(expected[i]) & set(got[i])) for i in range(nseg))
        n_false = sum(len([m for m in got[i] if m not in expected[i]]) for i in range(nseg))
        total_spots = int(gathered[0].sum()) if gathered is not None else dec.total_spots()
        if fast_old is not None:
            L.wspr_set_fano_fast_budget(C.c_uint(fast_old))
        return {"config": config, "nseg": nseg, "I": I, "Q": Q, "raw": raw, "expected": expected, "got": got,
                "workload": workload, "steps": n_timed, "elapsed": elapsed, "first_try": first, "untimed": untimed, "slots": slots_used, "inflight": inflight,
                "value": world * nseg * n_timed / elapsed, "ms_per_step": elapsed / n_timed * 1e3,
                "decoded_ok": "%d/%d" % (n_ok, n_sent), "false_decodes": n_false, "spots_total": total_spots,
                "spots_own": dec.total_spots(),
                "timings": dict(timings, message_cache_hit_rate=(timings.get("message_cache_hits", 0.0) /
                                                                  max(1.0, timings.get("message_cache_lookups", 0.0)))),
                "rotate": nrot,
                # the fan-in of the timed steps as the driving thread saw it (staging copy + H2D + gather + D2H on rank 0)
                "gather_ms_per_step": (1e3 * gather_s[0] / gather_s[1]) if gather_s[1] else None,
                "gather_cpu_ms_per_step": (1e3 * gather_s[2] / gather_s[1]) if gather_s[1] else None}

    nseg = args.segments or {2: 1024, 3: 8192, 4: 8192, 5: 1024}[args.config]
    t_prog = time.perf_counter()
    walls = {}

    def lap(name):
        """wall seconds since the previous lap, by block: where the default command's minutes go"""
        nonlocal t_prog
        now = time.perf_counter()
        walls[name] = round(now - t_prog, 1)
        t_prog = now
    m = measure(args.config, nseg, args.steps, args.warmup, rank)
    lap("headline (synthesis, untimed and timed steps)")
    warm_extra = None
    if not args.no_warm_extra and args.config != 5 and args.rotate != 1 and world == 1 and not use_dist:
        # the round-5 shape beside the cold figure (labelled extra, never `value`): ONE batch of synth.message_for texts
        # decoded on every step, so every message after the first step comes out of the host's per-thread cache
        keep = {k: m[k] for k in ("I", "Q")}
        mw = measure(args.config, nseg, max(4, args.steps // 2), args.warmup, rank, warm=True)
        warm_extra = {"value": mw["value"], "unit": "segments/s", "ms_per_step": mw["ms_per_step"], "steps": mw["steps"],
                      "seconds_timed": mw["elapsed"], "decoded_ok": mw["decoded_ok"], "false_decodes": mw["false_decodes"],
                      "message_cache_hit_rate_last_step": mw["timings"]["message_cache_hit_rate"],
                      "cpu_ms_books_last_step": mw["timings"].get("cpu_ms_books"),
                      "note": "ONE batch (7 600 possible messages) decoded on every step: the host's per-thread message cache is "
                              "warm after the first step -- what rounds 2-5 reported; `value` above is the cold figure"}
        del mw
        m.update(keep)
        torch.cuda.empty_cache()
        lap("warm_cache extra")

    # what every rank of the job did, as rank 0 sees it: the block of the job's segments it owned (contiguous shards,
    # segment index -> rank by shard_range), its CPU share and what its library added to the host's load
    mine = {"rank": rank, "segments": list(wd.shard_range(world * nseg, rank, world)), "host_threads": int(os.environ["WSPR_HOST_THREADS"]),
            "host_pool_workers": int(L.wspr_host_pool_workers()), "spots_last_step": int(m["spots_own"]),
            "decoded_ok": m["decoded_ok"], "false_decodes": m["false_decodes"], "device": list(my_dev)}
    ranks_info = [mine]
    if use_dist:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, mine)
    fanout = None
    if use_dist and args.config != 5:
        # ---- real-input fan-out, untimed (the reference's callers hold the IQ in ONE place,
        # rtlsdr_wsprd.c:316, :689): rank 0 scatters rows of ITS batch over RCCL (grouped send/recv of 360 000 B
        # per segment), every rank decodes the rows it received, the records come back in global order and
        # must equal what rank 0 decoded for the same segments in the timed steps.
        per = 4
        ntot = min(nseg, per * world)

        def decode_rows(mi, mq, o):
            d = w.BatchDecoder(mi.shape[0], max_results=16, options=o)
            if mi.shape[0]:
                d.decode(mi.to(dev).contiguous(), mq.to(dev).contiguous())
            return d.out, d.nres
        res = wd.decode_from_root(m["I"][:ntot, :NS] if rank == 0 else None, m["Q"][:ntot, :NS] if rank == 0 else None,
                                  ntot, NS, opt, decode_rows, max_results=16, record_size=rec, root=0)
        if rank == 0:
            cnt, recs = res
            back = [sorted(bytes(recs[s, k * rec + 28:k * rec + 51]).split(b"\0")[0].decode() for k in range(cnt[s]))
                    for s in range(ntot)]
            same = sum(1 for s in range(ntot) if back[s] == sorted(m["got"][s]))
            fanout = {"segments_scattered_from_rank0": ntot, "bytes_per_segment": 2 * 4 * NS,
                      "equal_to_rank0_own_decode": "%d/%d" % (same, ntot),
                      "over": "rccl send/recv (grouped)" if backend == "nccl" else backend + " send/recv"}

    if rank == 0:
        I, Q = m["I"], m["Q"]
        roof = None
        if not args.no_kernel_roofline:
            # ---- kernel-level roofline of the FFT+sync stage, HIP events on the launch stream.  The timing sets and the
            # calibration kernels are entry points of the LAB library (include/wspr_mi355x_bench.h; the same kernels from
            # the same sources: the product exports no benchmark); `value` above was measured on the product library.
            LL = w.lab()
            ms = (C.c_double * 8)()
            LL.wspr_bench_fft_sync(I.data_ptr(), Q.data_ptr(), nseg, NS, I.stride(0), 20 if nseg <= 2048 else 10, C.addressof(ms))
            k1, k2, k3 = ms[0], ms[1], ms[2]
            traffic, traffic_src = None, None
            if world == 1 and not use_dist and not args.no_pmc and args.config in (2, 3, 4):
                # measured in this run (the verdict of round 2: a number read from profiles/ goes stale when a kernel changes)
                pm = measure_k1_traffic(nseg, 10 if args.config == 3 else 1)
                if pm and pm.get("hbm_bytes_per_segment"):
                    traffic = pm["hbm_bytes_per_segment"] * nseg
                    kk = pm["kernels"][pm["dominant_kernel"]]
                    traffic_src = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/pmc_k1.py "
                                   "%d %d; read %.3f GB + written %.3f GB per launch; FETCH_SIZE x %.3f, WRITE_SIZE x %.3f by the 1 GiB "
                                   "copy" % (nseg, 10 if args.config == 3 else 1, kk["hbm_read_bytes"] / 1e9, kk["hbm_written_bytes"] / 1e9,
                                             pm["calibration"]["true_bytes_per_counted_read_byte"],
                                             pm["calibration"]["true_bytes_per_counted_written_byte"]))
            for name in (() if traffic else ("r06_k1_pmc_traffic.json", "r05_k1_pmc_traffic.json", "r04_k1_pmc_traffic.json", "r03_k1_pmc_traffic.json", "r02_k1_pmc_traffic.json", "r01_k1_pmc_traffic.json")):
                tf = os.path.join(ROOT, "profiles", name)
                if os.path.exists(tf):
                    jd = json.load(open(tf))
                    per_seg = jd.get("hbm_bytes_per_segment") or (jd.get("hbm_bytes_per_launch", 0) / jd.get("segments", 1024))
                    if per_seg:
                        traffic, traffic_src = per_seg * nseg, "profiles/" + name
                        break
            roof = {"bound": "hbm", "kernel": "fft_bank_avg_kernel<4> (K1 fused with the time average)" if nseg >= 256
                    else "fft_bank_kernel<4> (K1)", "achieved": K1_BYTES * nseg / (k1 * 1e-3) / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": K1_BYTES * nseg / (k1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": k1,
                    "bytes_per_launch": K1_BYTES * nseg,
                    # what the kernel really moves (PMC): the fused form does not send the 106 bin rows that only the
                    # time average needed to HBM, so its traffic is BELOW the algorithmic figure of SURVEY 8(d)
                    "traffic_GBs": (traffic / (k1 * 1e-3) / 1e9) if traffic else None,
                    "traffic_frac_of_peak": (traffic / (k1 * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                    "fft_sync_stage": {"kernels_ms": {"fft_bank": k1, "pick_peaks": k2, "coarse_sync": k3},
                                       "wall_ms": ms[4],
                                       "bytes_per_launch": STAGE_BYTES * nseg,
                                       "achieved_GBs": STAGE_BYTES * nseg / ((k1 + k2 + k3) * 1e-3) / 1e9,
                                       "frac": STAGE_BYTES * nseg / ((k1 + k2 + k3) * 1e-3) / 1e9 / HBM_PEAK_GBS}}
            # ---- the fp32-VALU-bound kernels: tiled lag scan (K4 mode 0), frequency scan + first rung, subtraction (K7)
            vms = (C.c_double * 8)()
            LL.wspr_bench_valu.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p]
            if LL.wspr_bench_valu(I.data_ptr(), Q.data_ptr(), min(nseg, 2048), NS, I.stride(0), 5, C.addressof(vms)) > 0 and vms[2] > 0:   # five timed passes after an untimed one
                def valu(flop_each, n, t_ms):
                    tf = flop_each * n / (t_ms * 1e-3) / 1e12
                    return {"avg_launch_ms": t_ms, "units": int(n), "achieved_TFs": tf, "frac_of_fp32_vector_peak": tf / VALU_PEAK_TF,
                            "frac_of_no_fma_bound": tf / VALU_NOFMA_TF}
                mtf = C.c_double(0.0)
                LL.wspr_calib_valu.argtypes = [C.c_int, C.c_void_p]
                LL.wspr_calib_valu(20, C.addressof(mtf))
                roof["valu"] = {"peak_TFs": VALU_PEAK_TF, "no_fma_bound_TFs": VALU_NOFMA_TF,
                                # register-only v_pk_mul_f32 + v_pk_add_f32 chains on every SIMD: the practical ceiling
                                "measured_no_fma_TFs": mtf.value,
                                "note": "separately rounded mul/add (no FMA, by parity): the packed-fp32 pipes issue at most "
                                        "half the FMA peak",
                                "K4_lag_scan (demod_lagsys_kernel + demod_metric_kernel)": valu(K4_FLOP, vms[2], vms[0]),
                                "K4_freq_scan_first_rung (freq_scalar_kernel + ...)": valu(K41_FLOP, vms[2], vms[4]),
                                "K7_subtract (sub_runs_wave_kernel + sub_fir_fused_kernel)": valu(K7_FLOP, vms[3], vms[1])}
            # measured ceiling: the library's plain stream-copy kernel over 1 GiB (read + write, far beyond
            # the 256 MiB Infinity Cache), same stream and launch path as the kernels above
            if args.config != 5 and not args.no_ceilings:
                n_copy = 1 << 28
                src = torch.empty(n_copy, device=dev, dtype=torch.float32).normal_()
                dst = torch.empty_like(src)
                torch.cuda.synchronize()
                LL.wspr_calib_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(n_copy), 2)
                t0 = time.perf_counter()
                LL.wspr_calib_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(n_copy), 10)
                roof["measured_copy_GBs"] = 10 * 8.0 * n_copy / (time.perf_counter() - t0) / 1e9
                # the tuned copy (16 bytes per lane, four loads in flight per lane, resident grid), by cache policy: HIP events
                LL.wspr_calib_copy16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
                cms = C.c_double(0.0)
                tuned = {}
                for variant, name in ((0, "nontemporal_loads_and_stores"), (1, "nontemporal_stores"), (2, "default_policy")):
                    LL.wspr_calib_copy16(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(n_copy), 3, variant, None)
                    LL.wspr_calib_copy16(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(n_copy), 10, variant,
                                        C.addressof(cms))
                    tuned[name] = 8.0 * n_copy / (cms.value * 1e-3) / 1e9
                roof["measured_copy16_GBs"] = max(tuned.values())            # the ceiling of mixed read + write traffic here
                roof["measured_copy16_by_policy_GBs"] = tuned
                LL.wspr_calib_copy16(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(n_copy), 3, 3, None)
                LL.wspr_calib_copy16(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(n_copy), 10, 3, C.addressof(cms))
                roof["measured_write_only_GBs"] = 4.0 * n_copy / (cms.value * 1e-3) / 1e9
                # ... and write-only in the spectrogram's pattern (64-byte pieces of 311 rows per group of 16 time blocks)
                LL.wspr_calib_copy16(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(n_copy), 3, 4, None)
                LL.wspr_calib_copy16(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(n_copy), 10, 4, C.addressof(cms))
                roof["measured_write_only_spectrogram_pattern_GBs"] = (n_copy // (417 * 352)) * 311 * 22 * 64 / (cms.value * 1e-3) / 1e9
                rd = (C.c_double * 1)()                                       # read-only: K0's access pattern over the same bytes
                LL.wspr_calib_read(C.c_void_p(src.data_ptr()), C.c_size_t(4 * n_copy), 1, 10, C.addressof(rd))
                roof["measured_read_only_GBs"] = 4.0 * n_copy / (rd[0] * 1e-3) / 1e9
                del src, dst
        if roof is not None and m["timings"].get("candidates_refined") is not None:
            # the step as a whole against the bound that really limits it: the prescribed arithmetic of the fine search and
            # the subtraction (SURVEY 8d per-unit figures x the counts of the last timed step) over the step's wall time,
            # against the no-FMA fp32 vector bound (separately rounded mul and add: half the FMA peak)
            tm = m["timings"]
            flop = tm["candidates_refined"] * (K4_FLOP + K41_FLOP) + tm.get("subtractions", 0.0) * K7_FLOP
            tf = flop / (m["ms_per_step"] * 1e-3) / 1e12
            roof["step"] = {"bound": "fp32 VALU, no FMA", "flop_per_step": flop, "achieved_TFs": tf, "peak_TFs": VALU_NOFMA_TF,
                            "frac": tf / VALU_NOFMA_TF, "frac_of_fp32_vector_peak": tf / VALU_PEAK_TF,
                            "counts": {"candidates_refined": tm["candidates_refined"], "subtractions": tm.get("subtractions")},
                            "per_unit_flop": {"lag_scan": K4_FLOP, "freq_scan": K41_FLOP, "subtract": K7_FLOP},
                            "note": "the roofline block above answers the metric's letter (the FFT+sync stage's dominant kernel, "
                                    "HBM-bound, < 2 % of a configs[2] step); THIS is what bounds the step: vector issue"}
        lap("kernel-level roofline, PMC passes, ceilings")
        cpu = None
        if args.config == 5 and roof is not None:
            roof["front_end_K0"] = k0_report(w.lab(), m, world == 1 and not args.no_cpu_baseline)
            cpu = roof["front_end_K0"].pop("cpu_baseline", None)
        if world == 1 and not args.no_cpu_baseline and args.config != 5:
            cnt = min(nseg, 512)
            Ih, Qh = I[:cnt].cpu().numpy(), Q[:cnt].cpu().numpy()
            cpu, cpu_msgs = cpu_baseline(Ih, Qh, m["expected"][:cnt], 25.0 if args.config == 2 else 14.0)
            same = sum(1 for i in range(len(cpu_msgs)) if cpu_msgs[i] == m["got"][i])
            cpu["gpu_equals_cpu_spots"] = "%d/%d segments" % (same, len(cpu_msgs))
        lap("cpu_baseline")
        secondary = None
        if world == 1 and args.config == 3 and not args.no_secondary and not use_dist:
            del I, Q
            m["I"] = m["Q"] = None
            torch.cuda.empty_cache()
            released = L.wspr_release_buffers()          # the twelve lanes' work buffers of the 8192-segment batches
            m2 = measure(2, 1024, 200, 8, rank)
            secondary = {"workload": m2["workload"], "value": m2["value"], "unit": "segments/s", "steps": m2["steps"],
                         "ms_per_step": m2["ms_per_step"], "batches_in_flight": m2["inflight"], "slots_per_batch": m2["slots"],
                         "seconds_timed": m2["elapsed"], "decoded_ok": m2["decoded_ok"],
                         "false_decodes": m2["false_decodes"], "stage_ms_last_step": m2["timings"]}
        lap("secondary")
        tertiary = None
        if world == 1 and args.config == 3 and not args.no_tertiary and not use_dist:
            # configs[4] (raw 2.4 Msps input through the on-GPU decimator) rides along in the default line: its
            # throughput, K0's roofline and the decimator's CPU baseline
            m["I"] = m["Q"] = None
            torch.cuda.empty_cache()
            L.wspr_release_buffers()
            m3 = measure(5, 1024, 6, 2, rank)
            tertiary = {"workload": m3["workload"], "value": m3["value"], "unit": "segments/s", "steps": m3["steps"],
                        "ms_per_step": m3["ms_per_step"], "batches_in_flight": m3["inflight"], "slots_per_batch": m3["slots"],
                        "seconds_timed": m3["elapsed"], "decoded_ok": m3["decoded_ok"],
                        "false_decodes": m3["false_decodes"], "stage_ms_last_step": m3["timings"],
                        "front_end_K0": k0_report(w.lab(), m3, not args.no_cpu_baseline)}
            del m3
            torch.cuda.empty_cache()
        lap("tertiary")
        shard = None
        if world == 1 and args.config == 3 and not args.no_shard_block and not use_dist:
            # configs[3]'s per-rank shard (8 192 single-signal segments): in this process with the whole host, then in
            # children with a rank's CPU share and as a world-1 RCCL rank
            I = Q = None
            m["I"] = m["Q"] = None
            torch.cuda.empty_cache()
            L.wspr_release_buffers()
            m4 = measure(4, 8192, 30, 4, rank)
            full = {"value": m4["value"], "unit": "segments/s", "ms_per_step": m4["ms_per_step"], "steps": m4["steps"],
                    "batches_in_flight": m4["inflight"], "slots_per_batch": m4["slots"], "workload": m4["workload"],
                    "decoded_ok": m4["decoded_ok"], "false_decodes": m4["false_decodes"], "stage_ms_last_step": m4["timings"]}
            del m4
            torch.cuda.empty_cache()
            L.wspr_release_buffers()
            shard = shard_block(args, full)
        lap("configs3_shard")
        host_entry = None
        if world == 1 and args.config == 3 and not args.no_host_entry and not use_dist:
            I = Q = None
            m["I"] = m["Q"] = None
            torch.cuda.empty_cache()
            L.wspr_release_buffers()
            host_entry = host_entry_block(dev, lanes[:12], m["value"])
        lap("host_entry")
        hashtable = None
        if world == 1 and args.config == 3 and not args.no_hashtable_block and not use_dist:
            torch.cuda.empty_cache()
            hashtable = hashtable_block(dev, lanes)
        lap("usehashtable_batch")
        out = {
            "metric": "2-minute WSPR segments decoded per second", "value": m["value"],
            "unit": "segments/s", "n_gpus": world, "distinct_devices": distinct_devices,
            "devices_shared": distinct_devices < world, "steps": m["steps"], "warmup": args.warmup,
            "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": m["workload"] + ", 45000 complex f32 samples @ 375 sps, resident in HBM; reference "
                                   "defaults (npasses 2, subtraction on, quickmode off)",
                       "segments_per_gpu": nseg, "parallelism": "segments sharded per GPU, spots gathered on rank 0",
                       "gathered_over": ("rccl" if backend == "nccl" else backend) if use_dist else "none (one process)",
                       "launched_by": "bench.py --gpus N (self-spawned ranks)" if os.environ.get("WSPR_BENCH_SPAWNED")
                       else ("torch.distributed.run" if "WORLD_SIZE" in os.environ else "single process"),
                       "batches_in_flight": inflight, "slots_per_batch": m["slots"], "untimed_steps": m["untimed"],
                       "distinct_batches_rotated": m["rotate"]},
            "steps_requested": args.steps, "seconds_timed": m["elapsed"], "first_try": m["first_try"],
            "decoded_ok": m["decoded_ok"], "false_decodes": m["false_decodes"], "spots_total": m["spots_total"],
            "stage_ms_last_step": dict(m["timings"], note="times: maximum over the slots of the lane that ran the last "
                                                         "step; counts: sum over its slots"),
            "host_threads": int(os.environ["WSPR_HOST_THREADS"]),
            "cpus_pinned_to": len(os.sched_getaffinity(0)) if args.cpu_share else None,
            "gather_ms_per_step": m.get("gather_ms_per_step"), "gather_cpu_ms_per_step": m.get("gather_cpu_ms_per_step"),
            "host": {"hw_threads": os.cpu_count(), "usable_cpus": usable_cpus()},
            "host_pool_workers": int(L.wspr_host_pool_workers()),
            "roofline": roof, "cpu_baseline": cpu, "secondary": secondary, "tertiary": tertiary,
            "configs3_shard": shard, "host_entry": host_entry, "usehashtable_batch": hashtable,
            "fanout_check": fanout, "ranks": ranks_info if use_dist else None,
        }
        if world == 1 and not use_dist and args.config == 3:
            if not args.no_reference_case:
                out["reference_case_configs0"] = reference_case_block(not args.no_cpu_baseline)
            if not args.no_share_block:
                torch.cuda.empty_cache()
                L.wspr_release_buffers()
                out["per_rank_share_of_8"] = share_of_8_block(args, inflight)
        lap("reference case, per_rank_share_of_8")
        out["wall_seconds_by_block"] = walls
        out["warm_cache"] = warm_extra
        # the figures a reader of the END of this (long) line needs, last: the driver records the tail of stdout
        out["summary"] = {
            "value_segments_per_s": m["value"], "ms_per_step": m["ms_per_step"], "workload": "configs[%d]" % (args.config - 1),
            "message_cache_hit_rate_last_step": m["timings"].get("message_cache_hit_rate"),
            "warm_cache_value": warm_extra["value"] if warm_extra else None,
            "roofline_frac_K1_hbm": roof["frac"] if roof else None,
            "fft_sync_stage_frac_hbm": roof["fft_sync_stage"]["frac"] if roof else None,
            "roofline_step_frac_no_fma": roof["step"]["frac"] if roof and "step" in roof else None,
            "cpu_baseline_segments_per_s": cpu["value"] if cpu else None,
            "secondary_configs1_value": secondary["value"] if secondary else None,
            "tertiary_configs4_value": tertiary["value"] if tertiary else None,
            "tertiary_K0_frac_hbm": (tertiary["front_end_K0"].get("frac") if tertiary and isinstance(tertiary.get("front_end_K0"), dict) else None),
            "configs3_shard_full_host_value": (shard.get("full_host", {}).get("value") if isinstance(shard, dict) else None),
            "configs3_shard_share_of_8_value": (shard.get("share_of_8", {}).get("value") if isinstance(shard, dict) else None),
            "configs3_shard_share_of_8_over_full_host": (shard.get("share_of_8_over_full_host") if isinstance(shard, dict) else None),
            "per_rank_share_of_8": (out.get("per_rank_share_of_8", {}) or {}).get("value"),
        }
    line = json.dumps(out) if rank == 0 else None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # The JSON line is the LAST thing on stdout: RCCL writes its version banner and warnings through C stdio
    # (NCCL_DEBUG=VERSION is exported on the GPU boxes), which, on a pipe, is only flushed at exit -- i.e. after a
    # line printed earlier from Python.  Every rank flushes C stdio now; rank 0 prints once the others have had
    # the time to leave.
    try:
        C.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    if rank == 0:
        if world > 1:
            time.sleep(0.5)
        print(line, flush=True)


if __name__ == "__main__":
    main()

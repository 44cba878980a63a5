// Host scheduler of the MI355X WSPR decoder: owns the HIP stream, the resident
// buffers and the host Fano pool, and drives the kernels of csrc/kernels/ so that
// a batch of independent 2-minute segments is decoded with the reference's
// semantics (wsprd/wsprd.c:416-855), including its sequential ones:
//   * pass 0 visits a segment's candidates strongest first and every successful
//     decode is subtracted from that segment's IQ before the next candidate is
//     refined -> candidates are processed in lock-step "ranks" across segments;
//   * pass 1 (and any pass without subtraction) has no such dependence -> all
//     (segment, candidate) pairs go through the GPU in one wave, and only the
//     host bookkeeping (hash table, de-duplication, early loop exits) is ordered;
//   * the jitter ladder stops at the first Fano success -> jitter 0 is demodulated
//     for everybody, the remaining 42 steps only for the candidates that need them.
// The Fano decoder, message unpacking and re-encoding stay on the host (north star).
#include "wspr_context_impl.h"

namespace wspr {

// ---------------------------------------------------------------- stages -----
// K1 + K2a over `nactive` segments; ev (optional): an event before and after each kernel.
// Batches that fill the GPU with one workgroup per segment take the fused kernel (the spectrogram is not
// read back for the time average); WSPR_K1_FUSED=0/1 forces either form.
static void fft_and_average(const float* dI, const float* dQ, const int* d_seglist, int nactive, int samples, float* ps,
                            float* psavg, const DeviceTables& tab, hipStream_t st, std::vector<hipEvent_t>* ev = nullptr) {
    const int blocks = 4 * (samples / kFftSize) - 1;
    auto mark = [&] {
        if (!ev) return;
        hipEvent_t e;
        HIP_OK(hipEventCreate(&e));
        HIP_OK(hipEventRecord(e, st));
        ev->push_back(e);
    };
    static const int fused_cfg = [] { const char* e = lab_env("WSPR_K1_FUSED"); return e ? atoi(e) : -1; }();
    const bool fused = fused_cfg < 0 ? nactive >= 256 : fused_cfg != 0;
    mark();
    if (fused) launch_fft_bank_avg(dI, dQ, d_seglist, nactive, samples, ps, psavg, tab, st);
    else launch_fft_bank(dI, dQ, d_seglist, nactive, samples, ps, tab, st);
    mark(); mark();
    if (!fused) launch_time_average(ps, d_seglist, nactive, blocks, psavg, st);
    mark();
}

void Context::run_fft_sync(int nseg, int samples, int maxdrift, bool coarse, const int* d_seglist, int nactive,
                           float* noise_out, float* smspec_out) {
    const int blocks = 4 * (samples / kFftSize) - 1;
    float* ps = ps_buffer(nseg);
    DevCand* cand = static_cast<DevCand*>(d->cand.need((size_t)nseg * kMaxCand * sizeof(DevCand)));
    int* npk = static_cast<int*>(d->npk.need((size_t)nseg * 4));
    float* psavg = static_cast<float*>(d->psavg.need((size_t)nseg * kPsStride * 4));
    fft_and_average(d->iqI.as<float>(), d->iqQ.as<float>(), d_seglist, nactive, samples, ps, psavg, d->tab, d->stream);
    launch_pick_peaks(ps, d_seglist, nactive, blocks, psavg, cand, npk, noise_out, smspec_out, d->tab, d->stream, true);
    if (coarse) launch_coarse_sync(ps, d_seglist, nactive, blocks, cand, npk, maxdrift, d->tab, d->stream);
}

// Candidate lists to the host.  A segment rarely holds more than a dozen candidates of its 200 slots, so only
// the head of every list is copied (2-D copy into pinned memory, `cand_head` entries per segment, adapted to
// the lists seen so far); a batch with a longer list is fetched again in full.
void Context::fetch_candidates_async(int nseg) {
    Impl& c = *d;
    const int head = c.cand_head;
    int* h_npk = static_cast<int*>(c.h_npk.need((size_t)nseg * 4));
    DevCand* h_cand = static_cast<DevCand*>(c.h_cand.need((size_t)nseg * kMaxCand * sizeof(DevCand)));
    HIP_OK(hipMemcpyAsync(h_npk, c.npk.p, (size_t)nseg * 4, hipMemcpyDeviceToHost, c.stream));
    HIP_OK(hipMemcpy2DAsync(h_cand, (size_t)head * sizeof(DevCand), c.cand.p, (size_t)kMaxCand * sizeof(DevCand),
                            (size_t)head * sizeof(DevCand), nseg, hipMemcpyDeviceToHost, c.stream));
}

void Context::finish_fetch_candidates(int nseg, std::vector<int>& npk, std::vector<DevCand>& cand) {
    Impl& c = *d;
    if (npk.size() < (size_t)nseg) npk.resize(nseg);
    if (cand.size() < (size_t)nseg * kMaxCand) cand.resize((size_t)nseg * kMaxCand);    // entries beyond npk[s] are never read
    const int* h_npk = c.h_npk.as<int>();
    const DevCand* h_cand = c.h_cand.as<DevCand>();
    int longest = 0;
    for (int s = 0; s < nseg; ++s) { npk[s] = h_npk[s]; longest = std::max(longest, std::min(h_npk[s], kMaxCand)); }
    const int head = c.cand_head;
    if (longest > head) {                                   // rare: fetch the full lists
        HIP_OK(hipMemcpyAsync(c.h_cand.p, c.cand.p, (size_t)nseg * kMaxCand * sizeof(DevCand), hipMemcpyDeviceToHost, c.stream));
        sync();
        memcpy(cand.data(), h_cand, (size_t)nseg * kMaxCand * sizeof(DevCand));
    } else {
        for (int s = 0; s < nseg; ++s)
            memcpy(cand.data() + (size_t)s * kMaxCand, h_cand + (size_t)s * head, (size_t)std::min(npk[s], head) * sizeof(DevCand));
    }
    c.cand_head = std::min(kMaxCand, std::max(16, (longest + 15) / 8 * 8));
    // The reference sorts the list (built in ascending bin order) by snr = 10 log10f(peak) - 26.3 with glibc's
    // stable merge sort (wsprd.c:616, 631).  The device ordered it with ocml's log10f, which differs from
    // glibc's in the last bit for some arguments and could swap two nearly equal peaks -- and with them the
    // order in which signals are subtracted.  Re-rank here with the host libm: the order (and the reported
    // snr) then never depends on ocml.
    c.pool->run(nseg, [&](int s) {
        const int n = std::min(npk[s], kMaxCand);
        if (n <= 0) return;
        DevCand* c0 = cand.data() + (size_t)s * kMaxCand;
        for (int j = 0; j < n; ++j) c0[j].snr = 10.0 * log10f(c0[j].peak) - (float)26.3;
        bool sorted = true;
        for (int j = 1; j < n && sorted; ++j)
            sorted = c0[j - 1].snr > c0[j].snr || (c0[j - 1].snr == c0[j].snr && c0[j - 1].bin < c0[j].bin);
        if (sorted) return;
        std::sort(c0, c0 + n, [](const DevCand& a, const DevCand& b) { return a.bin < b.bin; });
        std::stable_sort(c0, c0 + n, [](const DevCand& a, const DevCand& b) { return a.snr > b.snr; });
    });
}

void Context::fetch_candidates(int nseg, std::vector<int>& npk, std::vector<DevCand>& cand) {
    fetch_candidates_async(nseg);
    sync();
    finish_fetch_candidates(nseg, npk, cand);
}

// ---------------------------------------------------------------- decoding ---
namespace {

struct WaveItem {
    int seg, cand;
    // filled by the GPU wave
    FineState fine;
    bool worth = false, decoded = false, rung0_pending = false;
    int  jitter = 0;
    unsigned cycles = 0;
    unsigned char decdata[11];
};

// CPU time of the calling thread (not wall time: a thread asleep in an event wait costs nothing) added to *acc
struct CpuSpan {
    double* acc;
    double t0;
    static double now_ms() {
        timespec ts;
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
        return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    }
    explicit CpuSpan(double* a) : acc(a), t0(now_ms()) {}
    ~CpuSpan() { *acc += now_ms() - t0; }
};

struct Timer {
    hipEvent_t a, b;
    hipStream_t st;
    double* acc;
    Timer(hipEvent_t a_, hipEvent_t b_, hipStream_t s, double* acc_) : a(a_), b(b_), st(s), acc(acc_) {
        HIP_OK(hipEventRecord(a, st));
    }
    // waits for everything queued so far and surfaces launch/execution errors loudly
    void stop() {
        HIP_OK(hipGetLastError());
        HIP_OK(hipEventRecord(b, st));
        host_wait(b);
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, a, b));
        *acc += ms;
    }
};


}  // namespace

// Assigns each drift-free item its slot in the phasor-table buffer (drifting ones build their 162
// per-symbol tables inside the kernel) and splits the items into the two launch lists of the tiled
// demodulator.  Returns the table count.
static size_t plan_tables(FineState* items, int n, int* lists, int* n_shared, int* n_own) {
    size_t next = 0;
    int ns = 0, no = 0;
    for (int i = 0; i < n; ++i) {
        items[i].pad = (int)next;
        if (items[i].drift != 0.0f) { lists[n + no++] = i; }
        else                        { next += 1; lists[ns++] = i; }
    }
    *n_shared = ns;
    *n_own = no;
    return next;
}

// Fano work split.  A soft-symbol vector that decodes almost always does so within
// a few hundred cycles; one that does not costs the full 810 000-cycle time-out (wsprd.c:431,
// fano.c:149-153), milliseconds of a CPU core, and crowded bands produce thousands of those.
// The host pool therefore runs every attempt with a SMALL budget and treats "not finished" as a
// provisional failure, so that the batch keeps moving; the unfinished attempts are completed, with
// the reference's full budget, by the device Fano kernel (K6) at the end.  If any of them turns
// out to decode after all -- which would have changed what the reference did next -- the segment
// is decoded again from its original IQ with the host running the full budget, so the final
// spots are exactly the reference's; the re-decode takes the results of the attempts it repeats from
// a memo (FanoMemo).  WSPR_FANO_FAST / wspr_set_fano_fast_budget() = cycles-per-bit of the fast budget;
// the default 10000 (the reference's own budget) means no split.
// Where the Fano attempts of a wave run.  Host (north star, default): the host pool, optionally with the
// budget split above.  Device: every attempt goes to the wave-parallel device search (K6w) straight from the
// soft symbols in HBM -- no host Fano at all, nothing postponed, nothing decoded twice; a wave then costs two
// extra device round trips (~10 ms each when it holds a time-out), which more batches in flight cover.  It is
// what a rank with two CPUs wants (8 GPUs behind a 16-CPU quota) and what a crowded band wants (thousands of
// time-outs per batch: 25-27 k segments/s on configs[2] against 21.6 k with the host pool and the budget split);
// a quiet band decodes faster on the host (configs[1]: 4.4 vs 4.7 ms per step: almost every attempt decodes
// within microseconds and the device round trips are pure latency).  WSPR_FANO_DEVICE=1 forces it, =0 forbids
// it, unset = automatic for batches of >= 256 segments per pipeline: when the rank has fewer than four host
// threads, or when the previous batch of this pipeline ran into more than one time-out per ten segments.
std::atomic<int>& fano_device_setting() {
    static std::atomic<int> v{[] { const char* e = getenv("WSPR_FANO_DEVICE"); return e ? atoi(e) : -1; }()};
    return v;
}
static int fano_device_mode() { return fano_device_setting().load(); }

std::atomic<unsigned>& fano_fast_budget() {
    static std::atomic<unsigned> v{[] { const char* e = getenv("WSPR_FANO_FAST"); return e ? (unsigned)atoi(e) : 10000u; }()};
    return v;
}

int Context::decode_again(int nseg, int samples, const decoder_options& opt, decoder_results* out, int max_results,
                          int* n_results, const std::vector<int>& segs, HashBatch* hb, int hb_off) {
    PendingFano none;
    return decode_core(nseg, samples, opt, out, max_results, n_results, segs, 0u, none, nullptr, nullptr, hb, hb_off);
}

int Context::decode_resident(int nseg, int samples, const decoder_options& opt, decoder_results* out,
                             int max_results, int* n_results, const std::function<void(const std::vector<int>&)>& reload,
                             wspr_trace* trace, HashBatch* hb, int hb_off) {
    Impl& c = *d;
    for (double& v : c.t_ms) v = 0.0;
    c.n_fano = 0; c.n_timeout = 0; c.n_cycles = 0; c.n_kept = 0; c.n_subjobs = 0; c.n_mc_lookups = 0; c.n_mc_hits = 0;
    const auto t_all0 = std::chrono::steady_clock::now();
    CpuSpan cpu_all(&c.t_ms[16]);
    t_spin_us = nseg <= 128 ? 600 : 40;      // a call of up to a hundred segments is a few milliseconds: its waits poll (lone 17-127-segment calls: 1.7-2.8 -> 1.3-2.3 ms)
    for (int s = 0; s < nseg; ++s) n_results[s] = 0;
    const int blocks = 4 * (samples / kFftSize) - 1;
    if (nseg <= 0) return 0;
    if (samples > kMaxSamples || blocks < 23) return 0;      // outside what the reference arrays allow

    const unsigned fast_cfg = fano_fast_budget().load();
    // A handful of segments gain nothing from the device search (their time-outs fit the host pool) and would pay its
    // kernel's latency: 1 segment 3.0 against 3.9 ms, 4-17 segments alike.  From about 32 crowded segments on the host
    // pool is what a lone call waits for (tools/lone_call_latency.py, 10 signals per segment, 16 CPUs: 64 segments 76
    // against 32 ms, 127: 153 against 34, 256: 116 against 38; the limit was 256 until the end of round 5).
    const bool dev_fano = fano_device_mode() > 0 ||
                          (fano_device_mode() < 0 && nseg >= 32 && (rank_cpus() < 4 || d->crowded));
    // a traced decode runs every attempt with the full budget where it is first met (nothing provisional)
    // (a shared hash memory keeps the host's full budget too: a provisional failure would log look-ups of a decode
    // that is thrown away)
    const unsigned fast = (reload && nseg >= 256 && !dev_fano && !trace && !hb) ? std::min(fast_cfg, 10000u) : 0u;
    if (trace) memset(trace, 0, (size_t)nseg * sizeof(wspr_trace));
    d->dev_fano = dev_fano;
    std::vector<int> all(nseg);
    for (int s = 0; s < nseg; ++s) all[s] = s;
    PendingFano pend;
    decode_core(nseg, samples, opt, out, max_results, n_results, all, fast >= 10000u ? 0u : fast, pend, nullptr, trace, hb, hb_off);
    if (!pend.seg.empty()) {
        // ---- finish the provisional failures on the device, full budget ----------------------
        const auto t_t0 = std::chrono::steady_clock::now();
        const int np = (int)pend.seg.size();
        std::vector<int> ret(np);
        std::vector<unsigned> cyc(np), met(np), mnp(np);
        std::vector<unsigned char> dat((size_t)np * 10);
        fano_batch(pend.sym.data(), np, 10000u, ret.data(), cyc.data(), met.data(), mnp.data(), dat.data());
        std::vector<char> dirty(nseg, 0);
        for (int i = 0; i < np; ++i) {
            c.n_fano++; c.n_cycles += cyc[i];
            if (ret[i] == 0) dirty[pend.seg[i]] = 1; else c.n_timeout++;
        }
        c.t_ms[2] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_t0).count();
        std::vector<int> redo;
        for (int s = 0; s < nseg; ++s) if (dirty[s]) redo.push_back(s);
        c.t_ms[12] = np; c.t_ms[13] = (double)redo.size();
        if (!redo.empty()) {
            // ---- exact re-decode of the few segments where a late success changes the story ----
            reload(redo);
            FanoMemo memo;
            for (int i = 0; i < np; ++i)
                if (dirty[pend.seg[i]]) memo.add(pend.sym.data() + (size_t)i * kNSymD, ret[i], cyc[i], dat.data() + (size_t)i * 10);
            PendingFano none;
            decode_core(nseg, samples, opt, out, max_results, n_results, redo, 0u, none, &memo);
        }
    }
    c.t_ms[6] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_all0).count();
    c.t_ms[7] = (double)c.n_fano.load();
    c.t_ms[8] = (double)c.n_timeout.load();
    c.t_ms[9] = (double)c.n_cycles.load();
    c.t_ms[14] = (double)c.n_kept.load();                 // refined candidates whose result was consumed (the rest: cut speculation)
    c.t_ms[15] = (double)c.n_subjobs.load();
    c.t_ms[24] = (double)c.n_mc_lookups.load();           // the books are kept by the pool's threads: atomic counts
    c.t_ms[25] = (double)c.n_mc_hits.load();
    c.crowded = c.n_timeout.load() * 10 > nseg;
    return 0;
}

// State of one decode_core() call: the passes over a set of segments, wave by wave.
struct Context::DecodeRun {
    Context& ctx;
    Context::Impl& c;
    const int nseg, samples;
    const decoder_options& opt;
    decoder_results* const out;
    const int max_results;
    const unsigned fast;                      // host Fano budget (cycles per bit) or 0 = the reference's
    PendingFano& pend;
    const FanoMemo* memo = nullptr;           // results already known (re-decode after a late success)
    wspr_trace* trace = nullptr;              // per-candidate record of the fine search (wspr_decode_batch_trace)
    HashBatch* hb = nullptr;                  // usehashtable on a batch: the shared, ordered hash memory ...
    int hb_off = 0;                           // ... and this context's first segment in it
    struct ItemTrace {                        // one wave item's share of it, filled as the wave proceeds
        int m0_shift = 0; float m0_sync = 0; float sync0 = 0, rms0 = 0; int attempts = 0, fano_calls = 0;
        unsigned char sym0[kNSymD];
    };
    std::vector<ItemTrace> wtrace;

    // one Fano attempt on a soft-symbol vector in transmission order (wsprd.c:759-761)
    int fano_attempt(const unsigned char* tx_sym, unsigned* cycles, unsigned char* data11) const {
        memset(data11, 0, 11);
        if (memo)
            if (const FanoMemo::Entry* e = memo->find(tx_sym)) {
                *cycles = e->cycles;
                memcpy(data11, e->data, 11);
                return e->ret;
            }
        unsigned char sym[kNSymD];
        memcpy(sym, tx_sym, kNSymD);
        deinterleave162(sym);
        unsigned metric, maxnp;
        return fano_decode(&metric, cycles, &maxnp, data11, sym, kNBits, met.tab, delta, maxcycles);
    }

    // tuning constants of wsprd.c:423-433
    const float minsync1 = 0.10f;
    float minsync2 = 0.12f;
    int maxdrift = 4;
    const float minrms = 52.0 * (50 / 64.0);
    const int delta = 60;
    const unsigned maxcycles;
    const int lagstep, nlag0, njit_rest;
    const FanoMetrics& met = default_metrics();

    // per-segment state across passes (storage kept by the context between calls)
    std::vector<SegBook>& book;
    std::vector<int>& npk;
    std::vector<DevCand>& cand;
    // per-pass state
    int ipass = 0;
    bool lockstep = false;
    std::vector<char> stopped;
    std::vector<int> next_cand, win;
    // callsign hash memory: one zeroed table pair per segment, reused across batches
    const size_t per_seg = (size_t)kHashSlots * (kHashWidth + kLocWidth);
    bool persist;                             // hashtable.txt read and written by THIS run (single-segment calls; a batch: HashBatch)
    // buffers of the current wave
    int n_shared = 0, n_own = 0;
    FineState *h_items = nullptr, *d_items = nullptr;
    int *h_lists = nullptr, *d_lists = nullptr;
    float *d_tabs = nullptr, *d_pw = nullptr, *d_sync = nullptr, *d_rms = nullptr, *h_sync = nullptr, *h_rms = nullptr;
    unsigned char *d_sym = nullptr, *h_sym = nullptr;

    DecodeRun(Context& ctx_, int nseg_, int samples_, const decoder_options& opt_, decoder_results* out_, int max_results_,
              unsigned fast_, PendingFano& pend_)
        : ctx(ctx_), c(*ctx_.d), nseg(nseg_), samples(samples_), opt(opt_), out(out_), max_results(max_results_),
          fast(fast_), pend(pend_), maxcycles(fast_ ? fast_ : 10000u), lagstep(opt_.quickmode ? 16 : 8),
          nlag0(256 / (opt_.quickmode ? 16 : 8) + 1), njit_rest(opt_.quickmode ? 0 : kMaxLags - 1), book(ctx_.d->books),
          npk(ctx_.d->npk_host), cand(ctx_.d->cand_host), persist(opt_.usehashtable && nseg_ == 1) {
        if (book.size() < (size_t)nseg) book.resize((size_t)nseg);
        // the flat table pair exists for the one case that reads and writes hashtable.txt itself: a single call with
        // usehashtable (segment 0); every other segment keeps the slots it wrote (SegBook::hash)
        if (persist && c.hash_arena_segs < 1) {
            free(c.hash_arena);
            c.hash_arena = static_cast<char*>(calloc(1, per_seg));
            if (!c.hash_arena) throw std::runtime_error("out of host memory for hash tables");
            c.hash_arena_segs = 1;
        }
    }
    char* hashtab_of(int s) const { return c.hash_arena + (size_t)s * per_seg; }
    char* loctab_of(int s) const { return c.hash_arena + (size_t)s * per_seg + (size_t)kHashSlots * kHashWidth; }

    void load_hash_file();
    void save_hash_file();
    void start_pass(int pass, const std::vector<int>& active);
    std::vector<WaveItem> build_wave(const std::vector<int>& active);
    void refine_and_first_rung(std::vector<WaveItem>& wave);
    void remaining_rungs(std::vector<WaveItem>& wave);
    std::vector<SubJob> keep_books(std::vector<WaveItem>& wave);
    void subtract(const std::vector<SubJob>& jobs);
    void finish(const std::vector<int>& active0, int* n_results);
    void clear_hash(const std::vector<int>& segs);
};

// Callsign hash memory across calls (wsprd.c:481-494): hashtable.txt in the working directory.
// It makes the result depend on the order of calls, so it is honoured for single-segment calls
// (the daemon's one decode per two minutes) and ignored for batches (SURVEY 8e caveat, 8f3).
void Context::DecodeRun::load_hash_file() {
    if (!persist) return;
    if (FILE* fh = fopen("hashtable.txt", "r+")) {
        char line[80], hcall[13], hgrid[5];
        int nh;
        while (fgets(line, sizeof line, fh) != nullptr) {
            hgrid[0] = hcall[0] = '\0';
            if (sscanf(line, "%d %12s %4s", &nh, hcall, hgrid) < 2) continue;
            if (nh >= 0 && nh < kHashSlots) {
                snprintf(hashtab_of(0) + nh * kHashWidth, kHashWidth, "%s", hcall);
                if (strlen(hgrid) > 0) snprintf(loctab_of(0) + nh * kLocWidth, kLocWidth, "%s", hgrid);
            }
        }
        fclose(fh);
    }
}

void Context::DecodeRun::save_hash_file() {                   // wsprd.c:842-852
    if (!persist) return;
    if (FILE* fh = fopen("hashtable.txt", "w")) {
        for (int i = 0; i < kHashSlots; ++i)
            if (hashtab_of(0)[i * kHashWidth] != '\0')
                fprintf(fh, "%5d %s %s\n", i, hashtab_of(0) + i * kHashWidth, loctab_of(0) + i * kLocWidth);
        fclose(fh);
    }
    memset(hashtab_of(0), 0, per_seg);                        // the arena is reused by later batches
}

// FFT bank, peaks, coarse sync for the active segments; resets the per-pass state
void Context::DecodeRun::start_pass(int pass, const std::vector<int>& active) {
    ipass = pass;
    if (ipass < 2) { maxdrift = 4; minsync2 = 0.12f; }
    if (ipass == 2) { maxdrift = 0; minsync2 = 0.10f; }
    const int nact = (int)active.size();
    int* d_seglist = nullptr;
    if (nact != nseg) {
        int* h = static_cast<int*>(c.h_seglist.need((size_t)nact * 4));
        memcpy(h, active.data(), (size_t)nact * 4);
        d_seglist = static_cast<int*>(c.seglist.need((size_t)nact * 4));
        upload(d_seglist, h, (size_t)nact * 4, c.stream);
    }
    {
        Timer t(c.ev[0], c.ev[1], c.stream, &c.t_ms[0]);
        ctx.run_fft_sync(nseg, samples, maxdrift, true, d_seglist, nact, nullptr, nullptr);
        ctx.fetch_candidates_async(nseg);
        t.stop();                                   // the one host wait of the pass start
        c.resolve_deferred();
    }
    ctx.finish_fetch_candidates(nseg, npk, cand);
    if (trace && ipass < WSPR_TRACE_PASSES)
        for (int s : active) { trace[s].passes_run = ipass + 1; trace[s].npk[ipass] = npk[s]; }
    lockstep = opt.subtraction && ipass == 0;
    stopped.assign(nseg, 0);
    next_cand.assign(nseg, 0);
    win.assign(nseg, 1);
}

// Speculative windows (lockstep passes): a candidate only invalidates the ones after it when it
// decodes AND is subtracted.  Each segment therefore submits a window of `win` consecutive
// candidates per wave; the window is cut at the first subtraction (later results are dropped and
// recomputed on the new residual) and doubles, up to 64, after a window with none.
// The coarse sync value predicts which candidates can decode at all (of the candidates that decode,
// 0.1 % have a coarse sync below 0.12; most noise peaks are below it): a window also runs through
// all the unlikely candidates up to and including the next likely one, so that a segment's noise
// peaks cost one wave, not a doubling series of them.  Speculation is always validated, so the
// prediction only affects how much work is wasted, never the result.
std::vector<WaveItem> Context::DecodeRun::build_wave(const std::vector<int>& active) {
    constexpr float kLikelySync = 0.12f;
    // A refined candidate holds up to ~110 KB of scratch (tone amplitudes for 43 lags) plus its
    // phasor tables, so the wave size is bounded: speculative windows shrink first, and whatever
    // still does not fit waits for the next wave.
    constexpr int kMaxWave = 65536;
    auto window_of = [&](int s) {
        const int n = std::min(npk[s], kMaxCand), lo = next_cand[s];
        int w = 0;
        while (lo + w < n && w < kMaxCand && !(cand[(size_t)s * kMaxCand + lo + w].sync >= kLikelySync)) ++w;
        return std::max(win[s], std::min(w + 1, n - lo));
    };
    std::vector<int> weff(nseg, 0);
    if (lockstep) {
        for (int s : active) if (!stopped[s]) weff[s] = window_of(s);
        for (;;) {
            long total = 0;
            bool shrinkable = false;
            for (int s : active) {
                if (stopped[s]) continue;
                const int left = std::min(npk[s], kMaxCand) - next_cand[s];
                total += std::max(0, std::min(left, weff[s]));
                shrinkable |= weff[s] > 1;
            }
            if (total <= kMaxWave || !shrinkable) break;
            for (int s : active) { weff[s] = std::max(1, weff[s] / 2); win[s] = std::min(win[s], weff[s]); }
        }
    }
    std::vector<WaveItem> wave;
    for (int s : active) {
        if (stopped[s]) continue;
        const int n = std::min(npk[s], kMaxCand);
        const int lo = next_cand[s];
        const int hi = lockstep ? std::min(n, lo + weff[s]) : n;
        if (hi <= lo) continue;
        if (!wave.empty() && (int)wave.size() + (hi - lo) > kMaxWave) break;   // next wave
        for (int j = lo; j < hi; ++j) wave.push_back(WaveItem{s, j});
        if (!lockstep) next_cand[s] = n;
    }
    c.t_ms[10] += (double)wave.size();
    if (!wave.empty()) c.t_ms[11] += 1;
    return wave;
}

// GPU: fine sync (mode 0, mode 1) and first soft-symbol attempt; host: first rung of the jitter ladder
void Context::DecodeRun::refine_and_first_rung(std::vector<WaveItem>& wave) {
    const int nw = (int)wave.size();
    // One block up, one block down per wave (a small copy costs a blit kernel on the stream and ~10 us of
    // host time each): up = [items | launch lists], down = [items | rung-0 sync | rung-0 rms | rung-0 symbols].
    const size_t up_bytes = (size_t)nw * sizeof(FineState) + (size_t)nw * 2 * 4;
    const size_t o_sync = (size_t)nw * sizeof(FineState), o_rms = o_sync + (size_t)nw * 4, o_sym = o_rms + (size_t)nw * 4;
    const size_t down_bytes = o_sym + (size_t)nw * kNSymD;
    char* h_up = static_cast<char*>(c.h_items.need(up_bytes));
    char* h_down = static_cast<char*>(c.h_sym.need(std::max(down_bytes, (size_t)nw * kMaxLags * (kNSymD + 8))));
    char* d_blk = static_cast<char*>(c.items.need(std::max(up_bytes, down_bytes)));
    h_items = reinterpret_cast<FineState*>(h_up);
    h_lists = reinterpret_cast<int*>(h_up + (size_t)nw * sizeof(FineState));
    for (int i = 0; i < nw; ++i) {
        const DevCand& cd = cand[(size_t)wave[i].seg * kMaxCand + wave[i].cand];
        FineState f{};
        f.seg = wave[i].seg; f.freq = cd.freq; f.drift = cd.drift; f.shift = cd.shift; f.sync = cd.sync;
        f.shift_coarse = cd.shift; f.freq_coarse = cd.freq;
        h_items[i] = f;
    }
    n_shared = n_own = 0;
    const size_t ntabs = plan_tables(h_items, nw, h_lists, &n_shared, &n_own);
    d_items = reinterpret_cast<FineState*>(d_blk);
    // the launch lists are read by every kernel of the wave while the rung-0 outputs land behind the items:
    // keep the lists in their own buffer
    d_lists = static_cast<int*>(c.lists.need((size_t)nw * 2 * 4));
    d_tabs = static_cast<float*>(c.tabs.need(std::max(ntabs, (size_t)nw * 5) * 2048 * 4));
    d_pw = static_cast<float*>(c.pw.need((size_t)nw * kMaxLags * kNSymD * 16));
    const int nh_max = std::max(nlag0, kMaxLags);
    d_sync = static_cast<float*>(c.syncbuf.need((size_t)nw * nh_max * 4));
    c.symbuf.need((size_t)nw * kMaxLags * (kNSymD + 8));      // sized for the 43-lag block of remaining_rungs()
    float* d_sync0 = reinterpret_cast<float*>(d_blk + o_sync);
    float* d_rms0 = reinterpret_cast<float*>(d_blk + o_rms);
    unsigned char* d_sym0 = reinterpret_cast<unsigned char*>(d_blk + o_sym);
    const float* wi = c.iqI.as<float>();
    const float* wq = c.iqQ.as<float>();
    {
        Timer t(c.ev[0], c.ev[1], c.stream, &c.t_ms[3]);
        upload(d_items, h_items, (size_t)nw * sizeof(FineState), c.stream);
        upload(d_lists, h_lists, (size_t)nw * 2 * 4, c.stream);
        // mode 0: lag scan (tiled), mode 1: 5 frequencies, mode 2: first rung of the ladder
        launch_phasor_tables(d_items, nw, 0, d_tabs, c.stream);
        launch_demod_tiled(wi, wq, samples, d_items, nw, d_lists, n_shared, d_lists + nw, n_own, 0, nlag0, lagstep,
                           0.0f, d_tabs, d_pw, d_sync, nullptr, nullptr, c.tab, c.stream);
        launch_pick_lag(d_items, nw, d_sync, nlag0, lagstep, c.stream);
        std::vector<FineState> tr_items0;
        if (trace) {                                           // mode-0 result, before the frequency scan refines it
            tr_items0.resize(nw);
            HIP_OK(hipMemcpyAsync(tr_items0.data(), d_items, (size_t)nw * sizeof(FineState), hipMemcpyDeviceToHost, c.stream));
            HIP_OK(hipStreamSynchronize(c.stream));
            wtrace.assign(nw, ItemTrace{});
            for (int i = 0; i < nw; ++i) { wtrace[i].m0_shift = tr_items0[i].shift; wtrace[i].m0_sync = tr_items0[i].sync; }
        }
        {
            float* d_tabs1 = static_cast<float*>(c.tabs.need(std::max(ntabs, (size_t)n_shared * 5) * 2048 * 4));
            float* d_scr = static_cast<float*>(c.scrsync.need((size_t)nw * 5 * 4));
            // the frequency scan reads its centre hypothesis out of the lag scan's block (d_pw), so it writes elsewhere
            float* d_pwf = static_cast<float*>(c.pwfreq.need((size_t)nw * 5 * kNSymD * 16));
            launch_freq_scan_and_first_rung(wi, wq, samples, d_items, d_lists, n_shared, d_lists + nw, n_own, lagstep,
                                            minsync1, c.t_jitter.as<int>(), d_tabs1, d_pwf, d_scr, d_sync0, d_sym0,
                                            d_rms0, c.tab, c.stream, d_pw, nlag0);
        }
        // device-Fano mode: the soft symbols stay in HBM, only items / sync / rms come down
        HIP_OK(hipMemcpyAsync(h_down, d_blk, (c.dev_fano && !trace) ? o_sym : down_bytes, hipMemcpyDeviceToHost, c.stream));
        t.stop();
        c.resolve_deferred();
    }
    // host views of the block that came down (h_items is re-pointed: the uploaded copy is no longer needed)
    h_items = reinterpret_cast<FineState*>(h_down);
    h_sync = reinterpret_cast<float*>(h_down + o_sync);
    h_rms = reinterpret_cast<float*>(h_down + o_rms);
    h_sym = reinterpret_cast<unsigned char*>(h_down + o_sym);

    if (trace)
        for (int i = 0; i < nw; ++i) {
            ItemTrace& t = wtrace[i];
            const bool worth = h_items[i].sync > minsync1;   // rung 0 is only computed for these (wsprd.c:733-737)
            t.attempts = worth ? 1 : 0;
            if (!worth) continue;
            t.sync0 = h_sync[i]; t.rms0 = h_rms[i];
            memcpy(t.sym0, h_sym + (size_t)i * kNSymD, kNSymD);
            t.fano_calls = (h_sync[i] > minsync2 && h_rms[i] > minrms) ? 1 : 0;
        }
    // ---- first rung of the jitter ladder -----------------------------------
    const auto t_f0 = std::chrono::steady_clock::now();
    if (c.dev_fano) {
        // every gated vector of the wave to the device search, straight from the rung-0 symbols in HBM
        std::vector<int> att;
        for (int i = 0; i < nw; ++i) {
            WaveItem& w = wave[i];
            w.fine = h_items[i];
            w.worth = w.fine.sync > minsync1;
            w.decoded = false;
            w.jitter = 0;
            w.rung0_pending = false;
            if (w.worth && h_sync[i] > minsync2 && h_rms[i] > minrms) att.push_back(i);
        }
        const int na = (int)att.size();
        std::vector<int> ret(na);
        std::vector<unsigned> cyc(na);
        std::vector<unsigned char> dat((size_t)na * 10);
        ctx.fano_resident(d_sym0, att.data(), na, 10000u, ret.data(), cyc.data(), dat.data());
        for (int k = 0; k < na; ++k) {
            WaveItem& w = wave[att[k]];
            w.decoded = ret[k] == 0;
            w.cycles = cyc[k];
            memset(w.decdata, 0, sizeof w.decdata);
            memcpy(w.decdata, dat.data() + (size_t)k * 10, 10);
            c.n_fano++; c.n_cycles += cyc[k]; if (ret[k]) c.n_timeout++;
        }
        c.t_ms[2] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_f0).count();
        return;
    }
    // a candidate that passes the gates but does not decode costs a full time-out here
    // (milliseconds) while a decode costs microseconds: one task per grab, all threads
    Pool& pool0 = (nw >= 256) ? *c.bigpool : *c.pool;
    pool0.run(nw, [&](int i) {
        WaveItem& w = wave[i];
        w.fine = h_items[i];
        w.worth = w.fine.sync > minsync1;
        w.decoded = false;
        w.jitter = 0;
        if (!w.worth) return;
        if (h_sync[i] > minsync2 && h_rms[i] > minrms) {
            const int nd = fano_attempt(h_sym + (size_t)i * kNSymD, &w.cycles, w.decdata);
            w.decoded = (nd == 0);
            w.rung0_pending = (nd != 0) && fast;
            if (!w.rung0_pending) { c.n_fano++; c.n_cycles += w.cycles; if (nd) c.n_timeout++; }
        }
    }, nw >= 256 ? 1 : 0);
    c.t_ms[5] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_f0).count();

    if (fast)
        for (int i = 0; i < nw; ++i)
            if (wave[i].rung0_pending) pend.add(wave[i].seg, h_sym + (size_t)i * kNSymD);
}

// Remaining rungs of the jitter ladder, only for candidates that still need them
void Context::DecodeRun::remaining_rungs(std::vector<WaveItem>& wave) {
    const int nw = (int)wave.size();
    const float* wi = c.iqI.as<float>();
    const float* wq = c.iqQ.as<float>();
    std::vector<int> again;
    if (njit_rest > 0)
        for (int i = 0; i < nw; ++i)
            if (wave[i].worth && !wave[i].decoded) again.push_back(i);
    if (!again.empty()) {
        const int na = (int)again.size();
        FineState* h2 = static_cast<FineState*>(c.h_misc.need((size_t)na * sizeof(FineState)));
        for (int a = 0; a < na; ++a) h2[a] = wave[again[a]].fine;
        const size_t ntabs2 = plan_tables(h2, na, h_lists, &n_shared, &n_own);
        d_tabs = static_cast<float*>(c.tabs.need(ntabs2 * 2048 * 4));
        {
            // all 43 lags shift-63 .. shift+63 in steps of 3 (rung r of the ladder = lag index
            // (jitter+63)/3; index 21 repeats rung 0 and is ignored)
            Timer t(c.ev[0], c.ev[1], c.stream, &c.t_ms[3]);
            upload(d_items, h2, (size_t)na * sizeof(FineState), c.stream);
            upload(d_lists, h_lists, (size_t)na * 2 * 4, c.stream);
            launch_phasor_tables(d_items, na, 2, d_tabs, c.stream);
            // the 43-lag outputs in one block [sync | rms | symbols] -> one copy down
            const size_t o_rms = (size_t)na * kMaxLags * 4, o_sym = 2 * o_rms, blk = o_sym + (size_t)na * kMaxLags * kNSymD;
            char* d_blk = static_cast<char*>(c.symbuf.need(blk));
            char* h_blk = static_cast<char*>(c.h_sym.need(blk));
            d_sync = reinterpret_cast<float*>(d_blk);
            d_rms = reinterpret_cast<float*>(d_blk + o_rms);
            d_sym = reinterpret_cast<unsigned char*>(d_blk + o_sym);
            launch_demod_tiled(wi, wq, samples, d_items, na, d_lists, n_shared, d_lists + na, n_own, 2, kMaxLags, 3,
                               minsync1, d_tabs, d_pw, d_sync, d_sym, d_rms, c.tab, c.stream);
            HIP_OK(hipMemcpyAsync(h_blk, d_blk, c.dev_fano ? o_sym : blk, hipMemcpyDeviceToHost, c.stream));
            t.stop();
            c.resolve_deferred();
            h_sync = reinterpret_cast<float*>(h_blk);
            h_rms = reinterpret_cast<float*>(h_blk + o_rms);
            h_sym = reinterpret_cast<unsigned char*>(h_blk + o_sym);
        }
        // trace: the serial walk stops at the first success, rung `lastr` of the rest (or walks all of them)
        auto trace_ladder = [&](int a, int lastr) {
            if (!trace) return;
            ItemTrace& t = wtrace[again[a]];
            t.attempts = 1 + (lastr + 1);
            for (int r = 0; r <= lastr; ++r) {
                const int g = a * kMaxLags + (c.jitter_ladder[r + 1] + 63) / 3;
                if (h_sync[g] > minsync2 && h_rms[g] > minrms) t.fano_calls++;
            }
        };
        if (c.dev_fano) {
            // every gated (candidate, rung) vector to the device search; the ladder keeps the first success in
            // rung order, so all are run and the pick is made afterwards (a candidate that decodes on an early
            // rung wastes its later ones: few do)
            const auto t_d0 = std::chrono::steady_clock::now();
            std::vector<int> off, who;
            for (int a = 0; a < na; ++a)
                for (int r = 0; r < njit_rest; ++r) {
                    const int g = a * kMaxLags + (c.jitter_ladder[r + 1] + 63) / 3;
                    if (h_sync[g] > minsync2 && h_rms[g] > minrms) { off.push_back(g); who.push_back(a * njit_rest + r); }
                }
            const int nv = (int)off.size();
            std::vector<int> ret(nv);
            std::vector<unsigned> cyc(nv);
            std::vector<unsigned char> dat((size_t)nv * 10);
            ctx.fano_resident(d_sym, off.data(), nv, 10000u, ret.data(), cyc.data(), dat.data());
            std::vector<int> first(na, njit_rest), at(na, -1);
            for (int k = 0; k < nv; ++k) {
                const int a = who[k] / njit_rest, r = who[k] % njit_rest;
                if (r <= first[a]) { c.n_fano++; c.n_cycles += cyc[k]; if (ret[k]) c.n_timeout++; }   // what the serial walk would have run
                if (ret[k] == 0 && r < first[a]) { first[a] = r; at[a] = k; }
            }
            for (int a = 0; a < na; ++a) {
                trace_ladder(a, at[a] >= 0 ? first[a] : njit_rest - 1);
                if (at[a] < 0) continue;
                WaveItem& w = wave[again[a]];
                w.decoded = true;
                w.jitter = c.jitter_ladder[first[a] + 1];
                w.cycles = cyc[at[a]];
                memset(w.decdata, 0, sizeof w.decdata);
                memcpy(w.decdata, dat.data() + (size_t)at[a] * 10, 10);
            }
            c.t_ms[2] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_d0).count();
            return;
        }
        const auto t_f1 = std::chrono::steady_clock::now();
        // every (candidate, rung) Fano attempt is independent; the ladder keeps the
        // FIRST success in rung order, so run them all and pick afterwards
        struct Attempt { int ok; int pending; unsigned cycles; unsigned char data[11]; };
        std::vector<Attempt> att((size_t)na * njit_rest);
        std::vector<std::atomic<int>> first(na);
        for (auto& f : first) f.store(njit_rest);
        Pool& fpool = (na * njit_rest >= 256) ? *c.bigpool : *c.pool;
        // rung-major order and one task per grab: a time-out costs ~810 000 decoder cycles
        // (milliseconds) while a success costs microseconds, so costs are heavy-tailed; low
        // rungs finish first and cancel the higher rungs of the same candidate
        fpool.run(na * njit_rest, [&](int task) {
            const int r = task / na, a = task % na;
            const int idx = a * njit_rest + r;
            Attempt& at = att[idx];
            at.ok = 0;
            at.pending = 0;
            if (r > first[a].load()) return;           // an earlier rung already decoded
            const size_t g = (size_t)a * kMaxLags + (size_t)((c.jitter_ladder[r + 1] + 63) / 3);
            if (!(h_sync[g] > minsync2 && h_rms[g] > minrms)) return;
            const int nd = fano_attempt(h_sym + g * kNSymD, &at.cycles, at.data);
            at.pending = (nd != 0) && fast;
            if (!at.pending) { c.n_fano++; c.n_cycles += at.cycles; if (nd) c.n_timeout++; }
            if (nd == 0) {
                at.ok = 1;
                int cur = first[a].load();
                while (r < cur && !first[a].compare_exchange_weak(cur, r)) {}
            }
        }, 1);
        if (fast)      // unfinished attempts on rungs BEFORE the accepted one decide nothing yet
            for (int a = 0; a < na; ++a) {
                const int rmax = std::min(first[a].load(), njit_rest);
                for (int r = 0; r < rmax; ++r)
                    if (att[(size_t)a * njit_rest + r].pending) {
                        const size_t g = (size_t)a * kMaxLags + (size_t)((c.jitter_ladder[r + 1] + 63) / 3);
                        pend.add(wave[again[a]].seg, h_sym + g * kNSymD);
                    }
            }
        for (int a = 0; a < na; ++a) {
            const int r = first[a].load();
            trace_ladder(a, (r < njit_rest && att[(size_t)a * njit_rest + r].ok) ? r : njit_rest - 1);
            if (r < njit_rest && att[(size_t)a * njit_rest + r].ok) {
                WaveItem& w = wave[again[a]];
                w.decoded = true;
                w.jitter = c.jitter_ladder[r + 1];
                w.cycles = att[(size_t)a * njit_rest + r].cycles;
                memcpy(w.decdata, att[(size_t)a * njit_rest + r].data, 11);
            }
        }
        c.t_ms[5] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_f1).count();
    }
}

// Host bookkeeping in candidate order (wsprd.c:768-822).  Items of one segment are contiguous in the
// wave and must be handled in order; different segments are independent -> one pool task per segment.
// Returns the subtraction jobs of the wave.
std::vector<SubJob> Context::DecodeRun::keep_books(std::vector<WaveItem>& wave) {
    const int nw = (int)wave.size();
    const auto t_b0 = std::chrono::steady_clock::now();
    std::vector<int> group_start;
    for (int i = 0; i < nw; ++i)
        if (i == 0 || wave[i].seg != wave[i - 1].seg) group_start.push_back(i);
    group_start.push_back(nw);
    const int ngroups = (int)group_start.size() - 1;
    std::vector<SubJob> job_of(nw);
    std::vector<char> has_job(nw, 0);
    c.pool->run(ngroups, [&](int g) {
      const int sg = wave[group_start[g]].seg;
      bool cut = false;
      for (int i = group_start[g]; i < group_start[g + 1] && !cut; ++i) {
        WaveItem& w = wave[i];
        const int s = w.seg;
        if (stopped[s]) break;
        c.n_kept++;
        if (lockstep) next_cand[s] = w.cand + 1;
        DevCand& cd = cand[(size_t)s * kMaxCand + w.cand];
        cd.freq = w.fine.freq; cd.shift = w.fine.shift; cd.drift = w.fine.drift; cd.sync = w.fine.sync;
        wspr_cand_trace* tc = nullptr;
        if (trace && ipass < WSPR_TRACE_PASSES) {            // this visit is the one that counts (not a dropped speculation)
            const ItemTrace& t = wtrace[i];
            tc = &trace[s].cand[ipass][w.cand];
            memset(tc, 0, sizeof *tc);
            trace[s].n_visited[ipass] = w.cand + 1;
            tc->visited = 1;
            tc->mode0_shift = t.m0_shift; tc->mode0_sync = t.m0_sync;
            tc->freq = w.fine.freq; tc->shift = w.fine.shift; tc->drift = w.fine.drift; tc->sync = w.fine.sync;
            tc->attempts = t.attempts; tc->fano_calls = t.fano_calls;
            if (t.attempts > 0) { tc->first_sync = t.sync0; tc->first_rms = t.rms0; memcpy(tc->first_symbols, t.sym0, kNSymD); }
            if (w.worth && w.decoded) {
                tc->decoded = 1; tc->jitter = w.jitter; tc->cycles = w.cycles;
                memcpy(tc->decdata, w.decdata, 11);
            }
        }
        if (!(w.worth && w.decoded)) continue;

        signed char message[12] = {0};
        for (int k = 0; k < 11; ++k) message[k] = (signed char)w.decdata[k];
        char callsign[13] = {0}, call_loc_pow[23] = {0}, call[13] = {0}, loc[7] = {0}, pwr[3] = {0};
        SegBook& bk = book[s];
        // the segment's hash memory: its own zeroed tables (the reference's locals, wsprd.c:478-479; every slot written
        // is noted and cleared again when the batch ends), or its window on the batch's shared memory (usehashtable)
        FlatHashTable flat(persist ? hashtab_of(s) : nullptr, persist ? loctab_of(s) : nullptr, &bk.dirty);
        std::unique_ptr<SegHashView> shared(hb ? new SegHashView(hb, hb_off + s) : nullptr);
        HashTable& tab = hb ? static_cast<HashTable&>(*shared) : (persist ? static_cast<HashTable&>(flat) : static_cast<HashTable&>(bk.hash));
        // (what the bits unpack and re-encode to is computed once per host thread: MessageCache, wspr_hashmem.h)
        MessageCache& mc = MessageCache::of_this_thread();
        const unsigned long mc_hits0 = mc.hits;
        MessageCache::Handle mh = mc.unpack(w.decdata, tab, call_loc_pow, call, loc, pwr, callsign);
        c.n_mc_lookups++; c.n_mc_hits += (long)(mc.hits - mc_hits0);       // message-cache look-ups and hits of this call
        const int noprint = mh.noprint;
        auto symbols_of = [&](unsigned char* sym) { return mc.symbols(mh, call_loc_pow, tab, sym); };
        if (opt.subtraction && ipass == 0 && !noprint) {
            SubJob jb{};
            if (symbols_of(jb.sym)) {
                jb.seg = s; jb.f0 = w.fine.freq; jb.shift = w.fine.shift; jb.drift = w.fine.drift;
                job_of[i] = jb;
                has_job[i] = 1;
                if (tc) tc->subtracted = 1;
                cut = true;            // the IQ changes: later candidates of this window are redone
            } else {
                stopped[s] = 1;                      // wsprd.c:786-788: leaves the candidate loop
                continue;
            }
        }
        if (!strcmp(loc, "A000AA")) { stopped[s] = 1; continue; }      // wsprd.c:792-793
        bool dupe = false;
        for (int u = 0; u < bk.uniques; ++u)
            if (!strcmp(callsign, bk.allcalls[u]) && fabs(w.fine.freq - bk.allfreqs[u]) < 3.0) dupe = true;
        if (dupe || bk.uniques >= 100) continue;
        copy_text(bk.allcalls[bk.uniques], sizeof bk.allcalls[0], callsign);
        bk.allfreqs[bk.uniques] = w.fine.freq;
        bk.uniques++;
        {
            bk.spots.emplace_back();
            decoder_results* o = &bk.spots.back();
            memset(o, 0, sizeof *o);
            const double dial = (double)opt.freq / 1e6;
            o->sync = w.fine.sync;
            // candidates[j].snr, wsprd.c:616, recomputed with the host libm from the
            // peak value so that the reported figure does not depend on ocml's log10f
            o->snr = cd.snr;               // host libm, see fetch_candidates()
            o->dt = w.fine.shift * 1.0 / 375.0 - 2.0;
            o->freq = dial + (1500.0 + w.fine.freq) / 1e6;
            o->drift = w.fine.drift;
            o->cycles = (int)w.cycles;
            o->jitter = w.jitter;
            copy_text(o->message, sizeof o->message, call_loc_pow);       // snprintf(.., "%s", ..) of wsprd.c:817-820
            copy_text(o->call, sizeof o->call, call);
            copy_text(o->loc, sizeof o->loc, loc);
            copy_text(o->pwr, sizeof o->pwr, pwr);
        }
      }
      if (lockstep) win[sg] = cut ? 1 : std::min(64, 2 * win[sg]);
    });
    std::vector<SubJob> jobs;
    for (int i = 0; i < nw; ++i) if (has_job[i]) jobs.push_back(job_of[i]);
    c.n_subjobs += (long)jobs.size();
    c.t_ms[1] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_b0).count();
    return jobs;
}

// GPU: subtract everything that decoded in this wave
void Context::DecodeRun::subtract(const std::vector<SubJob>& jobs) {
    if (jobs.empty()) return;
    const int nj = (int)jobs.size();
    // two pinned staging buffers in turn: the upload is asynchronous, and the copy that read the other one
    // has completed by the time it is reused (a blocking wait on this stream lies between two subtractions)
    PinBuf& stage = (c.sub_flip ^= 1) ? c.h_jobs : c.h_jobs2;
    SubJob* hj = static_cast<SubJob*>(stage.need((size_t)nj * sizeof(SubJob)));
    memcpy(hj, jobs.data(), (size_t)nj * sizeof(SubJob));
    SubJob* dj = static_cast<SubJob*>(c.jobs.need((size_t)nj * sizeof(SubJob)));
    float* scratch = static_cast<float*>(c.subscratch.need(subtract_scratch_floats(nj) * 4));
    // no host wait here: the next wave's kernels queue behind the subtraction on the same stream, and nothing
    // the host does next depends on it.  The span is timed with a deferred event pair (or not at all if eight
    // are already pending).
    const int slot_ev = c.n_def < Impl::kDeferred ? c.n_def : -1;
    if (slot_ev >= 0) HIP_OK(hipEventRecord(c.ev_def[slot_ev][0], c.stream));
    upload(dj, hj, (size_t)nj * sizeof(SubJob), c.stream);
    launch_subtract(c.iqI.as<float>(), c.iqQ.as<float>(), samples, dj, nj, scratch, c.tab, c.stream);
    if (slot_ev >= 0) {
        HIP_OK(hipEventRecord(c.ev_def[slot_ev][1], c.stream));
        c.def_acc[slot_ev] = &c.t_ms[4];
        c.n_def = slot_ev + 1;
    }
    HIP_OK(hipGetLastError());
}

// results strongest first (wsprd.c:827; stable like glibc's merge sort) -- ALL unique spots of the segment
// are ranked, then the strongest max_results are handed out (a caller with a short array loses the weakest
// spots, never a strong one that happened to decode late); hash slots written by the batch are cleared again
void Context::DecodeRun::clear_hash(const std::vector<int>& segs) {
    if (persist) return;
    for (int s : segs) { book[s].hash.entries.clear(); book[s].dirty.clear(); }
}

void Context::DecodeRun::finish(const std::vector<int>& active0, int* n_results) {
    for (int s : active0) {
        SegBook& bk = book[s];
        std::stable_sort(bk.spots.begin(), bk.spots.end(),
                         [](const decoder_results& a, const decoder_results& b) { return a.snr > b.snr; });
        const int n = std::min((int)bk.spots.size(), max_results);
        if (n > 0) memcpy(out + (size_t)s * max_results, bk.spots.data(), (size_t)n * sizeof(decoder_results));
        n_results[s] = n;
    }
    clear_hash(active0);
    save_hash_file();
}

// One full decode (all passes) of the segments in `active0`.  fast = 0: the host Fano pool runs the
// reference's full cycle budget (exact on its own).  fast > 0: it runs `fast` cycles per bit and
// records every attempt it could not finish in `pend` (see decode_resident).
int Context::decode_core(int nseg, int samples, const decoder_options& opt, decoder_results* out, int max_results,
                         int* n_results, const std::vector<int>& active0, unsigned fast, PendingFano& pend,
                         const FanoMemo* memo, wspr_trace* trace, HashBatch* hb, int hb_off) {
    for (int s : active0) n_results[s] = 0;
    DecodeRun run(*this, nseg, samples, opt, out, max_results, fast, pend);
    for (int s : active0) { SegBook& b = run.book[s]; b.uniques = 0; b.dirty.clear(); b.hash.entries.clear(); b.spots.clear(); }
    run.memo = memo;
    run.trace = trace;
    run.hb = hb;
    run.hb_off = hb_off;
    if (hb) run.persist = false;
    if (hb) for (int s : active0) hb->log[(size_t)(hb_off + s)].clear();      // a segment decoded again starts a new log
    // whatever happens below (a HIP error surfaces as an exception), the call signs this run wrote into the
    // context's hash memory must not survive into the next batch on this lane/slot
    struct HashGuard {
        DecodeRun& r; const std::vector<int>& segs; bool armed = true;
        ~HashGuard() {
            if (!armed) return;
            if (r.persist) memset(r.hashtab_of(0), 0, r.per_seg);
            else r.clear_hash(segs);
        }
    } guard{run, active0};
    run.load_hash_file();
    std::vector<int> active = active0;
    for (int ipass = 0; ipass < opt.npasses; ++ipass) {
        if (ipass == 1) {                                      // wsprd.c:522-523
            std::vector<int> keep;
            for (int s : active) if (run.book[s].uniques > 0) keep.push_back(s);
            active.swap(keep);
        }
        if (active.empty()) break;
        { CpuSpan sp(&d->t_ms[17]); run.start_pass(ipass, active); }
        for (;;) {
            std::vector<WaveItem> wave;
            { CpuSpan sp(&d->t_ms[18]); wave = run.build_wave(active); }
            if (wave.empty()) break;
            { CpuSpan sp(&d->t_ms[19]); run.refine_and_first_rung(wave); }
            { CpuSpan sp(&d->t_ms[20]); run.remaining_rungs(wave); }
            std::vector<SubJob> jobs;
            { CpuSpan sp(&d->t_ms[21]); jobs = run.keep_books(wave); }
            { CpuSpan sp(&d->t_ms[22]); run.subtract(jobs); }
        }
    }
    { CpuSpan sp(&d->t_ms[23]); run.finish(active0, n_results); }
    guard.armed = false;
    return 0;
}

int Context::last_timings(double* ms, int cap) {
    const int n = std::min(cap, 26);
    for (int i = 0; i < n; ++i) ms[i] = d->t_ms[i];
    return n;
}

#ifdef WSPR_LAB   // kernel-level timing sets (include/wspr_mi355x_bench.h): lab build only
// average kernel durations of the FFT+sync stage, HIP events on the launch stream:
// ms[0] = K1 (all chunks), ms[1] = K2 (time average of all chunks + peak picking), ms[2] = K3,
// ms[3] = K1 launches per pass, ms[4] = wall time of the whole stage
int Context::bench_fft_sync(int nseg, int samples, int iters, double* ms) {
    const int blocks = 4 * (samples / kFftSize) - 1;
    float* ps = ps_buffer(nseg);
    DevCand* cand = static_cast<DevCand*>(d->cand.need((size_t)nseg * kMaxCand * sizeof(DevCand)));
    int* npk = static_cast<int*>(d->npk.need((size_t)nseg * 4));
    float* psavg = static_cast<float*>(d->psavg.need((size_t)nseg * kPsStride * 4));
    for (int k = 0; k < 5; ++k) ms[k] = 0.0;
    auto between = [](hipEvent_t a, hipEvent_t b) { float t = 0; HIP_OK(hipEventElapsedTime(&t, a, b)); return (double)t; };
    // one untimed pass first: the spectrogram buffer may be freshly allocated (first touch), the code not yet resident
    fft_and_average(d->iqI.as<float>(), d->iqQ.as<float>(), nullptr, nseg, samples, ps, psavg, d->tab, d->stream);
    launch_pick_peaks(ps, nullptr, nseg, blocks, psavg, cand, npk, nullptr, nullptr, d->tab, d->stream, true);
    launch_coarse_sync(ps, nullptr, nseg, blocks, cand, npk, 4, d->tab, d->stream);
    HIP_OK(hipStreamSynchronize(d->stream));
    for (int it = 0; it < iters; ++it) {
        std::vector<hipEvent_t> ev;
        fft_and_average(d->iqI.as<float>(), d->iqQ.as<float>(), nullptr, nseg, samples, ps, psavg, d->tab, d->stream, &ev);
        hipEvent_t e1, e2, e3;
        HIP_OK(hipEventCreate(&e1)); HIP_OK(hipEventCreate(&e2)); HIP_OK(hipEventCreate(&e3));
        HIP_OK(hipEventRecord(e1, d->stream));
        launch_pick_peaks(ps, nullptr, nseg, blocks, psavg, cand, npk, nullptr, nullptr, d->tab, d->stream, true);
        HIP_OK(hipEventRecord(e2, d->stream));
        launch_coarse_sync(ps, nullptr, nseg, blocks, cand, npk, 4, d->tab, d->stream);
        HIP_OK(hipEventRecord(e3, d->stream));
        HIP_OK(hipStreamSynchronize(d->stream));
        for (size_t i = 0; i + 3 < ev.size(); i += 4) {
            ms[0] += between(ev[i], ev[i + 1]) / iters;
            ms[1] += between(ev[i + 2], ev[i + 3]) / iters;
        }
        ms[1] += between(e1, e2) / iters;
        ms[2] += between(e2, e3) / iters;
        ms[3] = (double)(ev.size() / 4);
        ms[4] += between(ev[0], e3) / iters;
        for (auto& e : ev) (void)hipEventDestroy(e);
        (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); (void)hipEventDestroy(e3);
    }
    return 5;
}

// Kernel-level timing of the two fp32-VALU-bound stages on the resident batch (HIP events on the launch
// stream): the strongest candidate of every segment goes through the tiled lag scan (K4 mode 0:
// phasor tables are built before the timed region) and one coherent subtraction (K7) is run per segment
// at that candidate's coarse parameters.  ms[0] = lag scan, ms[1] = subtraction, ms[2] = candidates,
// ms[3] = jobs, ms[4] = the frequency scan + first rung (K4 mode 1 fused with K5 rung 0).
// The working IQ is modified by the subtraction (bench only).
int Context::bench_valu(int nseg, int samples, int iters, double* ms) {
    Impl& c = *d;
    for (int k = 0; k < 5; ++k) ms[k] = 0.0;
    run_fft_sync(nseg, samples, 4, true, nullptr, nseg, nullptr, nullptr);
    std::vector<int> npk;
    std::vector<DevCand> cand;
    fetch_candidates(nseg, npk, cand);
    std::vector<FineState> items;
    std::vector<SubJob> jobs;
    unsigned char sym[kNSymD];
    {
        std::vector<char> hashtab((size_t)kHashSlots * kHashWidth, 0), loctab((size_t)kHashSlots * kLocWidth, 0);
        char msg[32] = "K1JT FN20 20";
        if (!channel_symbols(msg, hashtab.data(), loctab.data(), sym)) return -1;
    }
    // WSPR_BENCH_VALU_REUSE=k (a measurement switch): the candidates take their samples from k segments only, so that the
    // kernels find them in the caches -- what a launch set costs when its sample reads are free (the results are not used)
    const char* reuse_env = lab_env("WSPR_BENCH_VALU_REUSE");
    const int reuse = reuse_env ? std::max(1, atoi(reuse_env)) : nseg;
    for (int s = 0; s < nseg; ++s) {
        if (npk[s] <= 0) continue;
        const DevCand& cd = cand[(size_t)s * kMaxCand];
        FineState f{};
        f.seg = s % reuse; f.freq = cd.freq; f.drift = 0.0f; f.shift = cd.shift; f.sync = cd.sync;
        f.shift_coarse = cd.shift; f.freq_coarse = cd.freq;
        items.push_back(f);
        SubJob jb{};
        jb.seg = s; jb.f0 = cd.freq; jb.shift = cd.shift; jb.drift = 0.0f;
        memcpy(jb.sym, sym, kNSymD);
        jobs.push_back(jb);
    }
    const int nw = (int)items.size();
    if (nw == 0) return 0;
    std::vector<int> lists(2 * (size_t)nw);
    int n_shared = 0, n_own = 0;
    const size_t ntabs = plan_tables(items.data(), nw, lists.data(), &n_shared, &n_own);
    FineState* d_items = static_cast<FineState*>(c.items.need((size_t)nw * sizeof(FineState)));
    int* d_lists = static_cast<int*>(c.lists.need((size_t)nw * 2 * 4));
    float* d_tabs = static_cast<float*>(c.tabs.need(std::max(ntabs, (size_t)nw * 5) * 2048 * 4));
    float* d_pw = static_cast<float*>(c.pw.need((size_t)nw * kMaxLags * kNSymD * 16));
    float* d_sync = static_cast<float*>(c.syncbuf.need((size_t)nw * kMaxLags * 4));
    unsigned char* d_sym = static_cast<unsigned char*>(c.symbuf.need((size_t)nw * kMaxLags * kNSymD));
    float* d_rms = static_cast<float*>(c.rmsbuf.need((size_t)nw * kMaxLags * 4));
    float* d_scr = static_cast<float*>(c.scrsync.need((size_t)nw * 5 * 4));
    float* d_pwf = static_cast<float*>(c.pwfreq.need((size_t)nw * 5 * kNSymD * 16));
    SubJob* dj = static_cast<SubJob*>(c.jobs.need((size_t)nw * sizeof(SubJob)));
    float* scratch = static_cast<float*>(c.subscratch.need(subtract_scratch_floats(nw) * 4));
    HIP_OK(hipMemcpyAsync(d_items, items.data(), (size_t)nw * sizeof(FineState), hipMemcpyHostToDevice, c.stream));
    HIP_OK(hipMemcpyAsync(d_lists, lists.data(), (size_t)nw * 2 * 4, hipMemcpyHostToDevice, c.stream));
    HIP_OK(hipMemcpyAsync(dj, jobs.data(), (size_t)nw * sizeof(SubJob), hipMemcpyHostToDevice, c.stream));
    launch_phasor_tables(d_items, nw, 0, d_tabs, c.stream);
    HIP_OK(hipStreamSynchronize(c.stream));
    hipEvent_t e[4];
    for (auto& x : e) HIP_OK(hipEventCreate(&x));
    const float* wi = c.iqI.as<float>();
    const float* wq = c.iqQ.as<float>();
    for (int it = -1; it < iters; ++it) {                    // pass -1 is not timed (first touch of the buffers, clocks)
        HIP_OK(hipEventRecord(e[0], c.stream));
        launch_demod_tiled(wi, wq, samples, d_items, nw, d_lists, n_shared, d_lists + nw, n_own, 0, 33, 8, 0.0f, d_tabs,
                           d_pw, d_sync, nullptr, nullptr, c.tab, c.stream);
        HIP_OK(hipEventRecord(e[1], c.stream));
        launch_pick_lag(d_items, nw, d_sync, 33, 8, c.stream);
        launch_freq_scan_and_first_rung(wi, wq, samples, d_items, d_lists, n_shared, d_lists + nw, n_own, 8, 0.10f,
                                        c.t_jitter.as<int>(), d_tabs, d_pwf, d_scr, d_sync, d_sym, d_rms, c.tab, c.stream,
                                        d_pw, 33);
        HIP_OK(hipEventRecord(e[2], c.stream));
        launch_subtract(c.iqI.as<float>(), c.iqQ.as<float>(), samples, dj, nw, scratch, c.tab, c.stream);
        HIP_OK(hipEventRecord(e[3], c.stream));
        HIP_OK(hipEventSynchronize(e[3]));
        float t = 0;
        if (it >= 0) {
            HIP_OK(hipEventElapsedTime(&t, e[0], e[1])); ms[0] += t / iters;
            HIP_OK(hipEventElapsedTime(&t, e[1], e[2])); ms[4] += t / iters;
            HIP_OK(hipEventElapsedTime(&t, e[2], e[3])); ms[1] += t / iters;
        }
        // the frequency scan refined the items: restore the coarse state for the next round
        HIP_OK(hipMemcpyAsync(d_items, items.data(), (size_t)nw * sizeof(FineState), hipMemcpyHostToDevice, c.stream));
        launch_phasor_tables(d_items, nw, 0, d_tabs, c.stream);
        HIP_OK(hipStreamSynchronize(c.stream));
    }
    for (auto& x : e) (void)hipEventDestroy(x);
    ms[2] = nw; ms[3] = nw;
    return 5;
}

// average duration of the whole front end (K0 a/b/c + normalise) over `iters` launches
int Context::bench_decimate(const void* d_raw, size_t bytes_per_seg, int nseg, float* dI, float* dQ, int iters,
                            double* ms) {
    Impl& c = *d;
    const hipStream_t st = front_end_stream();          // the stream (and CU share) decimate_device() uses
    const size_t nblocks = bytes_per_seg / 2 / 6401;
    if (nblocks == 0) return -1;
    int32_t* scratch = static_cast<int32_t*>(c.decscratch.need((size_t)nseg * nblocks * 24));
    int* d_nv = static_cast<int*>(c.nvalid.need((size_t)nseg * 4));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, st));
    for (int it = 0; it < iters; ++it) {
        if (!dI) {                                            // read calibration: K0's access pattern, no arithmetic
            launch_calib_read(static_cast<const uint8_t*>(d_raw), bytes_per_seg, nseg, reinterpret_cast<unsigned*>(d_nv), st);
            continue;
        }
        launch_decimate(static_cast<const uint8_t*>(d_raw), bytes_per_seg, nseg, dI, dQ, d_nv, scratch, st);
        launch_normalise(dI, dQ, d_nv, nseg, kMaxSamples, st);
    }
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipEventSynchronize(e1));
    float t = 0;
    HIP_OK(hipEventElapsedTime(&t, e0, e1));
    ms[0] = t / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 1;
}
#endif  // WSPR_LAB

// Device Fano search over n host vectors: interleaved soft symbols in, results out.  The wave-parallel
// kernel reports -2 for a vector whose pending-visit store overflowed (not seen in tests; the serial host
// decoder takes those).
int Context::fano_batch(const unsigned char* symbols, int n, unsigned maxcycles, int* ret, unsigned* cycles,
                        unsigned* metric, unsigned* maxnp, unsigned char* data, unsigned* steps) {
    Impl& c = *d;
    if (n <= 0) return 0;
    int* h_off = static_cast<int*>(c.h_misc.need((size_t)n * 4));
    for (int i = 0; i < n; ++i) h_off[i] = i;
    unsigned char* dsym = static_cast<unsigned char*>(c.fz_sym.need((size_t)n * kNSymD));
    int* doff = static_cast<int*>(c.fz_off.need((size_t)n * 4));
    int* dret = static_cast<int*>(c.fz_ret.need((size_t)n * 4));
    unsigned* dcyc = static_cast<unsigned*>(c.fz_cyc.need((size_t)n * 4));
    unsigned* dmet = static_cast<unsigned*>(c.fz_met.need((size_t)n * 4));
    unsigned* dmax = static_cast<unsigned*>(c.fz_max.need((size_t)n * 4));
    unsigned char* ddat = static_cast<unsigned char*>(c.fz_dat.need((size_t)n * 10));
    unsigned* dsteps = steps ? static_cast<unsigned*>(c.fz_steps.need((size_t)n * 4)) : nullptr;
    upload(dsym, symbols, (size_t)n * kNSymD, c.stream);
    upload(doff, h_off, (size_t)n * 4, c.stream);
    launch_fano_wave(dsym, doff, n, c.t_metric0.as<short>(), maxcycles, dret, dcyc, dmet, dmax, ddat, dsteps,
                     static_cast<uint32_t*>(c.fz_pool.need(fano_wave_scratch_words(n) * 4)), c.stream);
    char* hz = static_cast<char*>(c.h_fz.need((size_t)n * 30));
    HIP_OK(hipMemcpyAsync(hz, dret, (size_t)n * 4, hipMemcpyDeviceToHost, c.stream));
    HIP_OK(hipMemcpyAsync(hz + (size_t)n * 4, dcyc, (size_t)n * 4, hipMemcpyDeviceToHost, c.stream));
    HIP_OK(hipMemcpyAsync(hz + (size_t)n * 8, dmet, (size_t)n * 4, hipMemcpyDeviceToHost, c.stream));
    HIP_OK(hipMemcpyAsync(hz + (size_t)n * 12, dmax, (size_t)n * 4, hipMemcpyDeviceToHost, c.stream));
    HIP_OK(hipMemcpyAsync(hz + (size_t)n * 16, ddat, (size_t)n * 10, hipMemcpyDeviceToHost, c.stream));
    if (steps) HIP_OK(hipMemcpyAsync(hz + (size_t)n * 26, dsteps, (size_t)n * 4, hipMemcpyDeviceToHost, c.stream));
    sync();
    memcpy(ret, hz, (size_t)n * 4);
    memcpy(cycles, hz + (size_t)n * 4, (size_t)n * 4);
    memcpy(metric, hz + (size_t)n * 8, (size_t)n * 4);
    memcpy(maxnp, hz + (size_t)n * 12, (size_t)n * 4);
    memcpy(data, hz + (size_t)n * 16, (size_t)n * 10);
    if (steps) memcpy(steps, hz + (size_t)n * 26, (size_t)n * 4);
    {
        std::vector<int> redo;
        for (int i = 0; i < n; ++i) if (ret[i] == -2) redo.push_back(i);
        if (!redo.empty()) {
            const FanoMetrics& met = default_metrics();
            c.bigpool->run((int)redo.size(), [&](int k) {
                const int i = redo[k];
                unsigned char sym[kNSymD], out11[11] = {0};
                memcpy(sym, symbols + (size_t)i * kNSymD, kNSymD);
                deinterleave162(sym);
                ret[i] = fano_decode(&metric[i], &cycles[i], &maxnp[i], out11, sym, kNBits, met.tab, 60, maxcycles);
                memcpy(data + (size_t)i * 10, out11, 10);
            }, 1);
        }
    }
    return 0;
}

// Device Fano search over vectors that are already in HBM: attempt i is the 162 soft symbols at
// d_symbols + h_offsets[i] * 162 (transmission order).  Results on the host; a vector whose pending-visit
// store overflowed (-2) is fetched and decoded by the serial host routine.
int Context::fano_resident(const unsigned char* d_symbols, const int* h_offsets, int n, unsigned maxcycles, int* ret,
                           unsigned* cycles, unsigned char* data) {
    Impl& c = *d;
    if (n <= 0) return 0;
    int* h_off = static_cast<int*>(c.h_misc.need((size_t)n * 4));
    memcpy(h_off, h_offsets, (size_t)n * 4);
    int* doff = static_cast<int*>(c.fz_off.need((size_t)n * 4));
    int* dret = static_cast<int*>(c.fz_ret.need((size_t)n * 4));
    unsigned* dcyc = static_cast<unsigned*>(c.fz_cyc.need((size_t)n * 4));
    unsigned char* ddat = static_cast<unsigned char*>(c.fz_dat.need((size_t)n * 10));
    upload(doff, h_off, (size_t)n * 4, c.stream);
    launch_fano_wave(d_symbols, doff, n, c.t_metric0.as<short>(), maxcycles, dret, dcyc, nullptr, nullptr, ddat, nullptr,
                     static_cast<uint32_t*>(c.fz_pool.need(fano_wave_scratch_words(n) * 4)), c.stream);
    char* hz = static_cast<char*>(c.h_fz.need((size_t)n * 18));
    HIP_OK(hipMemcpyAsync(hz, dret, (size_t)n * 4, hipMemcpyDeviceToHost, c.stream));
    HIP_OK(hipMemcpyAsync(hz + (size_t)n * 4, dcyc, (size_t)n * 4, hipMemcpyDeviceToHost, c.stream));
    HIP_OK(hipMemcpyAsync(hz + (size_t)n * 8, ddat, (size_t)n * 10, hipMemcpyDeviceToHost, c.stream));
    sync();
    memcpy(ret, hz, (size_t)n * 4);
    memcpy(cycles, hz + (size_t)n * 4, (size_t)n * 4);
    memcpy(data, hz + (size_t)n * 8, (size_t)n * 10);
    const FanoMetrics& met = default_metrics();
    for (int i = 0; i < n; ++i) {
        if (ret[i] != -2) continue;
        unsigned char sym[kNSymD], out11[11] = {0};
        HIP_OK(hipMemcpy(sym, d_symbols + (size_t)h_offsets[i] * kNSymD, kNSymD, hipMemcpyDeviceToHost));
        deinterleave162(sym);
        unsigned metric, maxnp;
        ret[i] = fano_decode(&metric, &cycles[i], &maxnp, out11, sym, kNBits, met.tab, 60, maxcycles);
        memcpy(data + (size_t)i * 10, out11, 10);
    }
    return 0;
}

}  // namespace wspr

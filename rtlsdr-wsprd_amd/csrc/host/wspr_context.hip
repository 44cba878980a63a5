// Contexts of the MI355X WSPR decoder: one per (device, lane, slot) -- HIP streams, constant tables, grow-only device
// and pinned buffers, host pools -- and everything that is not the decode itself: how input reaches the working buffers
// (host rows pageable or pinned, resident rows), the front end's entry points, the single-call stages behind the
// reference's sync_and_demodulate() / subtract_signal*().  The decode is wspr_pipeline.hip.
#include "wspr_context_impl.h"
#include "../kernels/glibc_sincosf.h"

namespace wspr {

std::atomic<int>& pool_workers_alive() { static std::atomic<int> n{0}; return n; }

std::atomic<int>& node_share() {
    static std::atomic<int> v{1};
    return v;
}

// The device's sinf / cosf restate the FMA3 build of glibc's routine, which x86-64 glibc selects on every CPU that has
// FMA3 -- any host an MI355X sits in.  On a host whose libm is the SSE2 build the reference itself would compute 34 of
// the 2^32 inputs differently (one ulp; all of them |x| > 17, i.e. phases of the subtraction's reference signal): the
// host-side constant tables would follow that libm, the kernels would not.  Checked once, on six of the 34 inputs
// (found by an exhaustive scan of both builds); a mismatch is reported loudly instead of being left to a parity test.
static void check_host_libm_once() {
    static const bool done = [] {
        static const uint32_t probe[6] = {0x418a3adbu, 0x41bc76d9u, 0x4202eb4bu, 0x4255b0a9u, 0x42a35c07u, 0x42cf5854u};
        int bad = 0;
        for (uint32_t b : probe) {
            float x;
            memcpy(&x, &b, 4);
            volatile float hx = x;                              // keep the calls out of constant folding
            const float hs = sinf(hx), hc = cosf(hx);
            const float ds = glibc_sinf(x), dc = glibc_cosf(x);
            bad += (memcmp(&hs, &ds, 4) != 0) + (memcmp(&hc, &dc, 4) != 0);
        }
        if (bad)
            fprintf(stderr, "libwspr_mi355x: WARNING: this host's libm computes sinf/cosf with its non-FMA build (%d of 12 probe "
                            "values differ): the kernels reproduce the FMA build; rebuild with -DWSPR_SINCOS_FMA=0 for "
                            "bit-exact agreement with a reference running on this host (34 of 2^32 inputs are affected)\n", bad);
        return true;
    }();
    (void)done;
}


Context::Context(int nslots) : d(new Impl) {
    check_host_libm_once();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        throw std::runtime_error("libwspr_mi355x: no HIP device visible (the HIP path is mandatory; there is no CPU fallback)");
    HIP_OK(hipGetDevice(&d->device));
    HIP_OK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    // Waiting host threads sleep on blocking events instead of spinning: never slower here (184 k vs
    // 180 k segments/s with 16 CPUs, 129 k vs 123 k with 2) and it leaves the CPUs to the Fano pools and
    // to other ranks.  WSPR_BLOCKING_SYNC=0 restores spinning.
    d->blocking = wait_mode() != 0;
    const unsigned evflags = wait_mode() == 1 ? hipEventBlockingSync : hipEventDefault;
    HIP_OK(hipEventCreateWithFlags(&d->ev[0], evflags));
    HIP_OK(hipEventCreateWithFlags(&d->ev[1], evflags));
    HIP_OK(hipEventCreateWithFlags(&d->ev_sync, evflags | hipEventDisableTiming));
    for (auto& pr : d->ev_def) { HIP_OK(hipEventCreate(&pr[0])); HIP_OK(hipEventCreate(&pr[1])); }
    for (auto& e : d->ev_stage) HIP_OK(hipEventCreateWithFlags(&e, evflags | hipEventDisableTiming));
    {
        int least = 0, greatest = 0;
        HIP_OK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_OK(hipStreamCreateWithPriority(&d->copy_stream, hipStreamNonBlocking, greatest));
        HIP_OK(hipEventCreateWithFlags(&d->ev_copy, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&d->ev_final, (wait_mode() == 1 ? hipEventBlockingSync : hipEventDefault) | hipEventDisableTiming));
    }

    // constant tables, computed with the host libm exactly as the reference does
    std::vector<float> window(kFftSize), lpf(kLpfTaps), part(kLpfTaps);
    for (int j = 0; j < kFftSize; ++j) window[j] = sinf(0.006147931 * j);          // wsprd.c:509-513
    std::vector<float2> tw(256);
    for (int k = 0; k < 256; ++k) {
        const double a = 2.0 * M_PI * (double)k / 512.0;
        tw[k] = make_float2((float)cos(a), (float)(-sin(a)));
    }
    tw[0] = make_float2(1.0f, 0.0f);
    tw[128] = make_float2(0.0f, -1.0f);
    float norm = 0.0f;                                                               // wsprd.c:353-368
    for (int i = 0; i < kLpfTaps; ++i) { lpf[i] = sinf(M_PI * (float)i / (float)(kLpfTaps - 1)); norm = norm + lpf[i]; }
    for (int i = 0; i < kLpfTaps; ++i) lpf[i] = lpf[i] / norm;
    part[0] = 0.0f;
    for (int i = 1; i < kLpfTaps; ++i) part[i] = part[i - 1] + lpf[i];
    for (int idt = 0; idt < kMaxLags; ++idt) {                                       // wsprd.c:742-744
        int ii = (idt + 1) / 2;
        if (idt % 2 == 1) ii = -ii;
        d->jitter_ladder[idt] = 3 * ii;
    }
    upload(d->t_window.need(window.size() * 4), window.data(), window.size() * 4, d->stream);
    upload(d->t_twiddle.need(tw.size() * 8), tw.data(), tw.size() * 8, d->stream);
    upload(d->t_sync.need(kNSym), sync_vector(), kNSym, d->stream);
    upload(d->t_lpf.need(lpf.size() * 4), lpf.data(), lpf.size() * 4, d->stream);
    upload(d->t_part.need(part.size() * 4), part.data(), part.size() * 4, d->stream);
    upload(d->t_jitter.need(sizeof d->jitter_ladder), d->jitter_ladder, sizeof d->jitter_ladder, d->stream);
    {
        static short metric0[256];
        for (int i = 0; i < 256; ++i) metric0[i] = (short)default_metrics().tab[0][i];
        upload(d->t_metric0.need(sizeof metric0), metric0, sizeof metric0, d->stream);
    }
    HIP_OK(hipStreamSynchronize(d->stream));
    d->tab.window = d->t_window.as<float>();
    d->tab.twiddle = d->t_twiddle.as<float2>();
    d->tab.sync = d->t_sync.as<unsigned char>();
    d->tab.lpf = d->t_lpf.as<float>();
    d->tab.lpf_part = d->t_part.as<float>();
    d->tab.min_snr = powf(10.0, -8.0 / 10.0);                                        // wsprd.c:590
    d->tab.floor_snr = 0.1 * d->tab.min_snr;                                         // wsprd.c:595

    int nthreads = rank_cpus();
    nthreads = std::max(1, std::min(nthreads, 256) / std::max(1, nslots));   // the slots share the host's CPUs
    d->pool.reset(new Pool(std::min(nthreads, 16) - 1));   // short phases: more threads only add wake-up cost
    d->bigpool.reset(new Pool(nthreads - 1));
}

Context::~Context() {}

// Number of concurrent pipelines ("slots"): each owns a HIP stream, buffers and host pools and
// decodes its own share of a batch, so that one slot's host phases (Fano, bookkeeping, copies)
// overlap the other slots' kernels.
// Three slots need about three CPUs for their driver threads (kernel launches are the host's main
// cost); with fewer, extra slots only take each other's time slices (2 CPUs: 2 slots 155 k, 3 slots
// 128 k segments/s on config 2).
int Context::slots() {
    static const int n = [] {
        int v = std::min(3, host_cpus());
        if (const char* e = getenv("WSPR_SLOTS")) v = atoi(e);
        return std::max(1, std::min(v, 8));
    }();
    return n;
}

// Lanes: independent sets of slot contexts.  A host thread is bound to one lane (default 0); calls
// made from threads bound to different lanes share nothing but the device and may overlap, which
// lets a service pipeline batch k+1 under the tail of batch k.
static thread_local int t_lane = 0;
static thread_local int t_slot_cap = 8;
int Context::lane() { return t_lane; }
void Context::cap_slots(int n) { t_slot_cap = std::max(1, n); }
int Context::slot_cap() { return std::min(slots(), std::min(t_slot_cap, rank_cpus())); }
void Context::bind_lane(int lane) { t_lane = std::max(0, std::min(lane, kMaxLanes - 1)); }

// Contexts are kept per (device, lane, slot): a host thread decodes on the HIP device that is current for it
// (hipSetDevice / wspr_set_device), so one process can drive every GPU of a node, one thread (or more) each.
static std::mutex g_ctx_mutex;
static std::unique_ptr<Context> g_ctx[Context::kMaxDevices][Context::kMaxLanes][8];

Context& Context::slot(int i) {
    std::mutex& m = g_ctx_mutex;
    auto& ctx = g_ctx;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        throw std::runtime_error("libwspr_mi355x: no HIP device visible (the HIP path is mandatory; there is no CPU fallback)");
    if (dev < 0 || dev >= kMaxDevices) throw std::runtime_error("libwspr_mi355x: device index out of range");
    std::lock_guard<std::mutex> g(m);
    std::unique_ptr<Context>& p = ctx[dev][t_lane][i];
    if (!p) p.reset(new Context(slots()));
    return *p;
}

Context& Context::get() { return slot(0); }

Context* Context::slot_if_exists(int i) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices || i < 0 || i >= 8) return nullptr;
    std::lock_guard<std::mutex> g(g_ctx_mutex);
    return g_ctx[dev][t_lane][i].get();
}
static thread_local int t_slots_used = 1;
void Context::note_slots_used(int n) { t_slots_used = std::max(1, n); }
int Context::last_slots_used() { return t_slots_used; }

// Work buffers (device and pinned) of every context of the current device go back to the driver; the constant
// tables, streams and host pools stay, the next call allocates what it needs.  No call may be in flight.
size_t Context::release_buffers() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
    std::lock_guard<std::mutex> g(g_ctx_mutex);
    size_t freed = 0;
    for (int lane = 0; lane < kMaxLanes; ++lane)
        for (int i = 0; i < 8; ++i) {
            if (!g_ctx[dev][lane][i]) continue;
            Impl& c = *g_ctx[dev][lane][i]->d;
            HIP_OK(hipStreamSynchronize(c.stream));
            if (c.fe_stream) HIP_OK(hipStreamSynchronize(c.fe_stream));
            for (DevBuf* b : {&c.iqI, &c.iqQ, &c.ps, &c.cand, &c.npk, &c.noise, &c.smspec, &c.seglist, &c.items, &c.syncbuf,
                              &c.symbuf, &c.rmsbuf, &c.jobs, &c.subscratch, &c.nvalid, &c.decscratch, &c.tabs, &c.pw, &c.pwfreq,
                              &c.lists, &c.scrsync, &c.psavg, &c.densein, &c.fz_sym, &c.fz_off, &c.fz_ret, &c.fz_cyc, &c.fz_met, &c.fz_max,
                              &c.fz_dat, &c.fz_steps, &c.fz_pool, &c.streamraw, &c.streamstate})
                freed += b->release();
            for (PinBuf* b : {&c.h_npk, &c.h_cand, &c.h_items, &c.h_sync, &c.h_sym, &c.h_rms, &c.h_jobs, &c.h_jobs2, &c.h_seglist,
                              &c.h_misc, &c.h_lists, &c.h_fz, &c.h_stage[0], &c.h_stage[1], &c.h_streamraw, &c.h_streamstate, &c.h_streamout})
                b->release();
            c.stage_samples[0] = c.stage_samples[1] = 0;
            free(c.hash_arena);
            c.hash_arena = nullptr;
            c.hash_arena_segs = 0;
        }
    return freed;
}

int Context::device() { return d->device; }

hipStream_t Context::stream() { return d->stream; }
const DeviceTables& Context::tables() { return d->tab; }
int Context::host_threads() { return d->bigpool->size(); }

float* Context::work_i(int nseg) { return static_cast<float*>(d->iqI.need((size_t)nseg * kIqStride * 4)); }
float* Context::work_q(int nseg) { return static_cast<float*>(d->iqQ.need((size_t)nseg * kIqStride * 4)); }

// rows are kIqStride floats; everything past `samples` must read as zero (the FFT bank
// of the reference reads up to 512*floor(samples/512)+255, wsprd.c:536-542)
static void zero_tail(float* wi, float* wq, int nseg, int samples, hipStream_t st) {
    const size_t tail = (size_t)(kIqStride - samples) * 4;
    HIP_OK(hipMemset2DAsync(wi + samples, (size_t)kIqStride * 4, 0, tail, nseg, st));
    HIP_OK(hipMemset2DAsync(wq + samples, (size_t)kIqStride * 4, 0, tail, nseg, st));
}

// Is this host address pinned (hipHostMalloc / hipHostRegister / wspr_pin_host_buffer)?  Pageable memory is "not
// registered" (an error on older runtimes: cleared).
static bool host_is_pinned(const void* p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// One turnstile per device for the host-buffer loads: calls in flight on several lanes (and the slots of one call) take
// the PCIe link one after the other instead of sharing it, so the first of them has its data -- and starts computing
// under the others' transfers -- after 1/n of the time.  A turn ends when the holder's last copy has COMPLETED (a host
// wait).  Two ways of closing the gap that leaves between two lanes' transfers were measured and dropped
// (profiles/r05_host_entry_probe_history.txt): ending the turn one chunk early (two lanes then share the link for a
// moment: +3 % in a process of its own, -20 % inside bench.py's), and handing over on the device (the next lane queues
// behind the holder's last copy with a stream wait: -3 %, and the same -20 % for pageable rows).
static std::mutex& host_load_turn(int device) {
    static std::mutex m[Context::kMaxDevices];
    return m[std::max(0, std::min(device, Context::kMaxDevices - 1))];
}

// The gathers of pageable rows are memcpy-bound (about 9 GB/s per thread here): one pool per process for them, half
// the CPUs of the rank's share but at most sixteen threads (4 / 8 / 12 / 16: 0.80 / 0.86-0.87 / 0.89-0.91 / 0.91 of the link on
// configs[1], profiles/r05_host_entry_probe_history.txt), used by one gather at a time.
static Pool& gather_pool(std::unique_lock<std::mutex>& hold) {
    static std::mutex m;
    static Pool pool(std::max(1, std::min(16, rank_cpus() / 2)) - 1);
    hold = std::unique_lock<std::mutex>(m);
    return pool;
}

// A row into a pinned chunk with streaming stores: the chunk is written once and read by the DMA engine, never by this
// CPU again, so it need not pass through (and evict from) the caches, and a line need not be read before it is written.
// dst is 32-byte aligned (rows of kIqStride floats in a page-aligned chunk), src is wherever the caller's row starts.
static inline void copy_row_streaming(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    typedef float v8u __attribute__((vector_size(32), aligned(4)));
    typedef float v8 __attribute__((vector_size(32)));
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const v8u x = *reinterpret_cast<const v8u*>(src + i);
        __builtin_nontemporal_store(static_cast<v8>(x), reinterpret_cast<v8*>(dst + i));
    }
    for (; i < n; ++i) dst[i] = src[i];
}

// The reference's callers hand wspr_decode() HOST buffers (rtlsdr_wsprd.c:316, :689).  Pinned caller memory goes to
// the device as LINEAR asynchronous copies of up to 256 rows (DMA at the link's rate, no host work) into a dense
// buffer that a row kernel on a stream of its own spreads into the working layout under the next copy.  Pageable
// caller memory would make the runtime stage it through its own small bounce buffers, synchronously; instead the rows
// are gathered (host pool) into this context's two pinned chunks, already in the working layout, and each chunk leaves
// as ONE contiguous asynchronous copy per rail while the pool fills the other chunk.  A call of fewer than sixteen
// segments (a receiver's own record) does not queue at the link's turnstile (pageable: the runtime's own strided copy).
void Context::load_host(const float* I, const float* Q, int nseg, int samples, size_t stride) {
    Impl& c = *d;
    float* wi = work_i(nseg);
    float* wq = work_q(nseg);
    if (nseg <= 0) return;
    // The transfers run on a stream of the HIGHEST priority and the decode stream waits for its last event.  Measured
    // (tools/dma_interference.py): beside twelve lanes of decoder kernels a linear pinned-to-device copy on an ordinary
    // stream gets 10-18 GB/s of the link's 55 -- its queue's packets wait their turn behind kernels -- and 35 GB/s on a
    // high-priority stream, with the decoder's step unchanged either way (the DMA engines do the work).
    const hipStream_t ld = c.copy_stream;
    struct Join {                                                    // whatever path is taken: the decode stream follows the load
        Impl& c;
        ~Join() {
            (void)hipEventRecord(c.ev_copy, c.copy_stream);
            (void)hipStreamWaitEvent(c.stream, c.ev_copy, 0);
        }
    } join{c};
    // One load at a time per device (host_load_turn()).  A call of a few segments does not queue up (a receiver's
    // single record must not wait for a batch's 370 MB).
    std::unique_lock<std::mutex> turn(host_load_turn(c.device), std::defer_lock);
    auto take_turn = [&] { turn.lock(); };
    auto pass_turn = [&] {                                       // the turn ends when the LINK is free again, not when
        HIP_OK(hipEventRecord(c.ev_final, ld));                  // the copies have merely been queued
        host_wait(c.ev_final, 30000L);                           // short naps: the link idles for as long as this thread oversleeps
    };
    const bool queues_up = nseg >= 16;
    if (host_is_pinned(I) && host_is_pinned(Q)) {
        if (queues_up) take_turn();
        // Pinned rows: LINEAR copies (the DMA engines at the link's rate, no CU involved) of up to kDense rows at a time
        // into a dense device buffer, and the row kernel that also serves resident input spreads them into the working
        // layout (device to device, microseconds).  A strided host-to-device copy straight into the working rows measured
        // 40-45 GB/s against 55 for the linear one (round 5).
        // (rows far apart -- a stride of more than twice the record -- would make a linear copy carry the gaps: those
        // take the strided copy below)
        if ((samples & 3) == 0 && (stride & 3) == 0 && stride <= 2 * (size_t)samples &&
            !(reinterpret_cast<uintptr_t>(I) & 15) && !(reinterpret_cast<uintptr_t>(Q) & 15)) {
            constexpr int kDense = 256;
            const int per = std::min(nseg, kDense);
            // Two dense buffers in turn, the row kernels on a stream of their own (same priority): the kernel of chunk k
            // runs under the DMA of chunk k + 1 -- on the copy stream itself it would stand between two copies, and
            // beside a busy GPU that is 0.1-0.2 ms of idle link per chunk.
            if (!c.row_stream) {
                int least = 0, greatest = 0;
                HIP_OK(hipDeviceGetStreamPriorityRange(&least, &greatest));
                HIP_OK(hipStreamCreateWithPriority(&c.row_stream, hipStreamNonBlocking, greatest));
                for (int b = 0; b < 2; ++b) {
                    HIP_OK(hipEventCreateWithFlags(&c.ev_dense[b], hipEventDisableTiming));
                    HIP_OK(hipEventCreateWithFlags(&c.ev_rows[b], hipEventDisableTiming));
                }
            }
            float* dn0 = static_cast<float*>(c.densein.need((size_t)4 * per * stride * 4));
            const int nchunks = (nseg + per - 1) / per;
            for (int c0 = 0, k = 0; c0 < nseg; c0 += per, ++k) {
                const int n = std::min(per, nseg - c0), b = k & 1;
                float* dn = dn0 + (size_t)b * 2 * per * stride;
                const size_t fl = (size_t)(n - 1) * stride + samples;           // the last row may end at `samples`
                if (k >= 2) HIP_OK(hipStreamWaitEvent(ld, c.ev_rows[b], 0));    // the kernel that read this buffer last
                HIP_OK(hipMemcpyAsync(dn, I + (size_t)c0 * stride, fl * 4, hipMemcpyHostToDevice, ld));
                HIP_OK(hipMemcpyAsync(dn + (size_t)per * stride, Q + (size_t)c0 * stride, fl * 4, hipMemcpyHostToDevice, ld));
                HIP_OK(hipEventRecord(c.ev_dense[b], ld));
                HIP_OK(hipStreamWaitEvent(c.row_stream, c.ev_dense[b], 0));
                if (!launch_load_rows(dn, dn + (size_t)per * stride, stride, samples, n, wi + (size_t)c0 * kIqStride,
                                      wq + (size_t)c0 * kIqStride, c.row_stream))
                    throw std::runtime_error("load_rows refused an aligned dense chunk");
                HIP_OK(hipEventRecord(c.ev_rows[b], c.row_stream));
            }
            if (queues_up) pass_turn();
            for (int b = 0; b < std::min(2, nchunks); ++b) HIP_OK(hipStreamWaitEvent(ld, c.ev_rows[b], 0));   // the load ends with its last row kernel
        } else {
            zero_tail(wi, wq, nseg, samples, ld);
            HIP_OK(hipMemcpy2DAsync(wi, (size_t)kIqStride * 4, I, stride * 4, (size_t)samples * 4, nseg, hipMemcpyHostToDevice, ld));
            HIP_OK(hipMemcpy2DAsync(wq, (size_t)kIqStride * 4, Q, stride * 4, (size_t)samples * 4, nseg, hipMemcpyHostToDevice, ld));
            if (queues_up) pass_turn();
        }
        return;
    }
    if (nseg < 16) {                                             // a single call's record or a handful: the runtime's own path
        zero_tail(wi, wq, nseg, samples, ld);
        HIP_OK(hipMemcpy2DAsync(wi, (size_t)kIqStride * 4, I, stride * 4, (size_t)samples * 4, nseg, hipMemcpyHostToDevice, ld));
        HIP_OK(hipMemcpy2DAsync(wq, (size_t)kIqStride * 4, Q, stride * 4, (size_t)samples * 4, nseg, hipMemcpyHostToDevice, ld));
        return;
    }
    // Pageable rows: gathered by the host pool into this context's two pinned chunks, already in the working layout
    // (rows of kIqStride floats, zero tail), each chunk then ONE linear copy per rail.  The first two chunks are gathered
    // BEFORE the turn is taken (the link belongs to another lane meanwhile), the others under the DMA of their
    // predecessors.
    constexpr int chunk = 64;                                    // segments per chunk: 11.5 MB per rail, two rails, two chunks
    const size_t row = (size_t)kIqStride, rail = (size_t)chunk * row;      // floats; the layout of a chunk never changes
    auto gather = [&](int k) {
        const int c0 = k * chunk, n = std::min(chunk, nseg - c0), b = k & 1;
        const bool fresh = c.h_stage[b].cap < 2 * rail * 4;
        float* st = static_cast<float*>(c.h_stage[b].need(2 * rail * 4));
        if (fresh) { memset(st, 0, 2 * rail * 4); c.stage_samples[b] = 0; }
        else if (k >= 2) host_wait(c.ev_stage[b]);                           // the DMA that read this chunk two turns ago
        const int dirty = c.stage_samples[b];                    // a shorter record than the last one leaves old samples behind
        auto fill = [&](int r) {
            float* di = st + (size_t)r * row;
            float* dq = st + rail + (size_t)r * row;
            copy_row_streaming(di, I + (size_t)(c0 + r) * stride, (size_t)samples);
            copy_row_streaming(dq, Q + (size_t)(c0 + r) * stride, (size_t)samples);
            if (dirty > samples) {
                memset(di + samples, 0, (size_t)(dirty - samples) * 4);
                memset(dq + samples, 0, (size_t)(dirty - samples) * 4);
            }
            __builtin_ia32_sfence();                              // the streaming stores are visible before the DMA is queued
        };
        if (n >= 8) {
            std::unique_lock<std::mutex> hold;
            gather_pool(hold).run(n, fill, 2);
        } else {
            for (int r = 0; r < n; ++r) fill(r);
        }
        // every row of the chunk now ends at `samples` (rows beyond n: whatever they held, never sent)
        c.stage_samples[b] = (n == chunk) ? samples : std::max(dirty, samples);
    };
    auto send = [&](int k) {
        const int c0 = k * chunk, n = std::min(chunk, nseg - c0), b = k & 1;
        const float* st = c.h_stage[b].as<float>();
        HIP_OK(hipMemcpyAsync(wi + (size_t)c0 * row, st, (size_t)n * row * 4, hipMemcpyHostToDevice, ld));
        HIP_OK(hipMemcpyAsync(wq + (size_t)c0 * row, st + rail, (size_t)n * row * 4, hipMemcpyHostToDevice, ld));
        HIP_OK(hipEventRecord(c.ev_stage[b], ld));
    };
    const int nchunks = (nseg + chunk - 1) / chunk;
    gather(0);
    if (nchunks > 1) gather(1);
    take_turn();
    send(0);
    if (nchunks > 1) send(1);
    for (int k = 2; k < nchunks; ++k) { gather(k); send(k); }
    pass_turn();
    // the caller's rows were consumed by the gathers and may change from here on
}
void Context::load_device(const void* dI, const void* dQ, int nseg, int samples, size_t stride) {
    float* wi = work_i(nseg);
    float* wq = work_q(nseg);
    if (launch_load_rows(static_cast<const float*>(dI), static_cast<const float*>(dQ), stride, samples, nseg, wi, wq, d->stream))
        return;
    zero_tail(wi, wq, nseg, samples, d->stream);
    HIP_OK(hipMemcpy2DAsync(wi, (size_t)kIqStride * 4, dI, stride * 4, (size_t)samples * 4, nseg, hipMemcpyDeviceToDevice, d->stream));
    HIP_OK(hipMemcpy2DAsync(wq, (size_t)kIqStride * 4, dQ, stride * 4, (size_t)samples * 4, nseg, hipMemcpyDeviceToDevice, d->stream));
}
void Context::store_host(float* I, float* Q, int nseg, int samples, size_t stride) {
    HIP_OK(hipMemcpy2DAsync(I, stride * 4, d->iqI.p, (size_t)kIqStride * 4, (size_t)samples * 4, nseg, hipMemcpyDeviceToHost, d->stream));
    HIP_OK(hipMemcpy2DAsync(Q, stride * 4, d->iqQ.p, (size_t)kIqStride * 4, (size_t)samples * 4, nseg, hipMemcpyDeviceToHost, d->stream));
    sync();
}
void Context::sync() {
    HIP_OK(hipGetLastError());
    if (d->blocking) {
        HIP_OK(hipEventRecord(d->ev_sync, d->stream));
        host_wait(d->ev_sync);
    } else {
        HIP_OK(hipStreamSynchronize(d->stream));
    }
    d->resolve_deferred();
}

float* Context::ps_buffer(int nseg) {
    return static_cast<float*>(d->ps.need((size_t)nseg * kPsBins * kPsTPitch * 4));
}

// restore the original IQ of single segments (rows) of the working buffers
void Context::reload_rows(const float* I, const float* Q, bool device, size_t stride, int samples,
                          const std::vector<int>& segs) {
    Impl& c = *d;
    const hipMemcpyKind kind = device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    // the rows must still be there: a revisit (WSPR_HASH_REVISIT) comes back to working buffers a previous call filled, and
    // wspr_release_buffers() or a smaller share in between would leave nothing, or too little, to write into
    for (int s : segs)
        if (s < 0 || !c.iqI.p || !c.iqQ.p || ((size_t)s + 1) * kIqStride * 4 > c.iqI.cap || ((size_t)s + 1) * kIqStride * 4 > c.iqQ.cap)
            throw std::runtime_error("the working rows of the previous call are gone (buffers released or resized since)");
    for (int s : segs) {
        HIP_OK(hipMemcpyAsync(c.iqI.as<float>() + (size_t)s * kIqStride, I + (size_t)s * stride, (size_t)samples * 4, kind, c.stream));
        HIP_OK(hipMemcpyAsync(c.iqQ.as<float>() + (size_t)s * kIqStride, Q + (size_t)s * stride, (size_t)samples * 4, kind, c.stream));
    }
    if (!device) sync();       // pageable host memory: the copies must not outlive the caller's view
}

// ------------------------------------------------------- single-call stages --
void Context::demod_single(float* id, float* qd, long np, unsigned char* symbols, float* freq, int ifmin,
                           int ifmax, float fstep, int* shift, int lagmin, int lagmax, int lagstep,
                           float* drift, float* sync, int mode, int symfac) {
    Impl& c = *d;
    const int samples = (int)std::min<long>(np, kMaxSamples);
    load_host(id, qd, 1, samples, (size_t)samples);
    FineState f{};
    f.seg = 0; f.freq = *freq; f.drift = *drift; f.shift = *shift; f.sync = 1e30f;
    f.freq_coarse = *freq; f.shift_coarse = lagmin + 128;
    FineState* d_items = static_cast<FineState*>(c.items.need(sizeof(FineState)));
    upload(d_items, &f, sizeof f, c.stream);
    const float* wi = c.iqI.as<float>();
    const float* wq = c.iqQ.as<float>();
    if (mode == 0) {
        const int nl = (lagmax - lagmin) / lagstep + 1;
        float* d_sync = static_cast<float*>(c.syncbuf.need((size_t)nl * 4));
        launch_demod(wi, wq, (int)np, d_items, 1, 0, nl, lagstep, 0, 0.0f, nullptr, 0.0f, d_sync, nullptr, nullptr, c.tab, c.stream);
        launch_pick_lag(d_items, 1, d_sync, nl, lagstep, c.stream);
    } else if (mode == 1) {
        const int nf = ifmax - ifmin + 1;
        float* d_sync = static_cast<float*>(c.syncbuf.need((size_t)nf * 4));
        launch_demod(wi, wq, (int)np, d_items, 1, 1, nf, lagstep, ifmin, fstep, nullptr, 0.0f, d_sync, nullptr, nullptr, c.tab, c.stream);
        launch_pick_freq(d_items, 1, d_sync, nf, ifmin, fstep, c.stream);
    } else {
        float* d_sync = static_cast<float*>(c.syncbuf.need(4));
        unsigned char* d_sym = static_cast<unsigned char*>(c.symbuf.need(kNSymD));
        float* d_rms = static_cast<float*>(c.rmsbuf.need(4));
        launch_demod(wi, wq, (int)np, d_items, 1, 2, 1, lagstep, 0, 0.0f, c.t_jitter.as<int>(), -INFINITY, d_sync, d_sym, d_rms, c.tab, c.stream, symfac);
        float s2 = 0;
        HIP_OK(hipMemcpyAsync(&s2, d_sync, 4, hipMemcpyDeviceToHost, c.stream));
        HIP_OK(hipMemcpyAsync(symbols, d_sym, kNSymD, hipMemcpyDeviceToHost, c.stream));
        HIP_OK(hipStreamSynchronize(c.stream));
        *sync = s2;
        return;
    }
    HIP_OK(hipMemcpyAsync(&f, d_items, sizeof f, hipMemcpyDeviceToHost, c.stream));
    HIP_OK(hipStreamSynchronize(c.stream));
    *sync = f.sync; *shift = f.shift; *freq = f.freq;
}

void Context::subtract_single(float* id, float* qd, long np, float f0, int shift, float drift,
                              const unsigned char* sym) {
    Impl& c = *d;
    const int samples = (int)std::min<long>(np, kMaxSamples);
    load_host(id, qd, 1, samples, (size_t)samples);
    SubJob jb{};
    jb.seg = 0; jb.f0 = f0; jb.shift = shift; jb.drift = drift;
    memcpy(jb.sym, sym, kNSymD);
    SubJob* dj = static_cast<SubJob*>(c.jobs.need(sizeof jb));
    upload(dj, &jb, sizeof jb, c.stream);
    float* scratch = static_cast<float*>(c.subscratch.need(subtract_scratch_floats(1) * 4));
    launch_subtract(c.iqI.as<float>(), c.iqQ.as<float>(), (int)np, dj, 1, scratch, c.tab, c.stream);
    store_host(id, qd, 1, samples, (size_t)samples);
}

void Context::subtract_symbolwise_single(float* id, float* qd, long np, float f0, int shift, float drift,
                                         const unsigned char* sym) {
    Impl& c = *d;
    const int samples = (int)std::min<long>(np, kMaxSamples);
    load_host(id, qd, 1, samples, (size_t)samples);
    unsigned char* d_sym = static_cast<unsigned char*>(c.symbuf.need(kNSymD));
    upload(d_sym, sym, kNSymD, c.stream);
    launch_subtract_symbolwise(c.iqI.as<float>(), c.iqQ.as<float>(), samples, f0, shift, drift, d_sym, c.stream);
    store_host(id, qd, 1, samples, (size_t)samples);
}

// CUs the front end may occupy (0 = all).  K0 is HBM-bound and launches hundreds of thousands of short workgroups:
// on an unmasked stream they take every CU as it frees up and the decoder's fp32-bound kernels of the other lanes
// wait.  Confined to a share of the CUs (a stream created with hipExtStreamCreateWithCUMask; consecutive mask bits
// fall on different XCDs, so the share is spread over all eight and keeps every HBM stack busy), K0 still finds
// the memory bandwidth it needs while the rest of the chip keeps computing.
std::atomic<int>& front_end_cus() {
    static std::atomic<int> v{[] { const char* e = lab_env("WSPR_K0_CUS"); return e ? atoi(e) : 0; }()};
    return v;
}

hipStream_t Context::front_end_stream() {
    Impl& c = *d;
    const int want = front_end_cus().load();
    if (want != c.fe_cus) {
        if (c.fe_stream) { (void)hipStreamSynchronize(c.fe_stream); (void)hipStreamDestroy(c.fe_stream); c.fe_stream = nullptr; }
        c.fe_cus = want;
        if (want > 0) {
            hipDeviceProp_t prop;
            HIP_OK(hipGetDeviceProperties(&prop, c.device));
            const int ncu = prop.multiProcessorCount;
            const int n = std::max(8, std::min(want, ncu));
            std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
            for (int i = 0; i < n; ++i) mask[i >> 5] |= 1u << (i & 31);
            HIP_OK(hipExtStreamCreateWithCUMask(&c.fe_stream, (uint32_t)mask.size(), mask.data()));
        }
    }
    return c.fe_stream ? c.fe_stream : c.stream;
}

int Context::decimate_device(const void* d_raw, size_t bytes_per_seg, int nseg, float* dI, float* dQ, int normalise,
                             int* h_nout, DecimState* d_states) {
    Impl& c = *d;
    const hipStream_t st = d_states ? c.stream : front_end_stream();     // whole segments only: streaming chunks are small
    const size_t nblocks = (size_t)decimate_blocks(bytes_per_seg / 2, d_states != nullptr);
    if (nblocks == 0) return -1;
    int32_t* scratch = static_cast<int32_t*>(c.decscratch.need((size_t)nseg * nblocks * 24));
    int* d_nv = static_cast<int*>(c.nvalid.need((size_t)nseg * 4));
    if (!d_states) {                                          // whole segments: the unfilled tail must read as zero
        HIP_OK(hipMemsetAsync(dI, 0, (size_t)nseg * kIqStride * 4, st));
        HIP_OK(hipMemsetAsync(dQ, 0, (size_t)nseg * kIqStride * 4, st));
    }
    launch_decimate(static_cast<const uint8_t*>(d_raw), bytes_per_seg, nseg, dI, dQ, d_nv, scratch, st, d_states);
    if (normalise) launch_normalise(dI, dQ, d_nv, nseg, kMaxSamples, st);
    if (h_nout) HIP_OK(hipMemcpyAsync(h_nout, d_nv, (size_t)nseg * 4, hipMemcpyDeviceToHost, st));
    if (c.blocking) {
        HIP_OK(hipEventRecord(c.ev_sync, st));
        host_wait(c.ev_sync);
    } else {
        HIP_OK(hipStreamSynchronize(st));
    }
    return 0;
}

// one chunk of one receiver's stream: state in, appended outputs and state out (host buffers)
int Context::decimate_stream(DecimState* h_state, const uint8_t* iq, size_t nbytes, float* I, float* Q, uint32_t fill,
                             uint32_t cap, uint32_t* new_fill) {
    Impl& c = *d;
    if (nbytes == 0) { if (new_fill) *new_fill = fill; return 0; }
    uint8_t* d_raw = static_cast<uint8_t*>(c.streamraw.need(nbytes + 16));
    DecimState* d_st = static_cast<DecimState*>(c.streamstate.need(sizeof(DecimState)));
    upload(d_raw, iq, nbytes, c.stream);
    upload(d_st, h_state, sizeof(DecimState), c.stream);
    float* wi = work_i(1);
    float* wq = work_q(1);
    int nout = 0;
    const int rc = decimate_device(d_raw, nbytes, 1, wi, wq, 0, &nout, d_st);
    if (rc) return rc;
    HIP_OK(hipMemcpy(h_state, d_st, sizeof(DecimState), hipMemcpyDeviceToHost));
    const uint32_t room = fill < cap ? cap - fill : 0u;
    const uint32_t take = std::min<uint32_t>((uint32_t)nout, room);       // outputs beyond the capacity are dropped
    if (take) {                                                            // (rtlsdr_wsprd.c:236-242)
        HIP_OK(hipMemcpy(I + fill, wi, (size_t)take * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(Q + fill, wq, (size_t)take * 4, hipMemcpyDeviceToHost));
    }
    if (new_fill) *new_fill = fill + take;
    return 0;
}

// One chunk of each of n receivers' streams: what n calls of decimate_stream() do, as ONE host-to-device transfer, one
// launch set over n rows with n carried states, and three transfers back (states, the rows' first outputs per rail).
// A callback of 65 536 bytes yields five or six samples: its cost is the round trips, not the arithmetic.
int Context::decimate_stream_many(DecimState* const* h_states, const uint8_t* const* iq, size_t nbytes, int n, float* const* I,
                                  float* const* Q, const uint32_t* fill, uint32_t cap, uint32_t* new_fill) {
    Impl& c = *d;
    if (n <= 0 || nbytes == 0) { for (int k = 0; k < n; ++k) new_fill[k] = fill[k]; return 0; }
    const int maxout = (int)(nbytes / 2 / 6401) + 2;                       // outputs a chunk of this size can yield
    const size_t ocols = (size_t)((maxout + 3) & ~3);
    uint8_t* h_raw = static_cast<uint8_t*>(c.h_streamraw.need((size_t)n * nbytes));
    DecimState* h_st = static_cast<DecimState*>(c.h_streamstate.need((size_t)n * sizeof(DecimState) + (size_t)n * 4));
    int* h_nout = reinterpret_cast<int*>(h_st + n);
    float* h_out = static_cast<float*>(c.h_streamout.need(2 * (size_t)n * ocols * 4));
    for (int k = 0; k < n; ++k) {
        memcpy(h_raw + (size_t)k * nbytes, iq[k], nbytes);
        h_st[k] = *h_states[k];
    }
    uint8_t* d_raw = static_cast<uint8_t*>(c.streamraw.need((size_t)n * nbytes + 16));
    DecimState* d_st = static_cast<DecimState*>(c.streamstate.need((size_t)n * sizeof(DecimState)));
    HIP_OK(hipMemcpyAsync(d_raw, h_raw, (size_t)n * nbytes, hipMemcpyHostToDevice, c.stream));
    HIP_OK(hipMemcpyAsync(d_st, h_st, (size_t)n * sizeof(DecimState), hipMemcpyHostToDevice, c.stream));
    float* wi = work_i(n);
    float* wq = work_q(n);
    const size_t nblocks = (size_t)decimate_blocks(nbytes / 2, true);
    if (nblocks == 0) return -1;
    int32_t* scratch = static_cast<int32_t*>(c.decscratch.need((size_t)n * nblocks * 24));
    int* d_nv = static_cast<int*>(c.nvalid.need((size_t)n * 4));
    launch_decimate(d_raw, nbytes, n, wi, wq, d_nv, scratch, c.stream, d_st);
    HIP_OK(hipMemcpyAsync(h_nout, d_nv, (size_t)n * 4, hipMemcpyDeviceToHost, c.stream));
    HIP_OK(hipMemcpyAsync(h_st, d_st, (size_t)n * sizeof(DecimState), hipMemcpyDeviceToHost, c.stream));
    HIP_OK(hipMemcpy2DAsync(h_out, ocols * 4, wi, (size_t)kIqStride * 4, ocols * 4, n, hipMemcpyDeviceToHost, c.stream));
    HIP_OK(hipMemcpy2DAsync(h_out + (size_t)n * ocols, ocols * 4, wq, (size_t)kIqStride * 4, ocols * 4, n, hipMemcpyDeviceToHost, c.stream));
    sync();
    for (int k = 0; k < n; ++k) {
        *h_states[k] = h_st[k];
        const uint32_t room = fill[k] < cap ? cap - fill[k] : 0u;
        const uint32_t take = std::min<uint32_t>((uint32_t)std::max(0, std::min(h_nout[k], maxout)), room);   // beyond the capacity: dropped
        if (take) {                                                                          // (rtlsdr_wsprd.c:236-242)
            memcpy(I[k] + fill[k], h_out + (size_t)k * ocols, (size_t)take * 4);
            memcpy(Q[k] + fill[k], h_out + (size_t)n * ocols + (size_t)k * ocols, (size_t)take * 4);
        }
        new_fill[k] = fill[k] + take;
    }
    return 0;
}

}  // namespace wspr

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
#include "wspr_pipeline.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "wspr_message.h"
#include "../kernels/glibc_sincosf.h"

namespace wspr {

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

#define HIP_OK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) +   \
                                     " at " #expr);                                          \
    } while (0)

// ---------------------------------------------------------------- host waits --
// How a host thread waits for its stream.  Measured in round 5 (tools/shard_cpu_profile.py): hipEventSynchronize() --
// on events created with hipEventBlockingSync as well -- kept the waiting thread on a CPU for the whole wait in this
// runtime (twelve lanes in flight = twelve CPUs busy doing nothing; a rank with two CPUs was host-bound at 77 % of the
// GPU's rate on a single-signal batch).  The default is therefore a wait that costs no CPU: poll the event for a few
// tens of microseconds (a small batch's kernels are done by then: single-call latency is unchanged), then sleep
// between polls, with the sleep growing to a quarter of a millisecond.  WSPR_BLOCKING_SYNC=0: the runtime's spinning
// wait; =1: the runtime's wait on blocking events (rounds 2-4); unset or =2: poll and sleep.
static int wait_mode() {
    static const int m = [] { const char* e = getenv("WSPR_BLOCKING_SYNC"); return e ? atoi(e) : 2; }();
    return m;
}
// how long a wait polls before it starts sleeping: a single call's kernels finish within tens to hundreds of
// microseconds and its latency is what its caller sees (one wspr_decode() per two minutes), a large batch's take
// milliseconds and its lanes' CPUs are what the other lanes and ranks need
static thread_local int t_spin_us = 40;
static void host_wait(hipEvent_t ev) {
    if (wait_mode() != 2) {
        const hipError_t e = hipEventSynchronize(ev);
        if (e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e) + " at hipEventSynchronize");
        return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    constexpr long nap_cap_ns = 250000L;        // 60 / 120 / 250 / 500 / 1000 us measured alike (profiles/r05_sleep_cap_ab.txt)
    long nap_ns = 20000;
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e) + " at hipEventQuery");
        if (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(t_spin_us)) { __builtin_ia32_pause(); continue; }
        timespec ts{0, nap_ns};
        nanosleep(&ts, nullptr);
        nap_ns = std::min(nap_ns * 2, nap_cap_ns);
    }
}

// ------------------------------------------------------------------ buffers --
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    void* need(size_t bytes) {
        if (bytes > cap) {
            if (p) HIP_OK(hipFree(p));
            p = nullptr;
            size_t want = bytes + bytes / 4;
            HIP_OK(hipMalloc(&p, want));
            cap = want;
        }
        return p;
    }
    size_t release() {
        const size_t had = cap;
        if (p) HIP_OK(hipFree(p));
        p = nullptr;
        cap = 0;
        return had;
    }
    template <class T> T* as() { return static_cast<T*>(p); }
};
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    void* need(size_t bytes) {
        if (bytes > cap) {
            if (p) HIP_OK(hipHostFree(p));
            p = nullptr;
            size_t want = bytes + bytes / 4;
            HIP_OK(hipHostMalloc(&p, want, hipHostMallocDefault));
            cap = want;
        }
        return p;
    }
    void release() {
        if (p) HIP_OK(hipHostFree(p));
        p = nullptr;
        cap = 0;
    }
    template <class T> T* as() { return static_cast<T*>(p); }
};

// ------------------------------------------------------------- thread pool --
// Fork-join pool for the host phases between kernel launches (Fano attempts,
// per-segment bookkeeping).  A job is an immutable heap object with two counters;
// completion is "all tasks done", never "all workers checked in", so threads that
// wake up late cost nothing, and a late thread holding an exhausted old job can never
// touch a newer one.  Idle workers sleep on a condition variable; only the caller spins,
// briefly, for the last tasks to finish.
// worker threads of all host pools alive in this process (the calling threads of the pools are not counted)
std::atomic<int>& pool_workers_alive() { static std::atomic<int> n{0}; return n; }

class Pool {
    struct Job {
        const std::function<void(int)>* fn;
        int total, chunk;
        std::atomic<int> next{0}, done{0};
    };

public:
    explicit Pool(int n) {
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); });
        pool_workers_alive().fetch_add((int)workers_.size());
    }
    ~Pool() {
        pool_workers_alive().fetch_sub((int)workers_.size());
        quit_.store(true);
        { std::lock_guard<std::mutex> g(m_); ++epoch_; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    // runs fn(i) for i in [0, n); the calling thread participates.
    // chunk = indices handed out per grab; 0 = automatic (many cheap, uniform tasks)
    void run(int n, const std::function<void(int)>& fn, int chunk = 0) {
        if (n <= 0) return;
        if (workers_.empty() || n < 4) { for (int i = 0; i < n; ++i) fn(i); return; }
        auto job = std::make_shared<Job>();
        job->fn = &fn;
        job->total = n;
        job->chunk = chunk > 0 ? chunk : std::max(1, n / (8 * ((int)workers_.size() + 1)));
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = job;
            ++epoch_;
        }
        cv_.notify_all();
        drain(*job);
        while (job->done.load(std::memory_order_acquire) < n) cpu_relax();
    }
    int size() const { return (int)workers_.size() + 1; }

private:
    static void cpu_relax() { __builtin_ia32_pause(); }
    static void drain(Job& j) {
        for (;;) {
            const int lo = j.next.fetch_add(j.chunk);
            if (lo >= j.total) break;
            const int hi = std::min(j.total, lo + j.chunk);
            for (int i = lo; i < hi; ++i) (*j.fn)(i);
            j.done.fetch_add(hi - lo, std::memory_order_release);
        }
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return epoch_ != seen; });
                seen = epoch_;
                if (quit_.load()) return;
                job = job_;
            }
            if (job) drain(*job);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_;
    std::shared_ptr<Job> job_;
    unsigned long epoch_ = 0;
    std::atomic<bool> quit_{false};
};

// ---------------------------------------------------------------- context ----
namespace {
struct SegBook {                 // host bookkeeping of one segment across passes
    int   uniques = 0;
    float allfreqs[100];
    char  allcalls[100][13];
    std::vector<int> dirty;      // hash slots written (cleared when the batch ends)
    std::vector<decoder_results> spots;   // every unique spot, in decode order (the reference's 100 at most)
};
}  // namespace

struct Context::Impl {
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr; // host-buffer loads, at the highest stream priority (see load_host)
    hipEvent_t ev_copy = nullptr;
    hipStream_t fe_stream = nullptr;   // front end (K0) on a CU-masked stream, see front_end_cus()
    int fe_cus = 0;                    // CUs the mask of fe_stream admits (0: fe_stream not in use)
    int device = 0;
    DeviceTables tab{};
    DevBuf t_window, t_twiddle, t_sync, t_lpf, t_part, t_jitter, t_metric0;
    DevBuf iqI, iqQ, ps, cand, npk, noise, smspec, seglist, items, syncbuf, symbuf, rmsbuf, jobs, subscratch,
        nvalid, decscratch, tabs, pw, pwfreq, lists, scrsync, psavg, densein, fz_sym, fz_off, fz_ret, fz_cyc, fz_met, fz_max, fz_dat, fz_steps, fz_pool, streamraw, streamstate;
    PinBuf h_npk, h_cand, h_items, h_sync, h_sym, h_rms, h_jobs, h_jobs2, h_seglist, h_misc, h_lists;
    // host-buffer entry (wspr_decode_batch: the reference's calling convention, wsprd.h:106-111): pageable caller rows
    // are gathered into two pinned chunks in the working layout (rows of kIqStride floats, zero tail) that take turns,
    // so that the host's gather of chunk k+1 runs under the DMA of chunk k and every DMA is one contiguous copy
    PinBuf h_fz;                     // K6w's results on their way to the host (a copy into pageable memory would make
                                     // the runtime wait for the search itself, on a CPU)
    PinBuf h_stage[2];
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    int stage_samples[2] = {0, 0};   // columns [samples, kIqStride) of a chunk are zero from here on
    int sub_flip = 0;
    bool dev_fano = false;           // this batch: Fano attempts on the device (see fano_device_mode())
    bool crowded = false;            // the previous batch had more than one Fano time-out per ten segments
    int cand_head = 16;              // candidates per segment copied to the host (adapts to the lists seen)
    // host mirrors that keep their storage between calls: value-initialising 8 192 x 200 candidate slots (46 MB) and
    // 8 192 segment books (14 MB) per call was a fifth of the host's CPU time per step on a single-signal batch
    std::vector<DevCand> cand_host;
    std::vector<int> npk_host;
    std::vector<SegBook> books;
    std::unique_ptr<Pool> pool;      // <= 32 threads: the short phases (first-rung Fano, bookkeeping)
    std::unique_ptr<Pool> bigpool;   // every host thread we may use: the long Fano ladders of weak candidates
    int jitter_ladder[kMaxLags];
    // host-side per-segment callsign hash memory (reference: locals of wspr_decode)
    char* hash_arena = nullptr;
    size_t hash_arena_segs = 0;
    double t_ms[24] = {0};           // stage times (ms), Fano statistics and host CPU time by phase of the last batch
    std::atomic<long> n_fano{0}, n_timeout{0}, n_cycles{0}, n_kept{0}, n_subjobs{0};
    bool blocking = false;
    hipEvent_t ev_sync = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    // spans timed without a host wait: event pairs recorded around the launches, read back after a later
    // synchronisation of the same (in-order) stream has passed them
    static constexpr int kDeferred = 8;
    hipEvent_t ev_def[kDeferred][2] = {};
    double* def_acc[kDeferred] = {};
    int n_def = 0;
    void resolve_deferred() {
        for (int i = 0; i < n_def; ++i) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev_def[i][0], ev_def[i][1]) == hipSuccess) *def_acc[i] += ms;
        }
        n_def = 0;
    }
};

// CPUs this process may actually use: hardware threads capped by the cgroup CPU quota
// (a container on a shared GPU node typically owns a slice; running more runnable threads
// than the quota gets the whole process throttled)
static int usable_cpus();
// CPUs this process may count on: WSPR_HOST_THREADS (a rank's share when several ranks share a host),
// else the cgroup quota / affinity mask
static int host_cpus() {
    static const int n = [] {
        int v = usable_cpus();
        if (const char* e = getenv("WSPR_HOST_THREADS")) v = atoi(e);
        return std::max(1, v);
    }();
    return n;
}
// Shards of one node-level call that share this host's CPUs (wspr_decode_batch_node: one per device): contexts
// created from then on size their pools for a 1/n share, as a rank of an N-rank job does via WSPR_HOST_THREADS.
std::atomic<int>& node_share() {
    static std::atomic<int> v{1};
    return v;
}
static int rank_cpus() { return std::max(1, host_cpus() / std::max(1, node_share().load())); }
static int usable_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 1;
    long quota = -1, period = -1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                     // cgroup v2
        char q[32] = {0};
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atol(q);
        fclose(f);
    } else {
        if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(f1, "%ld", &quota) != 1) quota = -1; fclose(f1); }
        if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f2, "%ld", &period) != 1) period = -1; fclose(f2); }
    }
    if (quota > 0 && period > 0) n = std::min(n, (int)std::max(1L, (quota + period - 1) / period));
    return n;
}

static void upload(void* dst, const void* src, size_t bytes, hipStream_t st) {
    HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
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
                              &c.h_misc, &c.h_lists, &c.h_fz, &c.h_stage[0], &c.h_stage[1]})
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
// under the others' transfers -- after 1/n of the time.
static std::mutex& host_load_turn(int device) {
    static std::mutex m[Context::kMaxDevices];
    return m[std::max(0, std::min(device, Context::kMaxDevices - 1))];
}

// The gathers of pageable rows are memcpy-bound (about 9 GB/s per thread here): one pool per process for them, half
// the CPUs of the rank's share but at most eight threads, used by one gather at a time.
static Pool& gather_pool(std::unique_lock<std::mutex>& hold) {
    static std::mutex m;
    static Pool pool(std::max(1, std::min(8, rank_cpus() / 2)) - 1);
    hold = std::unique_lock<std::mutex>(m);
    return pool;
}

// The reference's callers hand wspr_decode() HOST buffers (rtlsdr_wsprd.c:316, :689).  Pinned caller memory goes to
// the device as one asynchronous strided copy per rail (DMA at the link's rate, no host work).  Pageable caller memory
// would make the runtime stage it through its own small bounce buffers, synchronously; instead the rows are gathered
// (host pool) into this context's two pinned chunks, already in the working layout, and each chunk leaves as ONE
// contiguous asynchronous copy per rail while the pool fills the other chunk.
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
    // One load at a time per device (the turnstile): the lane whose turn it is has the link to itself and its batch in
    // HBM after 1/n of the time n concurrent loads would take -- and starts computing under the next lane's transfer.
    std::unique_lock<std::mutex> turn(host_load_turn(c.device), std::defer_lock);
    auto finish_turn = [&] {                                     // the turn ends when the LINK is free again, not when
        HIP_OK(hipEventRecord(c.ev_copy, ld));                   // the copies have merely been queued
        host_wait(c.ev_copy);
    };
    if (host_is_pinned(I) && host_is_pinned(Q)) {
        turn.lock();
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
            // two dense buffers in turn: the row kernel of chunk k runs under the DMA of chunk k + 1
            float* dn0 = static_cast<float*>(c.densein.need((size_t)4 * per * stride * 4));
            for (int c0 = 0, k = 0; c0 < nseg; c0 += per, ++k) {
                const int n = std::min(per, nseg - c0);
                float* dn = dn0 + (size_t)(k & 1) * 2 * per * stride;
                const size_t fl = (size_t)(n - 1) * stride + samples;           // the last row may end at `samples`
                HIP_OK(hipMemcpyAsync(dn, I + (size_t)c0 * stride, fl * 4, hipMemcpyHostToDevice, ld));
                HIP_OK(hipMemcpyAsync(dn + (size_t)per * stride, Q + (size_t)c0 * stride, fl * 4, hipMemcpyHostToDevice, ld));
                if (!launch_load_rows(dn, dn + (size_t)per * stride, stride, samples, n, wi + (size_t)c0 * kIqStride,
                                      wq + (size_t)c0 * kIqStride, ld))
                    throw std::runtime_error("load_rows refused an aligned dense chunk");
            }
        } else {
            zero_tail(wi, wq, nseg, samples, ld);
            HIP_OK(hipMemcpy2DAsync(wi, (size_t)kIqStride * 4, I, stride * 4, (size_t)samples * 4, nseg, hipMemcpyHostToDevice, ld));
            HIP_OK(hipMemcpy2DAsync(wq, (size_t)kIqStride * 4, Q, stride * 4, (size_t)samples * 4, nseg, hipMemcpyHostToDevice, ld));
        }
        if (nseg >= 16) finish_turn();
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
            memcpy(di, I + (size_t)(c0 + r) * stride, (size_t)samples * 4);
            memcpy(dq, Q + (size_t)(c0 + r) * stride, (size_t)samples * 4);
            if (dirty > samples) {
                memset(di + samples, 0, (size_t)(dirty - samples) * 4);
                memset(dq + samples, 0, (size_t)(dirty - samples) * 4);
            }
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
    turn.lock();
    send(0);
    if (nchunks > 1) send(1);
    for (int k = 2; k < nchunks; ++k) { gather(k); send(k); }
    finish_turn();
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
    c.n_fano = 0; c.n_timeout = 0; c.n_cycles = 0; c.n_kept = 0; c.n_subjobs = 0;
    const auto t_all0 = std::chrono::steady_clock::now();
    CpuSpan cpu_all(&c.t_ms[16]);
    t_spin_us = nseg <= 16 ? 600 : 40;
    for (int s = 0; s < nseg; ++s) n_results[s] = 0;
    const int blocks = 4 * (samples / kFftSize) - 1;
    if (nseg <= 0) return 0;
    if (samples > kMaxSamples || blocks < 23) return 0;      // outside what the reference arrays allow

    const unsigned fast_cfg = fano_fast_budget().load();
    // small batches gain nothing from the split (their time-outs fit the host pool) and would pay
    // the device kernel's latency
    const bool dev_fano = fano_device_mode() > 0 ||
                          (fano_device_mode() < 0 && nseg >= 256 && (rank_cpus() < 4 || d->crowded));
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
        if (c.hash_arena_segs < (size_t)nseg) {
            free(c.hash_arena);
            c.hash_arena = static_cast<char*>(calloc((size_t)nseg, per_seg));
            if (!c.hash_arena) throw std::runtime_error("out of host memory for hash tables");
            c.hash_arena_segs = (size_t)nseg;
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
        FlatHashTable flat(hashtab_of(s), loctab_of(s), &bk.dirty);
        std::unique_ptr<SegHashView> shared(hb ? new SegHashView(hb, hb_off + s) : nullptr);
        HashTable& tab = hb ? static_cast<HashTable&>(*shared) : static_cast<HashTable&>(flat);
        // (what the bits unpack and re-encode to is computed once per host thread: MessageCache, wspr_hashmem.h)
        MessageCache& mc = MessageCache::of_this_thread();
        MessageCache::Handle mh = mc.unpack(w.decdata, tab, call_loc_pow, call, loc, pwr, callsign);
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
        snprintf(bk.allcalls[bk.uniques], sizeof bk.allcalls[0], "%s", callsign);
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
            snprintf(o->message, sizeof o->message, "%s", call_loc_pow);
            snprintf(o->call, sizeof o->call, "%s", call);
            snprintf(o->loc, sizeof o->loc, "%s", loc);
            snprintf(o->pwr, sizeof o->pwr, "%s", pwr);
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
    for (int s : segs) {
        for (int slot : book[s].dirty) {
            memset(hashtab_of(s) + (size_t)slot * kHashWidth, 0, kHashWidth);
            memset(loctab_of(s) + (size_t)slot * kLocWidth, 0, kLocWidth);
        }
        book[s].dirty.clear();
    }
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
    for (int s : active0) { SegBook& b = run.book[s]; b.uniques = 0; b.dirty.clear(); b.spots.clear(); }
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
    const int n = std::min(cap, 24);
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
    for (int s = 0; s < nseg; ++s) {
        if (npk[s] <= 0) continue;
        const DevCand& cd = cand[(size_t)s * kMaxCand];
        FineState f{};
        f.seg = s; f.freq = cd.freq; f.drift = 0.0f; f.shift = cd.shift; f.sync = cd.sync;
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

// restore the original IQ of single segments (rows) of the working buffers
void Context::reload_rows(const float* I, const float* Q, bool device, size_t stride, int samples,
                          const std::vector<int>& segs) {
    Impl& c = *d;
    const hipMemcpyKind kind = device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
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

}  // namespace wspr

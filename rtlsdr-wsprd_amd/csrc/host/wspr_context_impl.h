// Internals shared by the two host translation units of the scheduler (wspr_context.hip: contexts, lanes, buffers,
// loads, front end, single-call stages; wspr_pipeline.hip: the decode itself): how a host thread waits, grow-only
// buffers, the fork-join pool, the context's state, the CPUs a rank may count on.  Not part of any interface.
#pragma once
#include "wspr_pipeline.h"

#include <algorithm>
#include <atomic>
#include <chrono>
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

namespace wspr {

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
inline int wait_mode() {
    static const int m = [] { const char* e = getenv("WSPR_BLOCKING_SYNC"); return e ? atoi(e) : 2; }();
    return m;
}
// how long a wait polls before it starts sleeping: a single call's kernels finish within tens to hundreds of
// microseconds and its latency is what its caller sees (one wspr_decode() per two minutes), a large batch's take
// milliseconds and its lanes' CPUs are what the other lanes and ranks need
inline thread_local int t_spin_us = 40;
inline void host_wait(hipEvent_t ev, long nap_cap_ns = 250000L) {  // 60 / 120 / 250 / 500 / 1000 us measured alike (profiles/r05_sleep_cap_ab.txt)
    if (wait_mode() != 2) {
        const hipError_t e = hipEventSynchronize(ev);
        if (e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e) + " at hipEventSynchronize");
        return;
    }
    const auto t0 = std::chrono::steady_clock::now();
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
            cap = 0;                                   // an allocation that fails leaves an EMPTY buffer behind, not a stale size
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
            cap = 0;
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
// The callsign hash memory of ONE segment without usehashtable: the reference's locals hashtab[32768][13] / loctab[32768][5]
// (wsprd.c:478-479), zeroed for every call, of which a segment's decodes touch a handful of slots -- kept as the list of
// slots written instead of 590 KB per segment (8 192 segments: 4.8 GB of arena per context, and four cache misses per
// decode into it; round 6).  Same answers: a look-up returns the last call stored at the slot, "" if none; the locator
// table is never read without the option (it only goes to hashtable.txt), so it is not kept.
struct SparseHashTable : wspr::HashTable {
    struct Entry { int slot; char call[wspr::kHashWidth]; };
    std::vector<Entry> entries;
    const char* find(int slot) const {
        for (size_t i = entries.size(); i-- > 0;) if (entries[i].slot == slot) return entries[i].call;
        return "";
    }
    const char* call_at(int slot) override { return find(slot); }
    const char* peek(int slot) override { return find(slot); }
    void put(int slot, const char* call, const char*) override {
        for (Entry& e : entries) if (e.slot == slot) { wspr::copy_text(e.call, sizeof e.call, call); return; }
        Entry e;
        e.slot = slot;
        wspr::copy_text(e.call, sizeof e.call, call);
        entries.push_back(e);
    }
};

struct SegBook {                 // host bookkeeping of one segment across passes
    int   uniques = 0;
    float allfreqs[100];
    char  allcalls[100][13];
    SparseHashTable hash;        // the segment's hash memory (cleared when the batch ends); usehashtable: flat arena / HashBatch
    std::vector<int> dirty;      // hash slots written in the flat arena (single call with usehashtable)
    std::vector<decoder_results> spots;   // every unique spot, in decode order (the reference's 100 at most)
};

struct Context::Impl {
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr; // host-buffer loads, at the highest stream priority (see load_host)
    hipEvent_t ev_copy = nullptr;
    hipEvent_t ev_final = nullptr;     // this context's last host-to-device copy of a load: the turn at the link ends with it
    hipStream_t row_stream = nullptr;  // pinned host rows: the kernel that spreads a dense chunk, beside the next chunk's DMA
    hipEvent_t ev_dense[2] = {nullptr, nullptr}, ev_rows[2] = {nullptr, nullptr};
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
    PinBuf h_streamraw, h_streamstate, h_streamout;   // many receivers' callbacks at once (decimate_stream_many)
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
    double t_ms[26] = {0};           // stage times (ms), Fano statistics and host CPU time by phase of the last batch
    std::atomic<long> n_fano{0}, n_timeout{0}, n_cycles{0}, n_kept{0}, n_subjobs{0}, n_mc_lookups{0}, n_mc_hits{0};
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
inline int usable_cpus();
// CPUs this process may count on: WSPR_HOST_THREADS (a rank's share when several ranks share a host),
// else the cgroup quota / affinity mask
inline int host_cpus() {
    static const int n = [] {
        int v = usable_cpus();
        if (const char* e = getenv("WSPR_HOST_THREADS")) v = atoi(e);
        return std::max(1, v);
    }();
    return n;
}
// Shards of one node-level call that share this host's CPUs (wspr_decode_batch_node: one per device): contexts
// created from then on size their pools for a 1/n share, as a rank of an N-rank job does via WSPR_HOST_THREADS.
inline int rank_cpus() { return std::max(1, host_cpus() / std::max(1, node_share().load())); }
inline int usable_cpus() {
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

inline void upload(void* dst, const void* src, size_t bytes, hipStream_t st) {
    HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
}


}  // namespace wspr

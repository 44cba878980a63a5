// C ABI of libwspr_mi355x.so (declared in include/wspr_mi355x.h).
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <atomic>
#include <condition_variable>
#include <ctime>
#include <new>
#include <vector>

#include "wspr_message.h"
#include "wspr_metric_tables.h"
#include "wspr_pipeline.h"

using wspr::Context;

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr ": ") + hipGetErrorString(e_)); \
    } while (0)

namespace {
// The reference reports nothing but "zero spots" on failure (wsprd.c:854 returns 0
// always).  A missing GPU is a deployment error, not a weak-signal condition: say so
// loudly on stderr and return a negative code; there is no CPU fallback.
int fail(const char* where, const std::exception& e) {
    fprintf(stderr, "libwspr_mi355x: %s failed: %s\n", where, e.what());
    (void)hipGetLastError();          // the runtime's sticky error belongs to THIS call: the next one starts clean
    return -1;
}
#ifdef WSPR_LAB
// The callsign hash memory makes a segment's result depend on what was decoded before it (wsprd.c:481-494, 842-852:
// hashtable.txt read before, written after every decode).  The product decodes a usehashtable batch in parallel all the
// same (decode_hashed() below); the per-candidate TRACE of the lab library keeps the plain form of rounds 2-4: one
// segment after the other, each as the reference's own call would be (load the file, decode, save the file).
template <class One>
int decode_in_order(int nseg, int* n_results, One one) {
    int rc = 0;
    for (int s = 0; s < nseg; ++s) {
        const int r = one(s);
        if (r < 0) { rc = r; for (int k = s; k < nseg; ++k) n_results[k] = 0; break; }
    }
    return rc;
}
#endif
// One call at a time per (device, lane).  The library is not re-entrant within a lane (neither is the reference:
// global FFTW plan, static state and fixed file names, wsprd.c:81, :133); threads that never bound a lane all sit on
// lane 0, and until round 5 two of them calling at once shared a context -- streams, working buffers, pools -- silently.
// Now their calls take turns: slow instead of wrong.  Recursive, because entry points call each other on one thread
// (wspr_decode -> wspr_decode_batch -> wspr_decode_batch_hashed); the node-level calls do NOT take it (their worker
// threads bind the caller's lane on each device and call the batch entry points).
std::recursive_mutex& lane_turn_of(int dev, int lane) {
    static std::recursive_mutex turns[Context::kMaxDevices][Context::kMaxLanes];
    return turns[std::max(0, std::min(dev, Context::kMaxDevices - 1))][std::max(0, std::min(lane, Context::kMaxLanes - 1))];
}
int current_device_or_0() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return dev;
}
struct LaneTurn {                                  // the calling thread's lane of the current device
    std::unique_lock<std::recursive_mutex> hold;
    LaneTurn() : hold(lane_turn_of(current_device_or_0(), Context::lane())) {}
};
struct AllLanesTurn {                              // every lane of the current device, in index order (wspr_release_buffers)
    std::vector<std::unique_lock<std::recursive_mutex>> hold;
    AllLanesTurn() {
        const int dev = current_device_or_0();
        for (int lane = 0; lane < Context::kMaxLanes; ++lane) hold.emplace_back(lane_turn_of(dev, lane));
    }
};
// device scratch of one call, released on every exit path
struct TempDev {
    void* p = nullptr;
    explicit TempDev(size_t bytes) { HIP_TRY(hipMalloc(&p, bytes)); }
    ~TempDev() { if (p) (void)hipFree(p); }
    TempDev(const TempDev&) = delete;
    TempDev& operator=(const TempDev&) = delete;
    template <class T> T* as() { return static_cast<T*>(p); }
};
}  // namespace

namespace {
// Splits a batch over the pipelines (slots).  Segments are independent, so every slot decodes a
// contiguous share on its own stream while the others are in their host phases.
// hb (usehashtable on a batch): the shared, ordered hash memory; after the first round the segments whose look-ups no
// longer hold are decoded again, round by round, until none is left (see HashBatch in wspr_pipeline.h).
template <class Load, class Reload>
int decode_split(int nseg, int samples, const decoder_options& options, decoder_results* decodes, int max_results,
                 int* n_results, Load load, Reload reload, bool writeback, float* idat, float* qdat, size_t seg_stride,
                 wspr_trace* trace = nullptr, wspr::HashBatch* hb = nullptr, const std::vector<int>* revisit = nullptr,
                 const std::function<void()>& before_validation = nullptr) {
    const int nslots = (nseg >= 128) ? Context::slot_cap() : 1;
    Context::note_slots_used(nslots);
    Context& c0 = Context::get();
    const int dev = c0.device(), lane = Context::lane();
    struct Share { int lo, hi; };
    std::vector<Share> share(nslots);
    for (int g = 0; g < nslots; ++g) share[g] = {(int)((long)nseg * g / nslots), (int)((long)nseg * (g + 1) / nslots)};
    // runs fn(g, context of slot g) for every slot, slot 0 on the calling thread when it is the only one
    auto on_slots = [&](const std::function<void(int, Context&)>& fn) {
        if (nslots == 1) { fn(0, c0); return; }
        std::vector<std::thread> th;
        std::vector<std::string> errs(nslots);
        std::vector<char> bad(nslots, 0);
        for (int g = 0; g < nslots; ++g)
            th.emplace_back([&, g] {
                try {
                    if (hipSetDevice(dev) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
                    Context::bind_lane(lane);
                    fn(g, Context::slot(g));
                } catch (const std::exception& e) { bad[g] = 1; errs[g] = e.what(); }
            });
        for (auto& t : th) t.join();
        for (int g = 0; g < nslots; ++g)
            if (bad[g]) throw std::runtime_error(errs[g].empty() ? "slot failed" : errs[g]);
    };
    auto again = [&](const std::vector<int>& todo) {         // global (call-relative) indices, ascending
        on_slots([&](int g, Context& c) {
            const int lo = share[g].lo, hi = share[g].hi;
            std::vector<int> mine;
            for (int t : todo) if (t >= lo && t < hi) mine.push_back(t - lo);
            if (mine.empty()) return;
            reload(c, lo, mine);
            c.decode_again(hi - lo, samples, options, decodes + (size_t)lo * max_results, max_results, n_results + lo, mine, hb, lo);
        });
    };
    if (!revisit) {
        on_slots([&](int g, Context& c) {
            const int lo = share[g].lo, hi = share[g].hi;
            load(c, lo, hi - lo);
            const int rc = c.decode_resident(hi - lo, samples, options, decodes + (size_t)lo * max_results, max_results, n_results + lo,
                                             [&c, &reload, lo](const std::vector<int>& segs) { reload(c, lo, segs); },
                                             trace ? trace + lo : nullptr, hb, lo);
            if (rc < 0) throw std::runtime_error("decode failed");
        });
    } else if (!revisit->empty()) {
        // the batch of the previous call once more (its rows are still in the slots' working buffers, decoded): only
        // the listed segments are restored and decoded again
        ++hb->rounds; hb->redecoded += (int)revisit->size();
        again(*revisit);
    }
    if (hb && before_validation) before_validation();      // e.g. wait for the calls before this one, take their file as the base
    if (hb)
        for (;;) {
            hb->rebuild();
            const std::vector<int> todo = hb->invalid();
            if (todo.empty()) break;
            ++hb->rounds; hb->redecoded += (int)todo.size();
            again(todo);
        }
    if (writeback)
        on_slots([&](int g, Context& c) {
            const int lo = share[g].lo, hi = share[g].hi;
            c.store_host(idat + (size_t)lo * seg_stride, qdat + (size_t)lo * seg_stride, hi - lo, samples, seg_stride);
        });
    return 0;
}

// usehashtable: calls are ordered by definition -- each reads the file the previous one wrote.  The order is the order
// in which they ENTER (a ticket), and a call's turn comes when every earlier ticket has left.  A batch call decodes its
// first round BEFORE its turn (against the file as it is then: speculation, beside the calls ahead of it on other
// lanes), waits, takes the file its predecessors have written as its base, decodes again what that changes, writes the
// file and leaves; single calls and the sharded form simply wait for their turn first.
struct HashChain {
    std::mutex m;
    std::condition_variable cv;
    unsigned long next = 0, serving = 0;
    static HashChain& get() { static HashChain c; return c; }
    struct Ticket {
        HashChain& c;
        unsigned long t;
        bool left = false;
        explicit Ticket(HashChain& c_) : c(c_) { std::lock_guard<std::mutex> g(c.m); t = c.next++; }
        void wait_turn() { std::unique_lock<std::mutex> g(c.m); c.cv.wait(g, [&] { return c.serving == t; }); }
        void leave() {
            if (left) return;
            wait_turn();
            { std::lock_guard<std::mutex> g(c.m); ++c.serving; }
            left = true;
            c.cv.notify_all();
        }
        ~Ticket() { leave(); }                               // whatever happened: the calls behind must not wait for ever
        Ticket(const Ticket&) = delete;
        Ticket& operator=(const Ticket&) = delete;
    };
};
}  // namespace

namespace {
// RFC 3986 unreserved characters pass, everything else is %XX (what curl_easy_escape does)
std::string url_escape(const char* s) {
    static const char hex[] = "0123456789ABCDEF";
    std::string o;
    for (const unsigned char* p = reinterpret_cast<const unsigned char*>(s); *p; ++p) {
        if (isalnum(*p) || *p == '-' || *p == '.' || *p == '_' || *p == '~') o.push_back((char)*p);
        else { o.push_back('%'); o.push_back(hex[*p >> 4]); o.push_back(hex[*p & 15]); }
    }
    return o;
}
}  // namespace

namespace {
// usehashtable on a batch: parallel decode against the shared, ordered hash memory (HashBatch), to the fixed point.
// The memory of the calling thread's last such call is kept: WSPR_HASH_REVISIT decodes only what a new `prior` changes.
thread_local std::unique_ptr<wspr::HashBatch> t_hash;
static_assert(sizeof(wspr_hash_op) == sizeof(wspr::HashOp), "public and internal hash-op layouts differ");

template <class Load, class Reload>
int decode_hashed(int nseg, int samples, const decoder_options& options, decoder_results* decodes, int max_results,
                  int* n_results, Load load, Reload reload, bool writeback, float* idat, float* qdat, size_t seg_stride,
                  int seg_index0, const wspr_hash_op* prior, int n_prior, int flags, wspr_hash_op* stores_out, int cap,
                  int* n_stores, int* n_redecoded) {
    HashChain::Ticket ticket(HashChain::get());
    const bool revisit = (flags & WSPR_HASH_REVISIT) != 0;
    // a plain batch call (all of its job in one call, the file its own to write) may run its first round ahead of its
    // turn; a shard of a larger job is driven round by round from outside and waits first -- and keeps its turn for the
    // whole call, so shards driven from several threads of ONE process (several devices or lanes) decode one after the
    // other (include/wspr_mi355x.h says so; shards in different processes -- rtlsdr-wsprd_amd/dist.py -- do not meet here)
    const bool ahead = !revisit && !(flags & WSPR_HASH_KEEP_FILE) && n_prior <= 0;
    if (!ahead) ticket.wait_turn();
    // a revisit works on the state the previous call of this thread left: its log, and the decoded rows in the slots'
    // working buffers.  It is refused unless that call COMPLETED (a call that threw leaves a half-updated log) over the
    // same segments, samples and slot layout (wspr_set_thread_slots / a node-level share in between change the shares)
    const int nslots_now = (nseg >= 128) ? Context::slot_cap() : 1;
    if (revisit && !(t_hash && t_hash->valid && (int)t_hash->log.size() == nseg && t_hash->seg0 == seg_index0 &&
                     t_hash->samples == samples && t_hash->nslots == nslots_now))
        throw std::runtime_error("WSPR_HASH_REVISIT without a matching, completed previous call on this thread");
    if (!revisit) {
        t_hash.reset(new wspr::HashBatch);
        t_hash->load_file();
        t_hash->seg0 = seg_index0;
        t_hash->resize(nseg);
    }
    wspr::HashBatch& hb = *t_hash;
    hb.valid = false;                                      // until this call has run to its end
    hb.nslots = nslots_now; hb.samples = samples;
    hb.rounds = hb.redecoded = 0;
    hb.prior.assign(reinterpret_cast<const wspr::HashOp*>(prior), reinterpret_cast<const wspr::HashOp*>(prior) + std::max(0, n_prior));
    std::stable_sort(hb.prior.begin(), hb.prior.end(), [](const wspr::HashOp& a, const wspr::HashOp& b) { return a.seg < b.seg; });
    std::vector<int> todo;
    if (revisit) { hb.rebuild(); todo = hb.invalid(); }
    decode_split(nseg, samples, options, decodes, max_results, n_results, load, reload, writeback, idat, qdat, seg_stride,
                 nullptr, &hb, revisit ? &todo : nullptr,
                 ahead ? std::function<void()>([&] { ticket.wait_turn(); hb.load_file(); }) : std::function<void()>());
    hb.valid = true;
    const std::vector<wspr::HashOp> st = hb.stores();
    if (n_stores) *n_stores = (int)st.size();
    if (n_redecoded) *n_redecoded = hb.redecoded;
    // a store buffer that is too small fails the call BEFORE anything is committed: hashtable.txt is untouched, the
    // result arrays hold the decode, *n_stores the capacity needed, and the same call with WSPR_HASH_REVISIT (same
    // prior, a larger buffer) completes it without decoding anything again
    if (stores_out && (int)st.size() > cap) return -3;
    if (!(flags & WSPR_HASH_KEEP_FILE)) hb.commit_file();
    if (stores_out && !st.empty()) memcpy(stores_out, st.data(), st.size() * sizeof(wspr::HashOp));
    return 0;
}
}  // namespace

extern "C" {

const char* wspr_mi355x_version(void) { return "wspr-mi355x 0.3 (gfx950, HIP)"; }

int wspr_device_ready(void) {
    try { Context::get(); return 1; } catch (const std::exception& e) { fail("wspr_device_ready", e); return 0; }
}

size_t wspr_iq_stride(void) { return (size_t)wspr::kIqStride; }

int wspr_decode_batch(float* idat, float* qdat, int nseg, int samples, size_t seg_stride,
                      struct decoder_options options, struct decoder_results* decodes, int max_results,
                      int* n_results, int writeback) {
    LaneTurn lane_turn;
    if (options.usehashtable && nseg > 1)                 // the hash memory orders the segments: parallel all the same
        return wspr_decode_batch_hashed(idat, qdat, nseg, samples, seg_stride, options, decodes, max_results, n_results,
                                        writeback, 0, nullptr, 0, 0, nullptr, 0, nullptr, nullptr);
    try {
        // a single call with the option reads and writes hashtable.txt itself (wsprd.c:481-494, 842-852): in its turn
        std::unique_ptr<HashChain::Ticket> turn;
        if (options.usehashtable) { turn.reset(new HashChain::Ticket(HashChain::get())); turn->wait_turn(); }
        if (samples > wspr::kMaxSamples) {
            // the reference derives its block count from `samples` (wsprd.c:516) and would read past the 45 000 samples
            // its callers hold; this library's working rows are 45 000 samples, so a longer record is refused, not cut
            fprintf(stderr, "libwspr_mi355x: samples = %d exceeds the %d this library decodes\n", samples, wspr::kMaxSamples);
            for (int s = 0; s < nseg; ++s) n_results[s] = 0;
            return -2;
        }
        return decode_split(nseg, samples, options, decodes, max_results, n_results,
                            [&](Context& c, int lo, int n) {
                                c.load_host(idat + (size_t)lo * seg_stride, qdat + (size_t)lo * seg_stride, n, samples, seg_stride);
                            },
                            [&](Context& c, int lo, const std::vector<int>& segs) {
                                c.reload_rows(idat + (size_t)lo * seg_stride, qdat + (size_t)lo * seg_stride, false, seg_stride, samples, segs);
                            },
                            writeback != 0, idat, qdat, seg_stride);
    } catch (const std::exception& e) {
        for (int s = 0; s < nseg; ++s) n_results[s] = 0;
        return fail("wspr_decode_batch", e);
    }
}

int wspr_decode_batch_hashed(float* idat, float* qdat, int nseg, int samples, size_t seg_stride,
                             struct decoder_options options, struct decoder_results* decodes, int max_results,
                             int* n_results, int writeback, int seg_index0, const wspr_hash_op* prior, int n_prior,
                             int flags, wspr_hash_op* stores_out, int cap, int* n_stores, int* n_redecoded) {
    LaneTurn lane_turn;
    try {
        if (samples > wspr::kMaxSamples) {
            fprintf(stderr, "libwspr_mi355x: samples = %d exceeds the %d this library decodes\n", samples, wspr::kMaxSamples);
            for (int s = 0; s < nseg; ++s) n_results[s] = 0;
            return -2;
        }
        options.usehashtable = 1;
        return decode_hashed(nseg, samples, options, decodes, max_results, n_results,
                             [&](Context& c, int lo, int n) {
                                 c.load_host(idat + (size_t)lo * seg_stride, qdat + (size_t)lo * seg_stride, n, samples, seg_stride);
                             },
                             [&](Context& c, int lo, const std::vector<int>& segs) {
                                 c.reload_rows(idat + (size_t)lo * seg_stride, qdat + (size_t)lo * seg_stride, false, seg_stride, samples, segs);
                             },
                             writeback != 0, idat, qdat, seg_stride, seg_index0, prior, n_prior, flags, stores_out, cap,
                             n_stores, n_redecoded);
    } catch (const std::exception& e) {
        if (!(flags & WSPR_HASH_REVISIT)) for (int s = 0; s < nseg; ++s) n_results[s] = 0;
        return fail("wspr_decode_batch_hashed", e);
    }
}

int wspr_hash_commit(const wspr_hash_op* stores, int n) {
    try {
        HashChain::Ticket ticket(HashChain::get());
        ticket.wait_turn();
        wspr::HashBatch hb;
        hb.load_file();
        wspr::HashBatch::commit_file(hb.base_call, hb.base_grid, reinterpret_cast<const wspr::HashOp*>(stores), (size_t)std::max(0, n));
        return 0;
    } catch (const std::exception& e) { return fail("wspr_hash_commit", e); }
}

#ifdef WSPR_LAB   /* include/wspr_mi355x_bench.h: lab build only */
int wspr_decode_batch_trace(float* idat, float* qdat, int nseg, int samples, size_t seg_stride,
                            struct decoder_options options, struct decoder_results* decodes, int max_results,
                            int* n_results, wspr_trace* trace) {
    LaneTurn lane_turn;
    if (!trace) return -1;
    if (options.usehashtable && nseg > 1)
        return decode_in_order(nseg, n_results, [&](int s) {
            return wspr_decode_batch_trace(idat + (size_t)s * seg_stride, qdat + (size_t)s * seg_stride, 1, samples, seg_stride,
                                           options, decodes + (size_t)s * max_results, max_results, n_results + s, trace + s);
        });
    try {
        // a single traced call with the option reads and writes hashtable.txt itself: in its turn, like wspr_decode_batch()
        std::unique_ptr<HashChain::Ticket> turn;
        if (options.usehashtable) { turn.reset(new HashChain::Ticket(HashChain::get())); turn->wait_turn(); }
        if (samples > wspr::kMaxSamples) {
            // the reference derives its block count from `samples` (wsprd.c:516) and would read past the 45 000 samples
            // its callers hold; this library's working rows are 45 000 samples, so a longer record is refused, not cut
            fprintf(stderr, "libwspr_mi355x: samples = %d exceeds the %d this library decodes\n", samples, wspr::kMaxSamples);
            for (int s = 0; s < nseg; ++s) n_results[s] = 0;
            return -2;
        }
        return decode_split(nseg, samples, options, decodes, max_results, n_results,
                            [&](Context& c, int lo, int n) {
                                c.load_host(idat + (size_t)lo * seg_stride, qdat + (size_t)lo * seg_stride, n, samples, seg_stride);
                            },
                            [&](Context& c, int lo, const std::vector<int>& segs) {
                                c.reload_rows(idat + (size_t)lo * seg_stride, qdat + (size_t)lo * seg_stride, false, seg_stride, samples, segs);
                            },
                            false, idat, qdat, seg_stride, trace);
    } catch (const std::exception& e) {
        for (int s = 0; s < nseg; ++s) n_results[s] = 0;
        return fail("wspr_decode_batch_trace", e);
    }
}
#endif  // WSPR_LAB

int wspr_decode_batch_device(const void* d_idat, const void* d_qdat, int nseg, int samples, size_t seg_stride,
                             struct decoder_options options, struct decoder_results* decodes, int max_results,
                             int* n_results) {
    LaneTurn lane_turn;
    try {
        if (samples > wspr::kMaxSamples) {
            // the reference derives its block count from `samples` (wsprd.c:516) and would read past the 45 000 samples
            // its callers hold; this library's working rows are 45 000 samples, so a longer record is refused, not cut
            fprintf(stderr, "libwspr_mi355x: samples = %d exceeds the %d this library decodes\n", samples, wspr::kMaxSamples);
            for (int s = 0; s < nseg; ++s) n_results[s] = 0;
            return -2;
        }
        const float* di = static_cast<const float*>(d_idat);
        const float* dq = static_cast<const float*>(d_qdat);
        auto load = [&](Context& c, int lo, int n) {
            c.load_device(di + (size_t)lo * seg_stride, dq + (size_t)lo * seg_stride, n, samples, seg_stride);
        };
        auto reload = [&](Context& c, int lo, const std::vector<int>& segs) {
            c.reload_rows(di + (size_t)lo * seg_stride, dq + (size_t)lo * seg_stride, true, seg_stride, samples, segs);
        };
        if (options.usehashtable && nseg > 1)             // the hash memory orders the segments: parallel all the same
            return decode_hashed(nseg, samples, options, decodes, max_results, n_results, load, reload, false, nullptr, nullptr,
                                 seg_stride, 0, nullptr, 0, 0, nullptr, 0, nullptr, nullptr);
        std::unique_ptr<HashChain::Ticket> turn;
        if (options.usehashtable) { turn.reset(new HashChain::Ticket(HashChain::get())); turn->wait_turn(); }
        return decode_split(nseg, samples, options, decodes, max_results, n_results, load, reload, false, nullptr, nullptr, seg_stride);
    } catch (const std::exception& e) {
        for (int s = 0; s < nseg; ++s) n_results[s] = 0;
        return fail("wspr_decode_batch_device", e);
    }
}

// Pins caller memory for the host-buffer entry points (hipHostRegister without the caller needing HIP headers): the
// reference's callers keep their I/Q buffers for the life of the process (rtlsdr_wsprd.c:78-90, 331-336), so they
// pin them once and every wspr_decode*() call on them is a plain DMA.
int wspr_pin_host_buffer(void* p, size_t bytes) {
    if (!p || !bytes) return -1;
    try { Context::get(); } catch (const std::exception& e) { return fail("wspr_pin_host_buffer", e); }
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e == hipSuccess || e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return 0; }
    fprintf(stderr, "libwspr_mi355x: wspr_pin_host_buffer: %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    return -1;
}
int wspr_unpin_host_buffer(void* p) {
    if (!p) return -1;
    const hipError_t e = hipHostUnregister(p);
    (void)hipGetLastError();
    return e == hipSuccess ? 0 : -1;
}

void wspr_shard_range(int nseg, int shard, int nshards, int* lo, int* hi) {
    if (nshards < 1) nshards = 1;
    const int base = nseg / nshards, rem = nseg % nshards;
    const int a = shard * base + (shard < rem ? shard : rem);
    if (lo) *lo = a;
    if (hi) *hi = a + base + (shard < rem ? 1 : 0);
}

// The CPU share of a node-level call lasts as long as the call: raised on entry (to the largest share any call in
// flight asks for), back to 1 when the last such call returns -- so that later single-device calls size their slots
// and pick their Fano placement for the whole host again.  (Pools of contexts CREATED during the call keep the
// share they were sized for.)
namespace {
struct NodeShareGuard {
    // live shares and the published maximum change together under one mutex: a guard that ends while another call
    // begins can no longer publish a share of 1 over the newcomer's (advisor, round 4)
    static std::mutex& mu() { static std::mutex m; return m; }
    static std::vector<int>& live() { static std::vector<int> v; return v; }
    static void publish() {
        int share = 1;
        for (int n : live()) share = std::max(share, n);
        wspr::node_share().store(share);
    }
    int mine;
    explicit NodeShareGuard(int ndevices) : mine(ndevices) {
        std::lock_guard<std::mutex> g(mu());
        live().push_back(mine);
        publish();
    }
    ~NodeShareGuard() {
        std::lock_guard<std::mutex> g(mu());
        auto& v = live();
        for (size_t i = 0; i < v.size(); ++i) if (v[i] == mine) { v.erase(v.begin() + (long)i); break; }
        publish();
    }
};
}  // namespace

// One host process, every GPU of the node: contiguous blocks of segments, one host thread per device,
// each block through wspr_decode_batch() on its device (H2D of the block, decode, spots straight into the caller's
// arrays).  No collective: the segments are independent (wsprd.c:478-479).
int wspr_decode_batch_node(float* idat, float* qdat, int nseg, int samples, size_t seg_stride,
                           struct decoder_options options, struct decoder_results* decodes, int max_results,
                           int* n_results, int ndevices) {
    const int count = wspr_device_count();
    if (count <= 0) {
        fprintf(stderr, "libwspr_mi355x: wspr_decode_batch_node: no HIP device visible (there is no CPU fallback)\n");
        for (int s = 0; s < nseg; ++s) n_results[s] = 0;
        return -1;
    }
    if (ndevices <= 0) ndevices = count;
    const char* virt = wspr::lab_env("WSPR_NODE_VIRTUAL"); // test hook (lab build only): more shards than devices, folded onto lanes
    const int per_dev = (ndevices + count - 1) / count;
    if ((ndevices > count && !(virt && atoi(virt))) || ndevices > Context::kMaxDevices ||
        Context::lane() + per_dev > Context::kUserLanes) {
        fprintf(stderr, "libwspr_mi355x: wspr_decode_batch_node: %d devices asked for, %d visible\n", ndevices, count);
        for (int s = 0; s < nseg; ++s) n_results[s] = 0;
        return -1;
    }
    if (options.usehashtable && nseg > 1)                // ordered by definition: nothing to spread
        return wspr_decode_batch(idat, qdat, nseg, samples, seg_stride, options, decodes, max_results, n_results, 0);
    NodeShareGuard share(ndevices);
    const int lane0 = Context::lane();
    int home = 0;
    (void)hipGetDevice(&home);
    std::vector<int> rcs(ndevices, 0);
    std::vector<std::thread> th;
    for (int k = 0; k < ndevices; ++k) {
        int lo = 0, hi = 0;
        wspr_shard_range(nseg, k, ndevices, &lo, &hi);
        if (hi <= lo) continue;
        th.emplace_back([=, &rcs] {
            if (hipSetDevice(k % count) != hipSuccess) {
                rcs[k] = -1;
                for (int s = lo; s < hi; ++s) n_results[s] = 0;
                return;
            }
            Context::bind_lane(lane0 + k / count);
            rcs[k] = wspr_decode_batch(idat + (size_t)lo * seg_stride, qdat + (size_t)lo * seg_stride, hi - lo, samples,
                                       seg_stride, options, decodes + (size_t)lo * max_results, max_results,
                                       n_results + lo, 0);
        });
    }
    for (auto& t : th) t.join();
    (void)hipSetDevice(home);
    int rc = 0;
    for (int k = 0; k < ndevices; ++k) if (rcs[k] < rc) rc = rcs[k];
    if (rc < 0) for (int s = 0; s < nseg; ++s) n_results[s] = 0;      // as the _device variant: a failed call reports no spots
    return rc;
}

// The same fan-out for input that is already RESIDENT on one device (e.g. the front end's output on the GPU a
// receiver bank feeds): every other device pulls its block of rows over xGMI with a peer copy (one process: a peer
// DMA is what an ncclSend/ncclRecv pair between two devices of the same process comes down to), decodes it, and the
// spots land in the caller's arrays.  SURVEY 8e: "for real inputs ... (scatter) of 360 000 B per segment".
int wspr_decode_batch_node_device(const void* d_idat, const void* d_qdat, int src_device, int nseg, int samples,
                                  size_t seg_stride, struct decoder_options options, struct decoder_results* decodes,
                                  int max_results, int* n_results, int ndevices) {
    const int count = wspr_device_count();
    if (count <= 0 || src_device < 0 || src_device >= count) {
        fprintf(stderr, "libwspr_mi355x: wspr_decode_batch_node_device: no such source device %d (%d visible)\n", src_device, count);
        for (int s = 0; s < nseg; ++s) n_results[s] = 0;
        return -1;
    }
    if (ndevices <= 0) ndevices = count;
    const char* virt = wspr::lab_env("WSPR_NODE_VIRTUAL");
    const int per_dev = (ndevices + count - 1) / count;
    if ((ndevices > count && !(virt && atoi(virt))) || ndevices > Context::kMaxDevices ||
        Context::lane() + per_dev > Context::kUserLanes) {
        fprintf(stderr, "libwspr_mi355x: wspr_decode_batch_node_device: %d devices asked for, %d visible\n", ndevices, count);
        for (int s = 0; s < nseg; ++s) n_results[s] = 0;
        return -1;
    }
    int home = 0;
    (void)hipGetDevice(&home);
    const float* si = static_cast<const float*>(d_idat);
    const float* sq = static_cast<const float*>(d_qdat);
    if (options.usehashtable && nseg > 1) {              // ordered by definition: decoded where the data is
        (void)hipSetDevice(src_device);
        const int rc = wspr_decode_batch_device(si, sq, nseg, samples, seg_stride, options, decodes, max_results, n_results);
        (void)hipSetDevice(home);
        return rc;
    }
    NodeShareGuard share(ndevices);
    const int lane0 = Context::lane();
    std::vector<int> rcs(ndevices, 0);
    std::vector<std::thread> th;
    for (int k = 0; k < ndevices; ++k) {
        int lo = 0, hi = 0;
        wspr_shard_range(nseg, k, ndevices, &lo, &hi);
        if (hi <= lo) continue;
        th.emplace_back([=, &rcs] {
            const int dev = k % count;
            if (hipSetDevice(dev) != hipSuccess) { rcs[k] = -1; return; }
            Context::bind_lane(lane0 + k / count);
            const size_t off = (size_t)lo * seg_stride, floats = (size_t)(hi - lo) * seg_stride;
            const float *pi = si + off, *pq = sq + off;
            void *ti = nullptr, *tq = nullptr;
            // (under the test hook every block but the first takes the copy path, also on the source device itself)
            // fault injection (lab build only): WSPR_NODE_FAIL_PEER=1 makes every peer copy report failure (the staged copy
            // must take over), WSPR_NODE_FAIL_SHARD=k fails shard k outright (the whole call must fail, no spots reported)
            const char* fail_peer = wspr::lab_env("WSPR_NODE_FAIL_PEER");
            const char* fail_shard = wspr::lab_env("WSPR_NODE_FAIL_SHARD");
            if (fail_shard && atoi(fail_shard) == k) {
                fprintf(stderr, "libwspr_mi355x: shard %d (segments %d..%d, device %d) failed [injected]\n", k, lo, hi, dev);
                rcs[k] = -1;
                return;
            }
            if (dev != src_device || (virt && atoi(virt) && k > 0)) {     // pull the block over xGMI
                bool peer_ok = true;
                if (dev != src_device) {
                    // refused peer access (no link, IOMMU policy, ...) is not an error: hipMemcpyPeer then goes through
                    // the host by itself, and if it reports failure all the same the block is staged here explicitly
                    const hipError_t e = hipDeviceEnablePeerAccess(src_device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) peer_ok = false;
                    (void)hipGetLastError();
                }
                bool ok = hipMalloc(&ti, floats * 4) == hipSuccess && hipMalloc(&tq, floats * 4) == hipSuccess;
                bool copied = ok && !(fail_peer && atoi(fail_peer)) &&
                              hipMemcpyPeer(ti, dev, pi, src_device, floats * 4) == hipSuccess &&
                              hipMemcpyPeer(tq, dev, pq, src_device, floats * 4) == hipSuccess &&
                              // a device-to-device copy may return before it has run, and the library's streams are
                              // non-blocking (they do not wait for the null stream): wait here
                              hipStreamSynchronize(nullptr) == hipSuccess;
                if (ok && !copied) {
                    // staged copy: source device -> pinned host -> this device (what the peer copy does without a link)
                    (void)hipGetLastError();
                    void* hp = nullptr;
                    copied = hipHostMalloc(&hp, floats * 4, hipHostMallocDefault) == hipSuccess;
                    for (int rail = 0; copied && rail < 2; ++rail) {
                        copied = hipSetDevice(src_device) == hipSuccess &&
                                 hipMemcpy(hp, rail ? pq : pi, floats * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                                 hipSetDevice(dev) == hipSuccess &&
                                 hipMemcpy(rail ? tq : ti, hp, floats * 4, hipMemcpyHostToDevice) == hipSuccess;
                    }
                    (void)hipSetDevice(dev);
                    if (hp) (void)hipHostFree(hp);
                    if (copied)
                        fprintf(stderr, "libwspr_mi355x: segments %d..%d reached device %d through the host (peer %s)\n", lo, hi,
                                dev, peer_ok ? "copy failed" : "access refused");
                }
                if (!copied) {
                    fprintf(stderr, "libwspr_mi355x: copy of segments %d..%d to device %d failed\n", lo, hi, dev);
                    if (ti) (void)hipFree(ti);
                    if (tq) (void)hipFree(tq);
                    rcs[k] = -1;
                    return;
                }
                pi = static_cast<const float*>(ti); pq = static_cast<const float*>(tq);
            }
            rcs[k] = wspr_decode_batch_device(pi, pq, hi - lo, samples, seg_stride, options,
                                              decodes + (size_t)lo * max_results, max_results, n_results + lo);
            if (ti) (void)hipFree(ti);
            if (tq) (void)hipFree(tq);
        });
    }
    for (auto& t : th) t.join();
    (void)hipSetDevice(home);
    int rc = 0;
    for (int k = 0; k < ndevices; ++k) if (rcs[k] < rc) rc = rcs[k];
    if (rc < 0) for (int s = 0; s < nseg; ++s) n_results[s] = 0;
    return rc;
}

int wspr_decode(float* idat, float* qdat, int samples, struct decoder_options options,
                struct decoder_results* decodes, int* n_results) {
    // the reference caller owns decodes[] with room for its own count (50 in rtlsdr_wsprd.c:117)
    std::vector<decoder_results> tmp(MAX_UNIQUES);
    int n = 0;
    const int rc = wspr_decode_batch(idat, qdat, 1, samples, (size_t)samples, options, tmp.data(), MAX_UNIQUES, &n, 1);
    for (int i = 0; i < n; ++i) decodes[i] = tmp[i];
    *n_results = n;
    return rc < 0 ? rc : 0;
}

void sync_and_demodulate(float* id, float* qd, long np, unsigned char* symbols, float* freq, int ifmin, int ifmax,
                         float fstep, int* shift, int lagmin, int lagmax, int lagstep, float* drift, int symfac,
                         float* sync, int mode) {
    LaneTurn lane_turn;
    try {            // symfac scales the soft symbols of mode 2 (wsprd.c:250); the decoder itself always passes 50 (:427)
        Context::get().demod_single(id, qd, np, symbols, freq, ifmin, ifmax, fstep, shift, lagmin, lagmax, lagstep,
                                    drift, sync, mode, symfac);
    } catch (const std::exception& e) { fail("sync_and_demodulate", e); }
}

void subtract_signal2(float* id, float* qd, long np, float f0, int shift, float drift,
                      const unsigned char* channel_symbols) {
    LaneTurn lane_turn;
    try { Context::get().subtract_single(id, qd, np, f0, shift, drift, channel_symbols); }
    catch (const std::exception& e) { fail("subtract_signal2", e); }
}

void subtract_signal(float* id, float* qd, long np, float f0, int shift, float drift,
                     const unsigned char* channel_symbols) {
    LaneTurn lane_turn;
    try { Context::get().subtract_symbolwise_single(id, qd, np, f0, shift, drift, channel_symbols); }
    catch (const std::exception& e) { fail("subtract_signal", e); }
}

#ifdef WSPR_LAB   /* include/wspr_mi355x_bench.h: lab build only */
int wspr_stage_fft_bank(const float* idat, const float* qdat, int nseg, int samples, size_t seg_stride,
                        float* ps_out) {
    LaneTurn lane_turn;
    try {
        Context& c = Context::get();
        const int blocks = 4 * (samples / wspr::kFftSize) - 1;
        c.load_host(idat, qdat, nseg, samples, seg_stride);
        float* ps = c.ps_buffer(nseg);
        wspr::launch_fft_bank(c.work_i(nseg), c.work_q(nseg), nullptr, nseg, samples, ps, c.tables(), c.stream());
        std::vector<float> h((size_t)nseg * wspr::kPsBins * wspr::kPsTPitch);
        HIP_TRY(hipMemcpyAsync(h.data(), ps, h.size() * 4, hipMemcpyDeviceToHost, c.stream()));
        c.sync();
        memset(ps_out, 0, (size_t)nseg * wspr::kFftSize * blocks * sizeof(float));
        for (int s = 0; s < nseg; ++s)
            for (int t = 0; t < blocks; ++t)
                for (int b = 0; b < wspr::kPsBins; ++b)
                    ps_out[((size_t)s * wspr::kFftSize + (b + wspr::kPsBin0)) * blocks + t] =
                        h[((size_t)s * wspr::kPsBins + b) * wspr::kPsTPitch + t];
        return blocks;
    } catch (const std::exception& e) { return fail("wspr_stage_fft_bank", e); }
}

int wspr_stage_candidates(const float* idat, const float* qdat, int nseg, int samples, size_t seg_stride,
                          int coarse, int maxdrift, struct cand* cand_out, int* npk_out, float* noise_out,
                          float* smspec_out) {
    LaneTurn lane_turn;
    try {
        Context& c = Context::get();
        c.load_host(idat, qdat, nseg, samples, seg_stride);
        TempDev t_noise((size_t)nseg * 4), t_sm((size_t)nseg * wspr::kSmooth * 4);
        float *d_noise = t_noise.as<float>(), *d_sm = t_sm.as<float>();
        c.run_fft_sync(nseg, samples, maxdrift, coarse != 0, nullptr, nseg, d_noise, d_sm);
        std::vector<int> npk;
        std::vector<wspr::DevCand> cd;
        c.fetch_candidates(nseg, npk, cd);
        if (noise_out) HIP_TRY(hipMemcpy(noise_out, d_noise, (size_t)nseg * 4, hipMemcpyDeviceToHost));
        if (smspec_out) HIP_TRY(hipMemcpy(smspec_out, d_sm, (size_t)nseg * wspr::kSmooth * 4, hipMemcpyDeviceToHost));
        for (int s = 0; s < nseg; ++s) {
            npk_out[s] = npk[s];
            for (int j = 0; j < wspr::kMaxCand; ++j) {
                struct cand o = {0, 0, 0, 0, 0};
                if (j < npk[s]) {
                    const wspr::DevCand& v = cd[(size_t)s * wspr::kMaxCand + j];
                    o.freq = v.freq; o.snr = v.snr; o.shift = v.shift; o.drift = v.drift; o.sync = v.sync;
                }
                cand_out[(size_t)s * wspr::kMaxCand + j] = o;
            }
        }
        return 0;
    } catch (const std::exception& e) { return fail("wspr_stage_candidates", e); }
}
#endif  // WSPR_LAB

int wspr_host_pool_workers(void) { return wspr::pool_workers_alive().load(); }

int wspr_last_timings(double* ms, int capacity) {
    // times: the slowest slot (slots run concurrently); counts (index >= 7): summed over the slots -- of the slots
    // the calling thread's LAST batch call ran on (a capped or small call uses fewer than Context::slots(); contexts
    // are never created here)
    try {
        double acc[26] = {0};
        int n = 26;
        const int used = std::max(1, Context::last_slots_used());
        for (int g = 0; g < used; ++g) {
            Context* c = Context::slot_if_exists(g);
            if (!c) continue;
            double t[26] = {0};
            n = c->last_timings(t, 26);
            for (int i = 0; i < n; ++i) acc[i] = (i < 7) ? (t[i] > acc[i] ? t[i] : acc[i]) : acc[i] + t[i];
        }
        n = n < capacity ? n : capacity;
        for (int i = 0; i < n; ++i) ms[i] = acc[i];
        return n;
    } catch (const std::exception& e) { return fail("wspr_last_timings", e); }
}

#ifdef WSPR_LAB   /* include/wspr_mi355x_bench.h: lab build only */
int wspr_bench_fft_sync(const void* d_idat, const void* d_qdat, int nseg, int samples, size_t seg_stride, int iters,
                        double* ms) {
    LaneTurn lane_turn;
    try {
        Context& c = Context::get();
        c.load_device(d_idat, d_qdat, nseg, samples, seg_stride);
        return c.bench_fft_sync(nseg, samples, iters, ms);
    } catch (const std::exception& e) { return fail("wspr_bench_fft_sync", e); }
}

int wspr_bench_valu(const void* d_idat, const void* d_qdat, int nseg, int samples, size_t seg_stride, int iters,
                    double* ms) {
    LaneTurn lane_turn;
    try {
        Context& c = Context::get();
        c.load_device(d_idat, d_qdat, nseg, samples, seg_stride);
        return c.bench_valu(nseg, samples, iters, ms);
    } catch (const std::exception& e) { return fail("wspr_bench_valu", e); }
}

int wspr_calib_copy(const void* d_src, void* d_dst, size_t nfloats, int iters) {
    try {
        Context& c = Context::get();
        for (int i = 0; i < iters; ++i) wspr::launch_calib_copy((const float*)d_src, (float*)d_dst, nfloats, c.stream());
        c.sync();
        return 0;
    } catch (const std::exception& e) { return fail("wspr_calib_copy", e); }
}

int wspr_calib_copy16(const void* d_src, void* d_dst, size_t nfloats, int iters, int variant, double* ms) {
    try {
        if ((nfloats & 3) || ((uintptr_t)d_src & 15) || ((uintptr_t)d_dst & 15)) return -1;
        Context& c = Context::get();
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, c.stream()));
        for (int i = 0; i < iters; ++i) wspr::launch_calib_copy16((const float*)d_src, (float*)d_dst, nfloats, c.stream(), variant);
        HIP_TRY(hipEventRecord(e1, c.stream()));
        HIP_TRY(hipEventSynchronize(e1));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (ms) *ms = iters > 0 ? t / iters : 0.0;
        return 0;
    } catch (const std::exception& e) { return fail("wspr_calib_copy16", e); }
}

int wspr_calib_valu(int launches, double* tflops) {
    try {
        Context& c = Context::get();
        TempDev out(64);
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        wspr::launch_calib_valu((float*)out.p, 256, c.stream());                  // settle the clocks
        HIP_TRY(hipEventRecord(e0, c.stream()));
        double flops = 0;
        for (int i = 0; i < launches; ++i) flops += wspr::launch_calib_valu((float*)out.p, 2048, c.stream());
        HIP_TRY(hipEventRecord(e1, c.stream()));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (tflops) *tflops = flops / (ms * 1e-3) / 1e12;
        return 0;
    } catch (const std::exception& e) { return fail("wspr_calib_valu", e); }
}
#endif  // WSPR_LAB

int wspr_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int wspr_set_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n || device >= Context::kMaxDevices) {
        fprintf(stderr, "libwspr_mi355x: wspr_set_device(%d): no such HIP device (%d visible)\n", device, n);
        return -1;
    }
    return hipSetDevice(device) == hipSuccess ? 0 : -1;
}

int wspr_bind_thread_lane(int lane) {
    // the last lane is the receiver sessions' (wspr_session_feed runs beside a decode): callers get 0 .. kUserLanes - 1
    Context::bind_lane(lane < 0 ? 0 : (lane >= Context::kUserLanes ? Context::kUserLanes - 1 : lane));
    return Context::lane();
}

int wspr_set_thread_slots(int n) {
    Context::cap_slots(n <= 0 ? 8 : n);
    return Context::slot_cap();
}

size_t wspr_release_buffers(void) {
    try {
        AllLanesTurn every_lane;                   // calls in flight on this device finish first; new ones wait
        return Context::release_buffers();
    } catch (const std::exception& e) {
        fprintf(stderr, "libwspr_mi355x: wspr_release_buffers failed: %s\n", e.what());
        return 0;
    }
}

unsigned wspr_set_fano_fast_budget(unsigned cycles_per_bit) {
    return wspr::fano_fast_budget().exchange(cycles_per_bit);
}

#ifdef WSPR_LAB   /* include/wspr_mi355x_bench.h: lab build only */
int wspr_set_front_end_cus(int ncus) {
    return wspr::front_end_cus().exchange(ncus < 0 ? 0 : ncus);
}
#endif  // WSPR_LAB

int wspr_set_fano_device_mode(int mode) {
    return wspr::fano_device_setting().exchange(mode < 0 ? -1 : (mode ? 1 : 0));
}

int wspr_fano_batch_device_wave(const unsigned char* symbols, int n, unsigned maxcycles, int* ret, unsigned* cycles,
                                unsigned* metric, unsigned* maxnp, unsigned char* data, unsigned* steps) {
    LaneTurn lane_turn;
    try {
        return Context::get().fano_batch(symbols, n, maxcycles, ret, cycles, metric, maxnp, data, steps);
    } catch (const std::exception& e) { return fail("wspr_fano_batch_device_wave", e); }
}

#ifdef WSPR_LAB   /* include/wspr_mi355x_bench.h: lab build only */
int wspr_bench_decimate(const void* d_raw, size_t bytes_per_seg, int nseg, void* d_idat, void* d_qdat, int iters,
                        double* ms) {
    LaneTurn lane_turn;
    try {
        return Context::get().bench_decimate(d_raw, bytes_per_seg, nseg, (float*)d_idat, (float*)d_qdat, iters, ms);
    } catch (const std::exception& e) { return fail("wspr_bench_decimate", e); }
}

int wspr_calib_read(const void* d_raw, size_t bytes_per_seg, int nseg, int iters, double* ms) {
    try {
        return Context::get().bench_decimate(d_raw, bytes_per_seg, nseg, nullptr, nullptr, iters, ms);
    } catch (const std::exception& e) { return fail("wspr_calib_read", e); }
}
#endif  // WSPR_LAB

int wspr_decimate_u8_batch_device(const void* d_raw, size_t bytes_per_seg, int nseg, void* d_idat, void* d_qdat,
                                  int normalise) {
    LaneTurn lane_turn;
    try {
        // rows are read with aligned 16-byte vector loads (a misaligned row stride would also let the last vector of
        // the last row run past the caller's allocation)
        if ((bytes_per_seg & 15) || (reinterpret_cast<uintptr_t>(d_raw) & 15)) {
            fprintf(stderr, "libwspr_mi355x: wspr_decimate_u8_batch_device: d_raw and bytes_per_seg must be multiples of 16\n");
            return -1;
        }
        return Context::get().decimate_device(d_raw, bytes_per_seg, nseg, (float*)d_idat, (float*)d_qdat, normalise, nullptr);
    } catch (const std::exception& e) { return fail("wspr_decimate_u8_batch_device", e); }
}

int wspr_decimate_u8_batch_device_stateful(const void* d_raw, size_t bytes_per_seg, int nseg, void* d_states,
                                           void* d_idat, void* d_qdat, int* n_out) {
    LaneTurn lane_turn;
    try {
        if (!d_states || (bytes_per_seg & 15)) return -1;
        return Context::get().decimate_device(d_raw, bytes_per_seg, nseg, (float*)d_idat, (float*)d_qdat, 0, n_out,
                                              static_cast<wspr::DecimState*>(d_states));
    } catch (const std::exception& e) { return fail("wspr_decimate_u8_batch_device_stateful", e); }
}

void wspr_front_end_constants(float* taps33, int* samples_per_output) { wspr::front_end_constants(taps33, samples_per_output); }

void wspr_decim_stream_reset(wspr_decim_state* st) {
    if (st) std::memset(st, 0, sizeof *st);
}

int wspr_decimate_u8_stream(wspr_decim_state* st, const uint8_t* iq, size_t nbytes, float* I, float* Q, uint32_t fill,
                            uint32_t capacity, uint32_t* new_fill) {
    LaneTurn lane_turn;
    static_assert(sizeof(wspr_decim_state) == sizeof(wspr::DecimState), "public and device state layouts differ");
    try {
        if (!st || (nbytes & 15)) return -1;
        return Context::get().decimate_stream(reinterpret_cast<wspr::DecimState*>(st), iq, nbytes, I, Q, fill, capacity,
                                              new_fill);
    } catch (const std::exception& e) { return fail("wspr_decimate_u8_stream", e); }
}

int wspr_decimate_u8(const uint8_t* iq, size_t nbytes, float* I, float* Q, uint32_t* n_out, int normalise) {
    LaneTurn lane_turn;
    try {
        Context& c = Context::get();
        nbytes &= ~(size_t)7;
        TempDev raw(nbytes + 16);
        HIP_TRY(hipMemcpy(raw.p, iq, nbytes, hipMemcpyHostToDevice));
        float* wi = c.work_i(1);
        float* wq = c.work_q(1);
        int nout = 0;
        const int rc = c.decimate_device(raw.p, nbytes, 1, wi, wq, normalise, &nout);
        if (rc) return rc;
        HIP_TRY(hipMemcpy(I, wi, (size_t)wspr::kMaxSamples * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(Q, wq, (size_t)wspr::kMaxSamples * 4, hipMemcpyDeviceToHost));
        if (n_out) *n_out = (uint32_t)nout;
        return 0;
    } catch (const std::exception& e) { return fail("wspr_decimate_u8", e); }
}

// ---- receiver session: the reference's double buffer and decoder thread body ----------------
// rx_state of rtlsdr_wsprd.c:78-90 (two I/Q buffers, their fill counters, the active index) plus the
// decimator's static state (:135-160), as one object the application drives from its own threads:
//   RX thread       wspr_session_feed()      = rtlsdr_callback()                       (:126-244)
//   main loop       wspr_session_rollover()  = switch buffers on the even minute       (:1179-1182)
//   decoder thread  wspr_session_decode()    = decoder(): too-short check, zero the tail, normalise, decode (:263-328)
struct wspr_session {
    decoder_options opt;
    wspr::DecimState dec;                      // all zero = the receiver at start-up
    std::vector<float> I[2], Q[2];
    std::atomic<uint32_t> fill[2];
    std::atomic<uint32_t> active;
    // feed() holds it from reading `active` to committing the new fill, rollover() takes it: a roll-over waits for
    // the callback in flight (the reference tests bufferIndex per output sample, rtlsdr_wsprd.c:236-242, so its
    // window is one sample; a GPU round trip must not straddle the switch), and no feed can write into a buffer
    // after rollover() has handed it to the decoder thread.
    std::mutex feed_mu;
};

namespace {
constexpr uint32_t kSessionSamples = 120 * 375;              // SIGNAL_LENGHT * SIGNAL_SAMPLE_RATE
constexpr uint32_t kSessionMinSamples = (120 - 3) * 375;     // rtlsdr_wsprd.c:277
constexpr int kFrontEndLane = Context::kMaxLanes - 1;        // feed() runs beside decode(): its own lane
}  // namespace

extern "C" {

wspr_session* wspr_session_create(struct decoder_options options) {
    wspr_session* s = new (std::nothrow) wspr_session;
    if (!s) return nullptr;
    s->opt = options;
    std::memset(&s->dec, 0, sizeof s->dec);
    for (int b = 0; b < 2; ++b) {
        s->I[b].assign(kSessionSamples, 0.0f);
        s->Q[b].assign(kSessionSamples, 0.0f);
        s->fill[b].store(0);
    }
    s->active.store(0);                                     // initSampleStorage(), rtlsdr_wsprd.c:331-336
    return s;
}

void wspr_session_destroy(wspr_session* s) { delete s; }

int wspr_session_feed(wspr_session* s, const uint8_t* buf, uint32_t len) {
    if (!s || !buf || (len & 15u)) return -1;
    const int caller_lane = Context::lane();
    std::lock_guard<std::mutex> hold(s->feed_mu);
    try {
        Context::bind_lane(kFrontEndLane);
        // every session's front end runs on the ONE lane reserved for it: the RX threads of several receivers take
        // turns at its context (stream, staging buffers) -- a callback is ~0.1 ms of it against the 13.65 ms it covers
        LaneTurn lane_turn;
        const uint32_t idx = s->active.load();
        uint32_t nf = s->fill[idx].load();
        const int rc = Context::get().decimate_stream(&s->dec, buf, len, s->I[idx].data(), s->Q[idx].data(), nf,
                                                      kSessionSamples, &nf);           // :236-242: full buffer drops the rest
        Context::bind_lane(caller_lane);
        if (rc) return rc;
        s->fill[idx].store(nf);
        return (int)nf;
    } catch (const std::exception& e) {
        Context::bind_lane(caller_lane);
        return fail("wspr_session_feed", e);
    }
}

int wspr_session_feed_many(wspr_session* const* sessions, const uint8_t* const* bufs, uint32_t len, int n, int* fills) {
    if (!sessions || !bufs || n < 0 || (len & 15u)) return -1;
    if (n == 0) return 0;
    std::vector<wspr_session*> order(sessions, sessions + n);
    for (int k = 0; k < n; ++k) if (!sessions[k] || !bufs[k]) return -1;
    std::sort(order.begin(), order.end());                      // one locking order for every caller; duplicates refused
    if (std::adjacent_find(order.begin(), order.end()) != order.end()) return -1;
    std::vector<std::unique_lock<std::mutex>> held;
    for (wspr_session* s : order) held.emplace_back(s->feed_mu);
    const int caller_lane = Context::lane();
    try {
        Context::bind_lane(kFrontEndLane);
        LaneTurn lane_turn;
        std::vector<wspr::DecimState*> st(n);
        std::vector<float*> I(n), Q(n);
        std::vector<uint32_t> fill(n), nf(n);
        std::vector<uint32_t> idx(n);
        for (int k = 0; k < n; ++k) {
            idx[k] = sessions[k]->active.load();
            st[k] = &sessions[k]->dec;
            I[k] = sessions[k]->I[idx[k]].data();
            Q[k] = sessions[k]->Q[idx[k]].data();
            fill[k] = sessions[k]->fill[idx[k]].load();
        }
        const int rc = Context::get().decimate_stream_many(st.data(), bufs, len, n, I.data(), Q.data(), fill.data(),
                                                           kSessionSamples, nf.data());
        Context::bind_lane(caller_lane);
        if (rc) return rc;
        for (int k = 0; k < n; ++k) {
            sessions[k]->fill[idx[k]].store(nf[k]);
            if (fills) fills[k] = (int)nf[k];
        }
        return 0;
    } catch (const std::exception& e) {
        Context::bind_lane(caller_lane);
        return fail("wspr_session_feed_many", e);
    }
}

int wspr_session_rollover(wspr_session* s) {
    if (!s) return -1;
    std::lock_guard<std::mutex> hold(s->feed_mu);            // not while a callback's outputs are still on their way
    const uint32_t prev = s->active.load(), next = prev ^ 1u;
    s->fill[next].store(0);                                 // rx_state.iqIndex[rx_state.bufferIndex] = 0
    s->active.store(next);
    return (int)prev;
}

uint32_t wspr_session_fill(const wspr_session* s, int buffer) { return (s && (buffer & ~1) == 0) ? s->fill[buffer].load() : 0u; }

const float* wspr_session_samples(const wspr_session* s, int buffer, int rail) {
    if (!s || (buffer & ~1) != 0) return nullptr;
    return rail ? s->Q[buffer].data() : s->I[buffer].data();
}

}  // extern "C"

namespace {
// decoder()'s preparation of a completed buffer, rtlsdr_wsprd.c:277-305: false if it is too short to decode, else
// the tail zeroed and both rails scaled to a peak of 0.5
bool session_prepare(wspr_session* s, int buffer) {
    const uint32_t n = s->fill[buffer].load();
    if (n < kSessionMinSamples) return false;              // "Signal too short, skipping!" (:277-280)
    float* I = s->I[buffer].data();
    float* Q = s->Q[buffer].data();
    for (uint32_t i = n; i < kSessionSamples; ++i) { I[i] = 0.0f; Q[i] = 0.0f; }     // :284-288
    float peak = 1e-24f;                                    // :290-305
    for (uint32_t i = 0; i < kSessionSamples; ++i) {
        const float a = fabsf(I[i]), b = fabsf(Q[i]);
        if (a > peak) peak = a;
        if (b > peak) peak = b;
    }
    const float scale = (float)(0.5 / (double)peak);
    for (uint32_t i = 0; i < kSessionSamples; ++i) { I[i] *= scale; Q[i] *= scale; }
    return true;
}
}  // namespace

extern "C" {

int wspr_session_decode(wspr_session* s, int buffer, struct decoder_results* decodes, int* n_results) {
    if (!s || (buffer & ~1) != 0 || !n_results) return -1;
    *n_results = 0;
    if (!session_prepare(s, buffer)) return 0;
    const int rc = wspr_decode(s->I[buffer].data(), s->Q[buffer].data(), (int)kSessionSamples, s->opt, decodes, n_results);   // :312-317
    return rc < 0 ? rc : 1;
}

// Many receivers, one slot: the completed buffers of n sessions decoded TOGETHER -- the decoder thread's body
// (rtlsdr_wsprd.c:263-328) for every receiver of a service in one batch call per distinct set of decoder options
// (receivers of one band share theirs; `freq` enters the reported frequency in double precision, so receivers with
// different options are not folded into one call).  Results, and what the buffers hold afterwards, are those of
// wspr_session_decode() on each session in index order -- with usehashtable that order is the order of the hash memory,
// and only runs of consecutive sessions with equal options share a call (see `ordered` below).
int wspr_session_decode_many(wspr_session* const* sessions, const int* buffers, int n, struct decoder_results* decodes,
                             int max_results, int* n_results, int* decoded) {
    if (!sessions || !buffers || n < 0 || !decodes || max_results < 1 || !n_results) return -1;
    for (int k = 0; k < n; ++k) {
        n_results[k] = 0;
        if (decoded) decoded[k] = 0;
        if (!sessions[k] || (buffers[k] & ~1) != 0) return -1;
    }
    std::vector<int> ready;
    for (int k = 0; k < n; ++k)
        if (session_prepare(sessions[k], buffers[k])) { ready.push_back(k); if (decoded) decoded[k] = 1; }
    std::vector<char> taken(ready.size(), 0);
    std::vector<float> I, Q;
    std::vector<decoder_results> out;
    std::vector<int> nout, group;
    // the hash memory (hashtable.txt) is shared by every session with the option and ordered by the calls: as soon as one
    // ready session uses it, only RUNS of consecutive sessions with equal options are folded, so that the memory sees
    // the sessions in index order whatever their options (opt A, opt B, opt A stays 0, 1, 2 -- not 0, 2, 1)
    bool ordered = false;
    for (int k : ready) ordered = ordered || sessions[k]->opt.usehashtable != 0;
    for (size_t a = 0; a < ready.size(); ++a) {
        if (taken[a]) continue;
        const decoder_options& opt = sessions[ready[a]]->opt;
        group.clear();
        for (size_t b = a; b < ready.size(); ++b) {
            const bool same = !taken[b] && std::memcmp(&sessions[ready[b]]->opt, &opt, sizeof opt) == 0;
            if (same) { taken[b] = 1; group.push_back(ready[b]); }
            else if (ordered) break;
        }
        const int m = (int)group.size();
        int rc;
        if (m == 1) {
            wspr_session* s = sessions[group[0]];
            const int b = buffers[group[0]];
            rc = wspr_decode_batch(s->I[b].data(), s->Q[b].data(), 1, (int)kSessionSamples, kSessionSamples, opt,
                                   decodes + (size_t)group[0] * max_results, max_results, n_results + group[0], 1);
        } else {
            I.resize((size_t)m * kSessionSamples); Q.resize((size_t)m * kSessionSamples);
            out.assign((size_t)m * max_results, decoder_results{});
            nout.assign((size_t)m, 0);
            for (int g = 0; g < m; ++g) {
                std::memcpy(I.data() + (size_t)g * kSessionSamples, sessions[group[g]]->I[buffers[group[g]]].data(), kSessionSamples * sizeof(float));
                std::memcpy(Q.data() + (size_t)g * kSessionSamples, sessions[group[g]]->Q[buffers[group[g]]].data(), kSessionSamples * sizeof(float));
            }
            rc = wspr_decode_batch(I.data(), Q.data(), m, (int)kSessionSamples, kSessionSamples, opt, out.data(), max_results, nout.data(), 1);
            if (rc >= 0)
                for (int g = 0; g < m; ++g) {               // spots, and the residual the single call leaves in the buffer
                    std::memcpy(decodes + (size_t)group[g] * max_results, out.data() + (size_t)g * max_results, (size_t)nout[g] * sizeof(decoder_results));
                    n_results[group[g]] = nout[g];
                    std::memcpy(sessions[group[g]]->I[buffers[group[g]]].data(), I.data() + (size_t)g * kSessionSamples, kSessionSamples * sizeof(float));
                    std::memcpy(sessions[group[g]]->Q[buffers[group[g]]].data(), Q.data() + (size_t)g * kSessionSamples, kSessionSamples * sizeof(float));
                }
        }
        if (rc < 0) { for (int k = 0; k < n; ++k) n_results[k] = 0; return rc; }
    }
    return (int)ready.size();
}

uint32_t wspr_usec_to_next_slot(long tv_sec, long tv_usec) {                              // :1170-1175
    const uint32_t sec = (uint32_t)(tv_sec % 120);
    const uint32_t usec = sec * 1000000u + (uint32_t)tv_usec;
    return 120000000u - usec;
}

void wspr_frame_time(long unixtime_now, int* year, int* month, int* day, int* hour, int* minute) {   // :307-310
    time_t t = (time_t)unixtime_now - 120 + 1;
    struct tm g;
    gmtime_r(&t, &g);
    if (year) *year = g.tm_year + 1900;
    if (month) *month = g.tm_mon + 1;
    if (day) *day = g.tm_mday;
    if (hour) *hour = g.tm_hour;
    if (minute) *minute = g.tm_min;
}

}  // extern "C"

// ---- recorded-file formats ----------------------------------------
namespace {
// max-abs normalisation to 0.5 of the first n samples, as every reader of the reference does
// (rtlsdr_wsprd.c:574-589, :649-664)
void normalise_host(float* I, float* Q, int n) {
    float peak = 1e-24f;
    for (int i = 0; i < n; ++i) {
        const float a = fabsf(I[i]), b = fabsf(Q[i]);
        if (a > peak) peak = a;
        if (b > peak) peak = b;
    }
    const float scale = (float)(0.5 / (double)peak);
    for (int i = 0; i < n; ++i) { I[i] *= scale; Q[i] *= scale; }
}
int load_interleaved(FILE* fd, float* I, float* Q) {
    std::vector<float> buf(2 * (size_t)wspr::kMaxSamples);
    const int nread = (int)fread(buf.data(), sizeof(float), buf.size(), fd);
    const int n = nread / 2;
    for (int i = 0; i < n; ++i) { I[i] = buf[2 * i]; Q[i] = -buf[2 * i + 1]; }   // Q sign: wsprsim convention
    normalise_host(I, Q, n);
    return n;
}
}  // namespace

// readRawIQfile(), rtlsdr_wsprd.c:555-592: interleaved float32 I/Q, Q negated, normalised.
// I/Q must hold 45000 floats; returns the number of complex samples read (0 on error).
int wspr_read_iq_file(const char* filename, float* I, float* Q) {
    FILE* fd = fopen(filename, "rb");
    if (!fd) { fprintf(stderr, "Cannot open data file...\n"); return 0; }
    const int n = load_interleaved(fd, I, Q);
    fclose(fd);
    return n;
}

// readC2file(), rtlsdr_wsprd.c:620-667: 14-byte name, int type, double dial frequency, then the
// same interleaved payload.  *dial_hz receives the header frequency (the reference stores it in
// rx_options.dialfreq, :637).
int wspr_read_c2_file(const char* filename, float* I, float* Q, double* dial_hz) {
    FILE* fd = fopen(filename, "rb");
    if (!fd) { fprintf(stderr, "Cannot open data file...\n"); return 0; }
    char name[15];
    int type = 0;
    double frequency = 0.0;
    size_t got = fread(name, sizeof(char), 14, fd);
    got += fread(&type, sizeof(int), 1, fd);
    got += fread(&frequency, sizeof(double), 1, fd);
    (void)got;
    if (dial_hz) *dial_hz = frequency;
    const int n = load_interleaved(fd, I, Q);
    fclose(fd);
    return n;
}

// writeRawIQfile(), rtlsdr_wsprd.c:595-617: always 45000 complex samples, Q negated.
int wspr_write_iq_file(const char* filename, const float* I, const float* Q) {
    FILE* fd = fopen(filename, "wb");
    if (!fd) { fprintf(stderr, "Cannot open data file...\n"); return 0; }
    std::vector<float> buf(2 * (size_t)wspr::kMaxSamples);
    for (int i = 0; i < wspr::kMaxSamples; ++i) { buf[2 * i] = I[i]; buf[2 * i + 1] = -Q[i]; }
    const size_t nw = fwrite(buf.data(), sizeof(float), buf.size(), fd);
    fclose(fd);
    if (nw != buf.size()) { fprintf(stderr, "Cannot write all the data!\n"); return 0; }
    return wspr::kMaxSamples;
}

// The -r playback spot line, rtlsdr_wsprd.c:691-701 (without the trailing newline).
int wspr_format_spot(const struct decoder_results* r, char* out, size_t cap) {
    return snprintf(out, cap, "Spot : %6.2f %6.2f %10.6f %2d %7s %6s %2s", r->snr, r->dt, r->freq, (int)r->drift,
                    r->call, r->loc, r->pwr);
}

// ---- live-receiver output formats (formatting only, no network) ------
// printSpots(), rtlsdr_wsprd.c:447-474: the daemon's stdout line with the UTC frame time.
int wspr_format_spot_timestamped(const struct decoder_results* r, int year, int month, int day, int hour, int minute,
                                 char* out, size_t cap) {
    return snprintf(out, cap, "Spot :  %04d-%02d-%02d %02d:%02dz %6.2f %6.2f %10.6f %2d %7s %6s %2s", year, month, day,
                    hour, minute, r->snr, r->dt, r->freq, (int)r->drift, r->call, r->loc, r->pwr);
}

// The wsprnet.org report URL of postSpots(), rtlsdr_wsprd.c:414-429 (spot) and :390-397 (empty
// report when r == NULL).  Only the text is produced; nothing is sent.
int wspr_format_wsprnet_url(const struct decoder_results* r, const struct decoder_options* opt, double dial_hz,
                            int year, int month, int day, int hour, int minute, const char* app_version,
                            char* out, size_t cap) {
    const std::string rcall = url_escape(opt->rcall), rloc = url_escape(opt->rloc);
    if (!r)
        return snprintf(out, cap,
                        "https://wsprnet.org/post?function=wsprstat&rcall=%s&rgrid=%s&rqrg=%.6f&tpct=%.2f&tqrg=%.6f&dbm=%d&version=%s&mode=2",
                        rcall.c_str(), rloc.c_str(), dial_hz / 1e6, 0.0f, dial_hz / 1e6, 0, app_version);
    return snprintf(out, cap,
                    "https://wsprnet.org/post?function=wspr&rcall=%s&rgrid=%s&rqrg=%.6f&date=%02d%02d%02d&time=%02d%02d&sig=%.0f&dt=%.1f&tqrg=%.6f&tcall=%s&tgrid=%s&dbm=%s&version=%s&mode=2",
                    rcall.c_str(), rloc.c_str(), r->freq, year % 100, month, day, hour, minute, r->snr, r->dt, r->freq,
                    r->call, r->loc, r->pwr, app_version);
}

// ---- message layer under the reference's names ---------------------------------
char get_locator_character_code(char ch) { return wspr::locator_code(ch); }
char get_callsign_character_code(char ch) { return wspr::callsign_code(ch); }
long unsigned int pack_grid4_power(char const* grid4, int power) { return wspr::pack_grid_power(grid4, power); }
long unsigned int pack_call(char const* callsign) { return wspr::pack_callsign(callsign); }
void pack_prefix(char* callsign, int32_t* n, int32_t* m, int32_t* nadd) { wspr::pack_compound(callsign, n, m, nadd); }
void interleave(unsigned char* sym) { wspr::interleave162(sym); }
void deinterleave(unsigned char* sym) { wspr::deinterleave162(sym); }
int get_wspr_channel_symbols(char* message, char* hashtab, char* loctab, unsigned char* symbols) {
    return wspr::channel_symbols(message, hashtab, loctab, symbols);
}
void unpack50(signed char* dat, int32_t* n1, int32_t* n2) { wspr::unpack_50bits(dat, n1, n2); }
int unpackcall(int32_t ncall, char* call) { return wspr::unpack_callsign(ncall, call); }
int unpackgrid(int32_t ngrid, char* grid) { return wspr::unpack_grid(ngrid, grid); }
int unpackpfx(int32_t nprefix, char* call) { return wspr::unpack_prefix(nprefix, call); }
int unpk_(signed char* message, char* hashtab, char* loctab, char* call_loc_pow, char* call, char* loc, char* pwr,
          char* callsign) {
    return wspr::unpack_message(message, hashtab, loctab, call_loc_pow, call, loc, pwr, callsign);
}
int fano(unsigned int* metric, unsigned int* cycles, unsigned int* maxnp, unsigned char* data,
         unsigned char* symbols, unsigned int nbits, int mettab[2][256], int delta, unsigned int maxcycles) {
    return wspr::fano_decode(metric, cycles, maxnp, data, symbols, nbits, mettab, delta, maxcycles);
}
int encode(unsigned char* symbols, unsigned char* data, unsigned int nbytes) {
    return wspr::conv_encode(symbols, data, nbytes);
}
uint32_t nhash(const void* key, size_t length, uint32_t initval) { return wspr::nhash15(key, length, initval); }
void wspr_fano_metric_table(int mettab[2][256]) {
    memcpy(mettab, wspr::default_metrics().tab, sizeof(int) * 512);
}
int doublecomp(const void* a, const void* b) {
    const double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : (x > y);
}
int floatcomp(const void* a, const void* b) {
    const float x = *(const float*)a, y = *(const float*)b;
    return x < y ? -1 : (x > y);
}
// metric_tables (reference wsprd/metric_tables.h:8): a writable data symbol like the reference's, filled from the
// bit patterns before anything else runs
float metric_tables[5][256];
__attribute__((constructor)) static void metric_tables_init(void) {
    static_assert(sizeof(metric_tables) == sizeof(kMetricTableBits), "table shape");
    memcpy(metric_tables, kMetricTableBits, sizeof(metric_tables));
}
// 8-bit parity table (reference wsprd/tab.c:7), generated
unsigned char Partab[256];
__attribute__((constructor)) static void partab_init(void) {
    for (int i = 0; i < 256; ++i) Partab[i] = (unsigned char)__builtin_parity((unsigned)i);
}

}  // extern "C"

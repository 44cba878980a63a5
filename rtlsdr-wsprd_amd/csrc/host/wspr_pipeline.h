// Context of the MI355X WSPR decoder (one per process, created lazily on the first
// call): HIP stream, constant tables, grow-only device/pinned buffers, host pool.
#pragma once
#include <chrono>
#include <atomic>
#include <functional>
#include <string>
#include <unordered_map>
#include <memory>
#include <vector>

#include "../../../include/wspr_mi355x.h"
#include "../../../include/wspr_mi355x_bench.h"   // declarations only: the definitions exist in the lab build (-DWSPR_LAB)
#include "../kernels/wspr_device.h"
#include "wspr_hashmem.h"

namespace wspr {

// Fano attempts the host pool left unfinished (see Context::decode_resident)
// cycles/bit the host Fano pool spends before leaving an attempt to the device tail (K6)
std::atomic<unsigned>& fano_fast_budget();
// -1 automatic, 0 host pool only, 1 device search for every attempt of a batch (see wspr_pipeline.hip)
std::atomic<int>& fano_device_setting();
// worker threads of all host pools alive in this process
std::atomic<int>& pool_workers_alive();
// shards of a node-level call sharing this host's CPUs (see wspr_decode_batch_node)
std::atomic<int>& node_share();
// CUs the front end (K0) may occupy, 0 = all (wspr_set_front_end_cus / WSPR_K0_CUS)
std::atomic<int>& front_end_cus();

struct PendingFano {
    std::vector<int> seg;                 // owning segment of each attempt
    std::vector<unsigned char> sym;       // 162 soft symbols each, transmission order
    void add(int s, const unsigned char* v) { seg.push_back(s); sym.insert(sym.end(), v, v + 162); }
};

// Exact full-budget results of Fano attempts that were already run, keyed by the soft-symbol vector
// (the search is a pure function of it): the re-decode of a segment replays the same vectors up to
// the point where the late success changes the IQ, and takes their results from here.
struct FanoMemo {
    struct Entry { int ret; unsigned cycles; unsigned char data[11]; };
    std::unordered_map<std::string, Entry> map;
    void add(const unsigned char* sym, int ret, unsigned cycles, const unsigned char* data10) {
        Entry e{ret, cycles, {0}};
        for (int k = 0; k < 10; ++k) e.data[k] = data10[k];
        map.emplace(std::string(reinterpret_cast<const char*>(sym), 162), e);
    }
    const Entry* find(const unsigned char* sym) const {
        const auto it = map.find(std::string(reinterpret_cast<const char*>(sym), 162));
        return it == map.end() ? nullptr : &it->second;
    }
};

class Context {
public:
    static Context& get();          // slot 0; throws std::runtime_error when no HIP device is usable
    static Context& slot(int i);    // i in [0, slots())
    static int slots();             // concurrent pipelines per process (env WSPR_SLOTS, default 3)
    static constexpr int kMaxLanes = 17;         // lanes 0..15 for callers, the last one for receiver sessions (feed)
    static constexpr int kUserLanes = kMaxLanes - 1;
    static constexpr int kMaxDevices = 16;
    static int lane();              // lane of the calling thread
    static void bind_lane(int lane);
    static void cap_slots(int n);   // calling thread: use at most n slots per batch (a shard's share of the host)
    static size_t release_buffers();   // current device, every lane and slot: bytes of device memory returned
    static int slot_cap();          // min(slots(), the thread's cap, the CPUs of its share)
    static Context* slot_if_exists(int i);   // slot i of the calling thread's (device, lane) if it was ever created
    static void note_slots_used(int n);      // slots the calling thread's current batch call runs on ...
    static int last_slots_used();            // ... as wspr_last_timings() reads it back
    int device();
    ~Context();

    hipStream_t stream();
    hipStream_t front_end_stream();   // the CU-masked stream of K0 when a share is set, else stream()
    const DeviceTables& tables();
    int host_threads();

    // working IQ buffers (planar, rows of kIqStride floats)
    float* work_i(int nseg);
    float* work_q(int nseg);
    void load_host(const float* I, const float* Q, int nseg, int samples, size_t stride);
    void load_device(const void* dI, const void* dQ, int nseg, int samples, size_t stride);
    void store_host(float* I, float* Q, int nseg, int samples, size_t stride);
    void reload_rows(const float* I, const float* Q, bool device, size_t stride, int samples,
                     const std::vector<int>& segs);
    void sync();

    float* ps_buffer(int nseg);
    void run_fft_sync(int nseg, int samples, int maxdrift, bool coarse, const int* d_seglist, int nactive,
                      float* noise_out, float* smspec_out);
    void fetch_candidates(int nseg, std::vector<int>& npk, std::vector<DevCand>& cand);
    void fetch_candidates_async(int nseg);                 // copies queued on the stream ...
    void finish_fetch_candidates(int nseg, std::vector<int>& npk, std::vector<DevCand>& cand);   // ... consumed after a wait

    // the decoder proper, on the working buffers
    // reload(segs): restore the original IQ of the listed segments in the working buffers (needed
    // for the exact re-decode after a late Fano success); empty function = no fast/tail split
    // hb / hb_off: the batch's shared hash memory and the index of this context's segment 0 within the call
    int decode_resident(int nseg, int samples, const decoder_options& opt, decoder_results* out,
                        int max_results, int* n_results,
                        const std::function<void(const std::vector<int>&)>& reload = nullptr,
                        wspr_trace* trace = nullptr, HashBatch* hb = nullptr, int hb_off = 0);
    int decode_core(int nseg, int samples, const decoder_options& opt, decoder_results* out, int max_results,
                    int* n_results, const std::vector<int>& active0, unsigned fast, PendingFano& pend,
                    const FanoMemo* memo = nullptr, wspr_trace* trace = nullptr, HashBatch* hb = nullptr, int hb_off = 0);
    // a later round of a usehashtable batch: the listed segments (rows already restored) once more, everything else
    // of the working buffers and of out / n_results left alone
    int decode_again(int nseg, int samples, const decoder_options& opt, decoder_results* out, int max_results,
                     int* n_results, const std::vector<int>& segs, HashBatch* hb, int hb_off);
    int last_timings(double* ms, int cap);
    int bench_fft_sync(int nseg, int samples, int iters, double* ms);
    int bench_valu(int nseg, int samples, int iters, double* ms);

    void demod_single(float* id, float* qd, long np, unsigned char* symbols, float* freq, int ifmin, int ifmax,
                      float fstep, int* shift, int lagmin, int lagmax, int lagstep, float* drift, float* sync,
                      int mode, int symfac = 50);
    void subtract_single(float* id, float* qd, long np, float f0, int shift, float drift, const unsigned char* sym);
    void subtract_symbolwise_single(float* id, float* qd, long np, float f0, int shift, float drift,
                                    const unsigned char* sym);
    // the wave-parallel search (k6_fano_wave.hip) over host vectors; steps (may be null): expansion steps per vector
    int fano_batch(const unsigned char* symbols, int n, unsigned maxcycles, int* ret, unsigned* cycles,
                   unsigned* metric, unsigned* maxnp, unsigned char* data, unsigned* steps = nullptr);
    int fano_resident(const unsigned char* d_symbols, const int* h_offsets, int n, unsigned maxcycles, int* ret,
                      unsigned* cycles, unsigned char* data);
    int bench_decimate(const void* d_raw, size_t bytes_per_seg, int nseg, float* dI, float* dQ, int iters, double* ms);
    int decimate_device(const void* d_raw, size_t bytes_per_seg, int nseg, float* dI, float* dQ, int normalise,
                        int* h_nout, DecimState* d_states = nullptr);
    int decimate_stream(DecimState* h_state, const uint8_t* iq, size_t nbytes, float* I, float* Q, uint32_t fill,
                        uint32_t cap, uint32_t* new_fill);
    // one chunk of EACH of n receivers' streams (all of nbytes): one transfer and one launch set for all of them
    int decimate_stream_many(DecimState* const* h_states, const uint8_t* const* iq, size_t nbytes, int n, float* const* I,
                             float* const* Q, const uint32_t* fill, uint32_t cap, uint32_t* new_fill);

    struct Impl;
    struct DecodeRun;               // state of one decode_core() call (wspr_pipeline.hip)
    std::unique_ptr<Impl> d;

private:
    explicit Context(int nslots);
};

}  // namespace wspr

// K6w -- Fano search on the device, one wavefront per soft-symbol vector, 64 tree visits per step.
//
// Replaces (for the decoder's long tail) reference wsprd/fano.c:87-238 + wsprd_utils.c:196-213.
// The algorithm and its proof sketch are in fano_wave.h: the reference's serial walk is the
// pre-order traversal of a tree of (node, threshold) visits; this kernel expands that tree 64
// visits at a time from a stack kept in walk order, with per-visit ledgers that make the
// reference's cycle count come out exactly.  An undecodable vector (810 000 cycles in the
// reference, ~5 ms of a CPU core; a serial walk on one GPU lane takes 0.5 s)
// takes ~7 400 steps here (10 ms alone; 6 000 of them together: 18 ms).
//
// Per step (one wave, no divergence outside the three predicated store slots):
//   pop     lanes read the n <= 64 earliest visits (structure-of-arrays stack, conflict-free)
//   expand  fano_wave::expand(): parity of two 32-bit masks, two branch metrics, three compares
//   cut     a completed frame drops everything later in walk order
//   scan    output slots by ballot/mbcnt; ledger flow by one DPP prefix sum + one bpermute
//   push    children written back in walk order (top of the stack = earliest)
// Integer work, latency bound.  The stack is a per-wave slice of device memory (4 096 visits) whose TOP lives in LDS:
// a step pops at most 64 visits and pushes at most 192, all within a few hundred entries of the top, so a circular
// window of 512 entries (10 KB of LDS per wave, index & 511) serves every access of a step at LDS latency; when the
// top moves out of the window, half a window (256 entries, 5 KB) is spilled to / filled from the wave's slice in one
// coalesced burst.  (Round 2 kept the whole stack in device memory: two dependent memory round trips per step,
// ~1.3 us, 10 ms for a full time-out; or 1 024 visits in LDS: 21 KB per wave, 7 waves per CU.)
#include "wspr_device.h"
#include "fano_wave.h"
#include <cstdlib>
#include <algorithm>

namespace wspr {
namespace {

using namespace fano_wave;

// pending visits per wave: template parameter kCap (a serial walk needs < 100, 64-wide steps ~1700 at most in
// tests -- the step narrows when the stack fills, see below); 20 bytes of LDS each

// inclusive prefix sum over the 64 lanes (DPP row shifts + row broadcasts, gfx9 idiom)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
#define WSPR_DPP(x, ctrl, rmask, bmask) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, rmask, bmask, false)
    v += WSPR_DPP(v, 0x111, 0xf, 0xf);        // row_shr:1
    v += WSPR_DPP(v, 0x112, 0xf, 0xf);        // row_shr:2
    v += WSPR_DPP(v, 0x114, 0xf, 0xe);        // row_shr:4  (banks 1-3)
    v += WSPR_DPP(v, 0x118, 0xf, 0xc);        // row_shr:8  (banks 2-3)
    v += WSPR_DPP(v, 0x142, 0xa, 0xf);        // row_bcast:15 -> rows 1 and 3
    v += WSPR_DPP(v, 0x143, 0xc, 0xf);        // row_bcast:31 -> rows 2 and 3
#undef WSPR_DPP
    return v;
}

__device__ __forceinline__ int lanes_below(unsigned long long mask) {   // popcount(mask & lanes below me)
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

constexpr int kWin = 512, kHalf = kWin / 2;                 // LDS window over the top of the stack; see the invariants below
static_assert(kHalf >= 64 && kWin - 192 >= kHalf, "a step's pops (<= 64) and pushes (<= 192) must fit beside the half that moves");

template <int kCap>
__global__ __launch_bounds__(64)
void fano_wave_kernel(const unsigned char* __restrict__ symbols, const int* __restrict__ offsets, int n,
                      const short* __restrict__ metric0, unsigned maxcycles,
                      int* __restrict__ ret, unsigned* __restrict__ cycles, unsigned* __restrict__ metric,
                      unsigned* __restrict__ maxnp, unsigned char* __restrict__ data, unsigned* __restrict__ steps_out,
                      uint32_t* __restrict__ gpool, int* __restrict__ next_vector) {
    // entry i of the stack is in LDS slot i & (kWin - 1) while lo <= i < lo + kWin (lo: wave-uniform, a multiple of
    // kHalf); entries below lo are in the wave's slice of device memory gp[arr][i]
    __shared__ uint32_t lpool[5 * kWin];
    uint32_t* __restrict__ gp = gpool + (size_t)blockIdx.x * 5 * kCap;
    auto ld = [&](int arr, int i) -> uint32_t { return lpool[arr * kWin + (i & (kWin - 1))]; };
    auto st_ = [&](int arr, int i, uint32_t v) { lpool[arr * kWin + (i & (kWin - 1))] = v; };
    auto lds_order = [] {        // one wave: its LDS operations complete in order; only the compiler must not move them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    __shared__ uint2 bm[kBits];
    __shared__ unsigned char symd[kNSymD];
    __shared__ short mt[256];
    const int lane = threadIdx.x;
    for (int e = lane; e < 256; e += 64) mt[e] = metric0[e];
    // persistent grid: a wave takes the next unclaimed vector until none is left (the grid -- and with it the
    // device memory for the pending-visit stores -- is bounded whatever the number of vectors; quick decodes
    // and full time-outs balance themselves)
    for (;;) {
    int v;
    if (next_vector) {
        int got = 0;
        if (lane == 0) got = atomicAdd(next_vector, 1);
        v = __builtin_amdgcn_readfirstlane(got);
    } else {
        v = blockIdx.x;
    }
    if (v >= n) return;
    __syncthreads();                                        // the previous vector's LDS tables are no longer read

    // ---- branch metrics of the 81 nodes (fano.c:118-124), de-interleaving on the fly ------------
    {
        const unsigned char* __restrict__ sym = symbols + (size_t)offsets[v] * kNSymD;
        int base = 0;
        for (int c = 0; c < 4; ++c) {                       // p-th reversed counter value below 162 (wsprd_utils.c:196-213)
            const int i = 64 * c + lane;
            const int j = (int)(__brev((unsigned)i) >> 24);
            const bool valid = j < kNSymD;
            const unsigned long long m = __ballot(valid);
            if (valid) symd[base + lanes_below(m)] = sym[j];
            base += __popcll(m);
        }
    }
    __syncthreads();
    for (int k = lane; k < kBits; k += 64) {
        const int s0 = symd[2 * k], s1 = symd[2 * k + 1];
        const int a0 = mt[s0], a1 = mt[255 - s0], b0 = mt[s1], b1 = mt[255 - s1];   // "sent 1" row = mirrored table
        bm[k] = make_uint2((uint32_t)(uint16_t)(short)(a0 + b0) | ((uint32_t)(uint16_t)(short)(a0 + b1) << 16),
                           (uint32_t)(uint16_t)(short)(a1 + b0) | ((uint32_t)(uint16_t)(short)(a1 + b1) << 16));
    }
    if (lane == 0) {                                        // the root, visited at t = 0, -60, -120, ...
        st_(0, 0, 0u); st_(1, 0, 0u); st_(2, 0, pack_meta(0u, 0, true)); st_(3, 0, pack_gt(0, 0)); st_(4, 0, 0u);
    }
    __syncthreads();

    const unsigned budget = maxcycles * (unsigned)kBits;
    int size = 1;                    // wave-uniform
    int lo = 0;                      // wave-uniform: first stack index held in LDS
    unsigned settled = 0;            // looks before the earliest pending visit: final
    unsigned steps = 0;
    int rc = -2;
    unsigned out_cycles = 0, out_metric = 0;
    uint32_t out_dlo = 0, out_dhi = 0;

    for (;;) {
        if (settled >= budget) { rc = -1; out_cycles = budget + 2; break; }      // fano.c:149, 234-237
        // visits taken this step: all 64 lanes while the stack is at most half full, then 8, then 1 -- a
        // narrower step walks more depth-first and a serial walk holds < 100 pending visits (a lane adds at
        // most 2 net)
        const int room = kCap - size;
        const int wide = size <= kCap / 2 ? 64 : (room > 256 ? 8 : 1);
        const int take = min(min(wide, size), room >> 1);
        if (take < 1 || steps > 4u * budget + 1024u) { rc = -2; break; }
        ++steps;
        // the visits popped now are [size - take, size): slide the window down when they start below it.  The slots
        // the fill overwrites held [lo + kHalf, lo + kWin), all dead: size - take < lo and take <= 64 give size < lo + kHalf.
        if (size - take < lo) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // an earlier spill's stores (other lanes') have landed
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            lo -= kHalf;
#pragma unroll
            for (int arr = 0; arr < 5; ++arr)
#pragma unroll
                for (int u = 0; u < kHalf / 64; ++u) {
                    const int i = lo + 64 * u + lane;
                    lpool[arr * kWin + (i & (kWin - 1))] = gp[arr * kCap + i];
                }
            lds_order();
        }
        const bool active = lane < take;
        const int idx = size - 1 - (active ? lane : 0);
        Visit x{ld(0, idx), ld(1, idx), ld(2, idx), ld(3, idx), ld(4, idx)};
        const int pos = v_pos(x);
        const bool is_done_visit = active && pos == kPosDone;
        if (__builtin_amdgcn_readfirstlane((int)is_done_visit)) {                // earliest pending visit completes the frame
            const unsigned looks = settled + 1;
            rc = looks >= budget ? -1 : 0;                                       // fano.c:234-237
            out_cycles = looks + 1;
            out_metric = (unsigned)__builtin_amdgcn_readfirstlane(v_gamma(x));
            out_dlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)x.dlo);
            out_dhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(x.meta & 0x3ffffu));
            break;
        }
        const uint2 q = bm[min(pos, kLast)];
        Expansion e;
        expand(x, q.x, q.y, e);
        // a completed frame (new, or found earlier and still waiting): nothing later in walk order matters
        const bool stop_here = active && (is_done_visit || e.done);
        const unsigned long long stop_mask = __ballot(stop_here);
        bool live = active;
        int base = size - take;
        if (stop_mask) {
            const int jc = __builtin_ctzll(stop_mask);
            live = active && lane <= jc;
            base = 0;
            lo = 0;                                                  // the stack restarts at its bottom: so does the window
        }
        const bool keep_self = live && (is_done_visit || e.done);
        const bool has0 = live && !keep_self && e.has0;
        const bool has1 = live && !keep_self && e.has1;
        const bool again = live && (keep_self || e.again);
        const int c = (int)has0 + (int)has1 + (int)again;
        if (__ballot(again && !keep_self && v_thr(e.self) < -32000)) { rc = -2; break; }   // 16-bit threshold field
        // output slots in walk order
        const unsigned long long b0 = __ballot(has0), b1 = __ballot(has1), b2 = __ballot(again);
        const int before = lanes_below(b0) + lanes_below(b1) + lanes_below(b2);
        const int total = __popcll(b0) + __popcll(b1) + __popcll(b2);
        // ledger flow: a visit's first look precedes its children; with no children its ledger moves on too
        const uint32_t back = (live && !keep_self) ? 1u + (c == 0 ? x.led : 0u) : 0u;
        const uint32_t pre = wave_scan_add(back);
        const unsigned long long has_mask = b0 | b1 | b2;
        const unsigned long long later = has_mask & ~((2ull << lane) - 1ull);      // lanes after me with output
        const int nxt = later ? __builtin_ctzll(later) : 63;
        const uint32_t pre_nxt = (uint32_t)__builtin_amdgcn_ds_bpermute(nxt << 2, (int)pre);
        const uint32_t gain = pre_nxt - pre;                                      // looks between my last output and the next one
        const int first = has_mask ? __builtin_ctzll(has_mask) : 63;
        settled += (uint32_t)__builtin_amdgcn_readlane((int)pre, first);
        // the visits pushed now are [base, base + total), total <= 192: slide the window up when they end above it.  The
        // half that is spilled, [lo, lo + kHalf), lies below base: base + total > lo + kWin gives base > lo + kWin - 192.
        if (base + total > lo + kWin) {
#pragma unroll
            for (int arr = 0; arr < 5; ++arr)
#pragma unroll
                for (int u = 0; u < kHalf / 64; ++u) {
                    const int i = lo + 64 * u + lane;
                    gp[arr * kCap + i] = lpool[arr * kWin + (i & (kWin - 1))];
                }
            lo += kHalf;
            lds_order();                                             // the slots are read before the pushes below overwrite them
        }
        // push
        const int top = base + total - 1;
        const uint32_t tail_led = x.led + gain;
        if (has0) {
            const int w = top - before;
            st_(0, w, e.kid0.st); st_(1, w, e.kid0.dlo); st_(2, w, e.kid0.meta); st_(3, w, e.kid0.gt);
            st_(4, w, e.look1 + ((!has1 && !again) ? tail_led : 0u));
        }
        if (has1) {
            const int w = top - before - (int)has0;
            st_(0, w, e.kid1.st); st_(1, w, e.kid1.dlo); st_(2, w, e.kid1.meta); st_(3, w, e.kid1.gt);
            st_(4, w, !again ? tail_led : 0u);
        }
        if (again) {
            const int w = top - before - (int)has0 - (int)has1;
            const Visit s = is_done_visit ? x : e.self;
            st_(0, w, s.st); st_(1, w, s.dlo); st_(2, w, s.meta); st_(3, w, s.gt);
            st_(4, w, keep_self ? 0u : tail_led);
        }
        size = base + total;
        lds_order();
    }
    if (lane == 0) {
        ret[v] = rc;
        cycles[v] = out_cycles;
        if (metric) metric[v] = rc == 0 ? out_metric : 0u;
        if (maxnp) maxnp[v] = rc == 0 ? (unsigned)kLast : 0u;
        unsigned char d[10];
        decisions_to_bytes(out_dlo, out_dhi, d);
        for (int k = 0; k < 10; ++k) data[(size_t)v * 10 + k] = d[k];
        if (steps_out) steps_out[v] = steps;
    }
    if (!next_vector) return;
    }
}

}  // namespace

// resident grid: 13 single-wave workgroups per CU by LDS (11.5 KB each); a persistent loop takes the vectors
#ifndef WSPR_K6W_PER_CU
#define WSPR_K6W_PER_CU 13
#endif
constexpr int kWaveGrid = 256 * WSPR_K6W_PER_CU;

size_t fano_wave_scratch_words(int n) {                     // the waves' stack slices + the work counter
    return (size_t)std::min(n, kWaveGrid) * 5 * 4096 + 16;
}

void launch_fano_wave(const unsigned char* symbols, const int* offsets, int n, const short* metric0,
                      unsigned maxcycles, int* ret, unsigned* cycles, unsigned* metric, unsigned* maxnp,
                      unsigned char* data, unsigned* steps, uint32_t* scratch, hipStream_t st) {
    if (n <= 0) return;
    const int grid = std::min(n, kWaveGrid);
    int* counter = reinterpret_cast<int*>(scratch + (size_t)grid * 5 * 4096);
    static const int rep = [] { const char* e = lab_env("WSPR_REPEAT_FANO"); return e ? atoi(e) : 1; }();
    for (int r = 1; r < rep; ++r) {                       // measurement hook: the same launch again (same outputs)
        (void)hipMemsetAsync(counter, 0, sizeof(int), st);
        hipLaunchKernelGGL((fano_wave_kernel<4096>), dim3(grid), dim3(64), 0, st, symbols, offsets, n, metric0, maxcycles,
                           ret, cycles, metric, maxnp, data, steps, scratch, counter);
    }
    (void)hipMemsetAsync(counter, 0, sizeof(int), st);
    // WSPR_FANO_WAVE_CAP=1024 (test hook, tests/test_gpu_parity.py): a stack of 1 024 visits instead of 4 096, so that the
    // narrowed steps (8 visits, then 1, as the stack fills), the window's spills and fills at those widths and the
    // overflow exit (-2: the caller's host routine takes the vector) are exercised by ordinary time-out vectors
    static const bool small = [] { const char* e = lab_env("WSPR_FANO_WAVE_CAP"); return e && atoi(e) == 1024; }();
    if (small)
        hipLaunchKernelGGL((fano_wave_kernel<1024>), dim3(grid), dim3(64), 0, st, symbols, offsets, n, metric0, maxcycles,
                           ret, cycles, metric, maxnp, data, steps, scratch, counter);
    else
        hipLaunchKernelGGL((fano_wave_kernel<4096>), dim3(grid), dim3(64), 0, st, symbols, offsets, n, metric0, maxcycles,
                           ret, cycles, metric, maxnp, data, steps, scratch, counter);
}

}  // namespace wspr

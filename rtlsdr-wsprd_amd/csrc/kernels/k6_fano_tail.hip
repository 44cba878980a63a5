// K6 -- Fano search on the device, used for the long tail of the host decoder.
//
// The host Fano pool (north star) decodes every soft-symbol vector with a small cycle budget;
// >98 % of the vectors that decode at all do so within it.  What is left are vectors that will
// almost surely time out after 810 000 cycles (wsprd/fano.c:149-153 with the decoder's
// maxcycles = 10 000, wsprd.c:431) -- milliseconds of one CPU core each, and the dominant host cost
// on crowded bands.  Those are finished here, thousands at a time, off the critical path.
//
// One lane per vector, the storage-free search of fano_stateless.h (bit-identical to the host
// routine; tests/test_fano_stateless.py) as a wave-friendly state machine.  The only per-lane
// memory is the 81 x 4 branch-metric table, kept in LDS as one 8-byte word per node and laid out
// [node][lane] so that a wave's accesses never conflict, whatever node each lane is at.
// Integer work, latency-bound by construction (a serial tree walk); throughput comes from running
// 3 waves per CU x 256 CUs = 49 152 vectors at once.
#include "wspr_device.h"
#include "fano_stateless.h"

namespace wspr {
namespace {

// ---- the same search as fano_stateless(), reshaped for a SIMT wave -----------------------
// fano_stateless() nests a "look backward" loop inside the iteration loop; on a wave, lanes with
// different back-off depths would serialise each other.  Here every trip of ONE loop is a
// micro-step: either the start of an iteration (forward move, or the decision to back off) or one
// backward probe, so all lanes advance every trip.  The reference's iteration counter only ticks
// when an iteration completes.  Path bits live in three 32-bit registers per bitset, the metrics
// of the two nodes a step can move to are fetched from LDS at the top of the step.
// Path history as shift registers relative to the current node: bit j of `hd` is the hypothesised
// bit of node pos-1-j, bit j of `hp` its branch index.  A forward move shifts a bit in, a backward
// move shifts one out, so nothing is ever indexed by a per-lane variable; and the low word of `hd`
// IS the encoder state of the previous node (state(q) = sum_j D[q-j] << j).
struct Hist96 {
    uint32_t w0, w1, w2;
    __device__ __forceinline__ void push(unsigned b) {
        w2 = (w2 << 1) | (w1 >> 31);
        w1 = (w1 << 1) | (w0 >> 31);
        w0 = (w0 << 1) | b;
    }
    __device__ __forceinline__ void pop() {
        w0 = (w0 >> 1) | (w1 << 31);
        w1 = (w1 >> 1) | (w2 << 31);
        w2 >>= 1;
    }
    __device__ __forceinline__ unsigned bit(int j) const {          // j is a compile-time constant at the call sites
        return ((j < 32 ? w0 : (j < 64 ? w1 : w2)) >> (j & 31)) & 1u;
    }
};

__device__ __forceinline__ int metric_of(const uint2 q, unsigned pair) {
    const uint32_t w = (pair & 2u) ? q.y : q.x;
    return (int)(short)((pair & 1u) ? (w >> 16) : (w & 0xffffu));
}

struct RankedD { int best, second; unsigned better; };
__device__ __forceinline__ RankedD rank_d(const uint2 q, uint32_t st0, bool in_tail) {
    const unsigned zp = fano_detail::pair_of(st0);
    const int m0 = metric_of(q, zp), m1 = metric_of(q, 3u ^ zp);
    RankedD r;
    const bool zero_better = in_tail || (m0 > m1);
    r.best = zero_better ? m0 : m1;
    r.second = in_tail ? m0 : (zero_better ? m1 : m0);
    r.better = zero_better ? 0u : 1u;
    return r;
}

__device__ __forceinline__ void fano_lane(const uint2* __restrict__ bm /* LDS [node][64] */, int lane, int delta,
                                          unsigned maxcycles, FanoResult& out) {
    constexpr int nbits = kNBitsD, last = nbits - 1, tail = nbits - 31;
    Hist96 hd{0, 0, 0}, hp{0, 0, 0};
    int pos = 0, deepest = 0, t = 0, g = 0;
    uint32_t st = 0;                        // encoder state of the current node incl. its hypothesised bit
    RankedD rk = rank_d(bm[lane], 0u, false);
    st |= rk.better;
    unsigned pk = 0;
    const unsigned budget = maxcycles * (unsigned)nbits;
    unsigned it = 1;
    bool backing = false, done = false;
    while (!done) {
        // metrics of the two nodes this step may move to
        const uint2 q_next = bm[min(pos + 1, last) * 64 + lane];
        const uint2 q_prev = bm[max(pos - 1, 0) * 64 + lane];
        if (!backing) {
            if (it > budget) break;                                   // time-out
            deepest = max(deepest, pos);
            const int ahead = g + (pk ? rk.second : rk.best);
            if (ahead >= t) {
                if (g < t + delta && ahead >= t + delta)
                    t += delta * (((ahead - t) * 34953) >> 21);        // floor((ahead-t)/60), exact below 2^15
                hd.push(st & 1u);
                hp.push(pk);
                g = ahead;
                st <<= 1;
                ++pos;
                if (pos == last + 1) { done = true; }
                else {
                    rk = rank_d(q_next, st, pos >= tail);
                    st |= rk.better;
                    pk = 0;
                    ++it;
                }
            } else {
                backing = true;
            }
        }
        if (backing && !done) {
            const int q = pos - 1;
            const uint32_t pst = hd.w0;                               // state of node q with its current bit
            const unsigned ppk = hp.w0 & 1u;
            const RankedD prk = rank_d(q_prev, pst & ~1u, q >= tail);
            const int pg = g - (ppk ? prk.second : prk.best);
            if (pos == 0 || pg < t) {
                t -= delta;
                st ^= pk;                                             // back to the better branch
                pk = 0;
                backing = false;
                ++it;
            } else {
                pos = q; st = pst; rk = prk; pk = ppk; g = pg;
                hd.pop();
                hp.pop();
                if (pos < tail && pk != 1) { pk = 1; st ^= 1u; backing = false; ++it; }
            }
        }
    }
    out.maxnp = (unsigned)deepest;
    out.metric = (unsigned)g;
    // decoded bytes (only meaningful after a full-length path, pos == 81): D[k] = hd bit (80 - k)
#pragma unroll
    for (int k = 0; k < (nbits >> 3); ++k) {
        unsigned b = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) b = (b << 1) | hd.bit(80 - (8 * k + j));
        out.data[k] = (unsigned char)b;
    }
    out.cycles = it + 1;
    out.ret = (it >= budget) ? -1 : 0;
}

__global__ __launch_bounds__(64)
void fano_tail_kernel(const unsigned char* __restrict__ symbols, const int* __restrict__ offsets, int n,
                      const short* __restrict__ metric0, int delta, unsigned maxcycles,
                      int* __restrict__ ret, unsigned* __restrict__ cycles, unsigned* __restrict__ metric,
                      unsigned* __restrict__ maxnp, unsigned char* __restrict__ data) {
    __shared__ uint2 bm[kNBitsD * 64];
    __shared__ short mt[256];
    const int lane = threadIdx.x;
    const int idx = blockIdx.x * 64 + lane;
    for (int e = lane; e < 256; e += 64) mt[e] = metric0[e];
    __syncthreads();
    if (idx < n) {
        const unsigned char* __restrict__ sym = symbols + (size_t)offsets[idx] * kNSymD;
        // de-interleave on the fly: position p holds symbol bitrev8(i_p), the p-th reversed counter
        // value below 162 (wsprd_utils.c:196-213)
        int p = 0;
        unsigned char s0 = 0;
        for (int i = 0; i < 256 && p < kNSymD; ++i) {
            const int j = (int)(__brev((unsigned)i) >> 24);
            if (j < kNSymD) {
                const unsigned char v = sym[j];
                if (p & 1) {
                    // branch metrics of node p/2 (fano.c:118-124); "sent 1" row = mirrored table
                    const int a0 = mt[s0], a1 = mt[255 - s0], b0 = mt[v], b1 = mt[255 - v];
                    const uint32_t lo = (uint32_t)(uint16_t)(short)(a0 + b0) | ((uint32_t)(uint16_t)(short)(a0 + b1) << 16);
                    const uint32_t hi = (uint32_t)(uint16_t)(short)(a1 + b0) | ((uint32_t)(uint16_t)(short)(a1 + b1) << 16);
                    bm[(p >> 1) * 64 + lane] = make_uint2(lo, hi);
                } else {
                    s0 = v;
                }
                ++p;
            }
        }
        FanoResult r;
        fano_lane(bm, lane, delta, maxcycles, r);
        ret[idx] = r.ret;
        cycles[idx] = r.cycles;
        if (metric) metric[idx] = r.metric;
        if (maxnp) maxnp[idx] = r.maxnp;
        for (int k = 0; k < 10; ++k) data[(size_t)idx * 10 + k] = r.data[k];
    }
}
}  // namespace

void launch_fano_tail(const unsigned char* symbols, const int* offsets, int n, const short* metric0, int delta,
                      unsigned maxcycles, int* ret, unsigned* cycles, unsigned* metric, unsigned* maxnp,
                      unsigned char* data, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(fano_tail_kernel, dim3((n + 63) / 64), dim3(64), 0, st, symbols, offsets, n, metric0, delta,
                       maxcycles, ret, cycles, metric, maxnp, data);
}

}  // namespace wspr

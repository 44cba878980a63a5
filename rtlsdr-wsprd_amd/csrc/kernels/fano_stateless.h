// Fano sequential decoder without per-node storage (host + device).
//
// Same search as the reference decoder (wsprd/fano.c:87-238) and as the host routine
// wspr::fano_decode(): identical move order, threshold schedule, cycle count, path metric, maxnp
// and decoded bytes.  The reference keeps, for every tree node, the encoder state, the path
// metric, the two sorted branch metrics and the branch index.  All of that is a function of
//   * the hypothesised bit of every node on the current path  (bitset D, 81 bits),
//   * which of its two branches each node is exploring          (bitset P, 81 bits),
//   * the current node's own state, kept in a handful of scalars,
// because  state(k) = sum_m D[m] << (k - m)  and  gamma(k-1) = gamma(k) - metric(k-1, P[k-1]).
// Moving backwards therefore recomputes the previous node from the bitsets instead of loading it.
// On the GPU this turns ~1.5 KB of randomly indexed per-lane arrays into two 64-bit register
// pairs plus one 8-byte branch-metric fetch per visited node.
//
// `Metrics` is a callable  m(pos) -> 4 branch metrics of node pos  (index = transmitted symbol
// pair 0..3), i.e. the table the reference fills at fano.c:118-124.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define WSPR_FANO_HD __host__ __device__ __forceinline__
#define WSPR_FANO_MEM __host__ __device__ __forceinline__
#else
#define WSPR_FANO_HD static inline
#define WSPR_FANO_MEM inline
#endif

namespace wspr {

struct FanoResult {
    int      ret;        // 0 = decoded, -1 = cycle budget exhausted (fano.c:234-237)
    unsigned metric;     // final path metric
    unsigned cycles;     // iterations + 1, as the reference reports them
    unsigned maxnp;      // deepest node reached
    unsigned char data[10];
};

struct Metric4 { short m[4]; };

namespace fano_detail {

constexpr uint32_t kPolyA = 0xf2d05351u, kPolyB = 0xe4613c47u;   // fano.c:51-52

WSPR_FANO_HD unsigned parity32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (unsigned)__popc(v) & 1u;
#else
    return (unsigned)__builtin_popcount(v) & 1u;
#endif
}
WSPR_FANO_HD unsigned pair_of(uint32_t state) { return (parity32(state & kPolyA) << 1) | parity32(state & kPolyB); }

struct Bits128 {
    uint64_t lo, hi;
    WSPR_FANO_MEM unsigned get(int k) const { return (unsigned)(((k & 64) ? hi : lo) >> (k & 63)) & 1u; }
    WSPR_FANO_MEM void put(int k, unsigned b) {
        const uint64_t m = 1ull << (k & 63);
        if (k & 64) hi = (hi & ~m) | (b ? m : 0ull);
        else        lo = (lo & ~m) | (b ? m : 0ull);
    }
};

// sorted branch metrics of a node whose encoder state (low bit cleared) is st0
struct Ranked { int best, second; unsigned better_bit; };
WSPR_FANO_HD Ranked rank_node(const Metric4& bm, uint32_t st0, bool in_tail) {
    const unsigned zp = pair_of(st0);
    const int m0 = bm.m[zp];
    Ranked r;
    if (in_tail) { r.best = m0; r.second = m0; r.better_bit = 0; return r; }
    const int m1 = bm.m[3 ^ zp];
    if (m0 > m1) { r.best = m0; r.second = m1; r.better_bit = 0; }
    else         { r.best = m1; r.second = m0; r.better_bit = 1; }
    return r;
}

}  // namespace fano_detail

template <class Metrics>
WSPR_FANO_HD void fano_stateless(const Metrics& metrics, unsigned nbits, int delta, unsigned maxcycles,
                                 FanoResult& out) {
    using namespace fano_detail;
    const int last = (int)nbits - 1, tail = (int)nbits - 31;
    Bits128 D{0, 0}, P{0, 0};        // decisions / explored-branch index of the nodes before `pos`

    int pos = 0, deepest = 0, t = 0, g = 0;
    uint32_t st = 0;                 // encoder state of node `pos` incl. its hypothesised bit
    Ranked rk = rank_node(metrics(0), 0u, false);
    st |= rk.better_bit;
    unsigned pk = 0;                 // branch index of node `pos`
    const unsigned budget = maxcycles * nbits;
    unsigned it;
    for (it = 1; it <= budget; ++it) {
        if (pos > deepest) deepest = pos;
        const int ahead = g + (pk ? rk.second : rk.best);
        if (ahead >= t) {
            if (g < t + delta)
                while (ahead >= t + delta) t += delta;
            D.put(pos, st & 1u);
            P.put(pos, pk);
            g = ahead;
            st <<= 1;
            if (++pos == last + 1) break;
            rk = rank_node(metrics(pos), st, pos >= tail);
            st |= rk.better_bit;
            pk = 0;
            continue;
        }
        for (;;) {
            // the node before `pos`, rebuilt from the bitsets
            bool stay = (pos == 0);
            uint32_t pst = 0;
            Ranked prk{0, 0, 0};
            unsigned ppk = 0;
            int pg = 0;
            if (!stay) {
                const int q = pos - 1;
                pst = (st >> 1) | ((q >= 31 ? D.get(q - 31) : 0u) << 31);
                ppk = P.get(q);
                prk = rank_node(metrics(q), pst & ~1u, q >= tail);
                pg = g - (ppk ? prk.second : prk.best);
                stay = pg < t;
            }
            if (stay) {
                t -= delta;
                if (pk != 0) { pk = 0; st ^= 1u; }
                break;
            }
            --pos;
            st = pst; rk = prk; pk = ppk; g = pg;
            if (pos < tail && pk != 1) { pk = 1; st ^= 1u; break; }
        }
    }
    out.maxnp = (unsigned)deepest;
    out.metric = (unsigned)g;
    // data byte k = encoder state of node 7 + 8k = decisions 8k .. 8k+7, first decision in the MSB
    // (a time-out returns whatever the current path holds for the nodes it covers, like the
    // reference returns its node array; only decoded frames are ever used)
    if (pos <= last) { D.put(pos, st & 1u); }
    for (unsigned k = 0; k < (nbits >> 3); ++k) {
        unsigned b = 0;
        for (int j = 0; j < 8; ++j) b = (b << 1) | D.get((int)(8 * k) + j);
        out.data[k] = (unsigned char)b;
    }
    out.cycles = it + 1;
    out.ret = (it >= budget) ? -1 : 0;
}

}  // namespace wspr

// Fano sequential decoder as a wave-parallel search (host + device).
//
// Same result as the reference decoder (wsprd/fano.c:87-238) and as wspr::fano_decode(): return
// code, cycle count and decoded bytes -- but the walk is not serial.
//
// What the reference's loop does, seen from above.  One "cycle" of fano.c is one forward look from
// a node along one of its two branches (fano.c:154-213: the look either moves forward or starts a
// chain of free backward moves).  Unrolling the move rules gives a recursion: visit(node, t) with
// threshold t <= gamma(node)
//     look at the better branch:   ahead = gamma + metric;  fails (ahead < t) -> back to the parent
//     else enter the child.  If this is a first visit (gamma < t + delta, fano.c:158) the threshold
//          is first tightened to the largest grid value <= ahead; branch metrics are <= +10, so that is
//          at most ONE step of delta = 60 (ahead < t + 70).  The child's subtree is searched at the
//          tightened threshold and, when that fails, again at t (the relaxations of fano.c:195-201
//          happen at the child because its parent's gamma is below the tightened threshold)
//     then the same for the other branch (not in the tail, fano.c:206), then back to the parent;
// and the root is visited with t = 0, -60, -120, ... for ever.  The reference's trajectory is the
// pre-order walk of this tree of (node, threshold) visits, its cycle counter the number of looks
// so far, and it stops at the first visit that moves forward from the last node.
//
// Subtrees of that tree are independent of each other (a visit needs only the node's encoder state,
// gamma and t), so they can be expanded in any order, 64 at a time.  To keep the cycle count
// exact, the pending visits are held in a stack sorted by their position in the pre-order walk
// (top = earliest), a step pops the W earliest, expands them and pushes their children back in
// order, and every pending visit carries a ledger: the number of looks already performed that
// fall, in walk order, after it and before the next pending visit.  Looks that fall before the
// earliest pending visit are final ("settled"): when they reach the cycle budget the search has
// timed out, exactly; when a successful visit becomes the earliest pending one, settled + 1 is
// exactly the reference's cycle count.  Visits after a found success are dropped.
//
// A time-out of 810 000 cycles takes about 8 700 steps of 64 visits instead of 810 000 serial
// iterations (tests/test_fano_wave.py; the serial form is the host routine in wspr_message.cpp).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define WSPR_FW_HD __host__ __device__ __forceinline__
#else
#define WSPR_FW_HD static inline
#endif

namespace wspr {
namespace fano_wave {

constexpr int kBits = 81, kLast = 80, kTail = 81 - 31, kDelta = 60;
constexpr uint32_t kPolyA = 0xf2d05351u, kPolyB = 0xe4613c47u;      // fano.c:51-52
constexpr int kPosDone = 127;                                         // marks a successful visit

// One pending visit, 5 words.
//   st   encoder state of the node, hypothesised bit not yet set (bit 0 = 0)
//   dlo  decisions of nodes 0..31 (bit k = node k)
//   meta bits 0..17 decisions of nodes 32..49, bits 18..24 node index, bit 25 "retry": when this
//        visit fails, the node is visited again one threshold step lower
//   gt   gamma (low 16 bits, signed) | threshold (high 16 bits, signed)
//   led  looks already performed between this visit and the next pending one
struct Visit { uint32_t st, dlo, meta, gt, led; };

WSPR_FW_HD int      v_pos(const Visit& v)   { return (int)((v.meta >> 18) & 127u); }
WSPR_FW_HD bool     v_retry(const Visit& v) { return (v.meta >> 25) & 1u; }
WSPR_FW_HD int      v_gamma(const Visit& v) { return (int)(int16_t)(v.gt & 0xffffu); }
WSPR_FW_HD int      v_thr(const Visit& v)   { return (int)(int16_t)(v.gt >> 16); }
WSPR_FW_HD uint32_t pack_gt(int g, int t)   { return ((uint32_t)g & 0xffffu) | ((uint32_t)t << 16); }
WSPR_FW_HD uint32_t pack_meta(uint32_t dhi, int pos, bool retry) {
    return (dhi & 0x3ffffu) | ((uint32_t)pos << 18) | ((uint32_t)retry << 25);
}

WSPR_FW_HD unsigned parity32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (unsigned)__popc(v) & 1u;
#else
    return (unsigned)__builtin_popcount(v) & 1u;
#endif
}

// What expanding one visit yields.  Children in walk order: kid[0] (better branch), kid[1] (other
// branch), then the visit itself one step lower (retry) -- each only if its flag is set.
struct Expansion {
    bool has0, has1, again, done;   // done: this visit completes the frame (no children, no looks counted here)
    uint32_t look1;                 // 1 if the second branch is looked at (after kid[0]'s subtree)
    Visit kid0, kid1, self;
};

// bm4: the node's four branch metrics, index = symbol pair, as two packed words (lo: pairs 0,1; hi: 2,3)
WSPR_FW_HD int metric_of(uint32_t lo, uint32_t hi, unsigned pair) {
    const uint32_t w = (pair & 2u) ? hi : lo;
    return (int)(int16_t)((pair & 1u) ? (w >> 16) : (w & 0xffffu));
}

WSPR_FW_HD void expand(const Visit& v, uint32_t bm_lo, uint32_t bm_hi, Expansion& e) {
    const int pos = v_pos(v), g = v_gamma(v), t = v_thr(v);
    const bool tail = pos >= kTail;
    const unsigned zp = (parity32(v.st & kPolyA) << 1) | parity32(v.st & kPolyB);
    const int m0 = metric_of(bm_lo, bm_hi, zp), m1 = metric_of(bm_lo, bm_hi, 3u ^ zp);
    const bool zero_better = tail || (m0 > m1);                       // fano.c:127-137, 170-184
    const int best = zero_better ? m0 : m1, second = zero_better ? m1 : m0;
    const uint32_t better = zero_better ? 0u : 1u;
    const int a0 = g + best, a1 = g + second;
    const bool first_visit = g < t + kDelta;                          // fano.c:158
    const bool ok0 = a0 >= t;
    const bool ok1 = ok0 && !tail && a1 >= t;
    e.done = ok0 && pos == kLast;
    e.has0 = ok0 && !e.done;
    e.has1 = ok1;
    e.look1 = (ok0 && !tail) ? 1u : 0u;
    e.again = v_retry(v) && !e.done;
    const bool up0 = first_visit && a0 >= t + kDelta;                 // tightened by exactly one step
    const bool up1 = first_visit && a1 >= t + kDelta;
    uint32_t dlo0 = v.dlo, dhi0 = v.meta & 0x3ffffu, dlo1 = v.dlo, dhi1 = dhi0;
    if (pos < 32)      { dlo0 |= better << pos;        dlo1 |= (better ^ 1u) << pos; }
    else if (pos < 50) { dhi0 |= better << (pos - 32); dhi1 |= (better ^ 1u) << (pos - 32); }
    e.kid0.st = (v.st | better) << 1;
    e.kid0.dlo = dlo0;
    e.kid0.meta = pack_meta(dhi0, pos + 1, up0);
    e.kid0.gt = pack_gt(a0, up0 ? t + kDelta : t);
    e.kid0.led = 0;
    e.kid1.st = (v.st | (better ^ 1u)) << 1;
    e.kid1.dlo = dlo1;
    e.kid1.meta = pack_meta(dhi1, pos + 1, up1);
    e.kid1.gt = pack_gt(a1, up1 ? t + kDelta : t);
    e.kid1.led = 0;
    // the visit itself one step lower; the root (node 0) keeps its retry flag for ever
    e.self = v;
    e.self.meta = pack_meta(v.meta & 0x3ffffu, pos, pos == 0);
    e.self.gt = pack_gt(g, t - kDelta);
    e.self.led = 0;
    if (e.done) {                                                     // the frame: keep decisions and the final metric
        e.self = v;
        e.self.meta = pack_meta(v.meta & 0x3ffffu, kPosDone, false);
        e.self.gt = pack_gt(a0, t);
    }
}

// decoded bytes from the 50 message decisions (byte k = decisions 8k..8k+7, first in the MSB;
// the 31 tail decisions are zero)
WSPR_FW_HD void decisions_to_bytes(uint32_t dlo, uint32_t dhi, unsigned char* data10) {
    for (int k = 0; k < 10; ++k) {
        unsigned b = 0;
        for (int j = 0; j < 8; ++j) {
            const int n = 8 * k + j;
            const unsigned bit = n < 32 ? (dlo >> n) & 1u : (n < 50 ? (dhi >> (n - 32)) & 1u : 0u);
            b = (b << 1) | bit;
        }
        data10[k] = (unsigned char)b;
    }
}

struct Result {
    int ret;              // 0 decoded, -1 cycle budget exhausted, -2 the pending-visit store overflowed (caller falls back)
    unsigned cycles;      // as the reference reports them
    unsigned metric;      // final path metric (decoded frames only)
    unsigned steps;       // expansion steps taken (diagnostics)
    unsigned char data[10];
};

}  // namespace fano_wave
}  // namespace wspr

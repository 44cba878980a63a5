// Exact jump-ahead for a float running sum  phi <- phi + d  (host + device).
//
// The reference regenerates a decoded signal's phase with a *float* accumulator: 41 472 serial
// additions per signal, the increment changing every 256 samples (wsprd/wsprd.c:340-351).  The
// rounded sums are not associative, but they are piecewise linear in exact integer arithmetic:
// while phi stays inside one binade [2^e, 2^(e+1)) its spacing is u = 2^(e-23), phi = M u with an
// integer 2^23 <= |M| < 2^24, and
//     RN(phi + d) = u * RNint(M + d/u) = u * (M + Q),   Q = RNint(d/u),
// as long as the exact sum is still inside the binade (d/u is exact: a power-of-two scaling).  Q
// does not depend on M unless d/u lies exactly half-way between two integers; then ties-to-even
// makes the result even, and from an even M the increment is again a constant (the even one of
// floor(d/u), floor(d/u)+1).  So the walk decomposes into runs
//     phi_k = (M0 + k Q) * 2^(e-23),  k = 0..n,
// whose ends are found in O(1); a step that leaves the binade (or starts from zero / an odd M on
// a tie) is taken with one real float addition.  About 200 runs replace the 41 472 additions and
// every sample's phase is then computed independently, bit-identical to the serial walk
// (tests/test_phase_runs.py compares them exhaustively over each walk).
#pragma once
#include <stdint.h>
#include <string.h>
#include <limits.h>

#if defined(__HIPCC__)
#define WSPR_PR_HD __host__ __device__ __forceinline__
#else
#define WSPR_PR_HD static inline
#endif

namespace wspr {

struct PhaseRun {
    int32_t start;      // index of the first sample of the run
    int32_t m0;         // signed 24-bit significand of its phase (0 for a zero phase)
    int32_t q;          // increment per sample, in units of 2^(e-23)
    int32_t e;          // binade exponent
};

constexpr int kPhaseMaxRuns = 512;          // per signal; a walk that needs more is evaluated per symbol instead
constexpr int32_t kPhaseRawBits = INT32_MIN; // PhaseRun::e marker: m0 holds the float's bits (zero, subnormal)

WSPR_PR_HD uint32_t pr_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
WSPR_PR_HD float pr_float(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }

// phase of sample `start + k` of a run
WSPR_PR_HD float phase_of(const PhaseRun& r, int k) {
    if (r.e == kPhaseRawBits) return pr_float((uint32_t)r.m0);
    const int32_t m = r.m0 + k * r.q;
    if (m == 0) return 0.0f;
    // (float)m is exact (|m| < 2^24); scaling by a power of two is exact for normal results
    const uint32_t scale = (uint32_t)(r.e - 23 + 127) << 23;
    return (float)m * pr_float(scale);
}

// Longest n <= steps such that phi_k = (m + k q) 2^(e-23) holds for k = 1..n when d is added n times
// to phi = m 2^(e-23); 0 if the next step has to be a real addition.  Writes q.
WSPR_PR_HD int phase_batch(float phi, float d, int steps, int32_t* m_out, int32_t* q_out, int32_t* e_out) {
    const uint32_t b = pr_bits(phi);
    const int be = (int)((b >> 23) & 0xffu);
    if (be == 0 || be == 0xff) return 0;                       // zero / subnormal / non-finite: real steps
    const int e = be - 127;
    const int32_t mag = (int32_t)((b & 0x7fffffu) | 0x800000u);
    const int32_t m = (b >> 31) ? -mag : mag;
    *m_out = m; *e_out = e;
    const uint32_t db = pr_bits(d);
    const int dbe = (int)((db >> 23) & 0xffu);
    if (dbe == 0xff) return 0;
    if ((db & 0x7fffffffu) == 0) { *q_out = 0; return steps; }  // d = +-0: phi never changes
    if (dbe == 0) return 0;                                    // subnormal increment: real steps
    // rho = d / u with u = 2^(e-23), as an exact fixed-point number: d = md 2^(de-23)  =>
    // rho = md 2^(de - e); only |rho| < 2^24 can keep the sum inside the binade
    const int de = dbe - 127;
    const int sh = de - e;                                     // rho = md * 2^sh, md in [2^23, 2^24)
    if (sh > 0) return 0;                                      // |rho| >= 2^24
    // (all quantities below are < 2^25: 32-bit integer arithmetic)
    const int32_t md = (int32_t)((db & 0x7fffffu) | 0x800000u);
    const bool dneg = (db >> 31) != 0;
    int32_t fl;              // floor(|rho|)
    int frac;                // 0: |rho| integer, 1: fraction < 1/2, 2: exactly 1/2, 3: > 1/2
    const int s = -sh;
    if (s == 0) { fl = md; frac = 0; }
    else if (s >= 25) { fl = 0; frac = 1; }                    // |rho| < 1/2
    else {
        fl = md >> s;
        const int32_t rem = md & ((1 << s) - 1), half = 1 << (s - 1);
        frac = rem == 0 ? 0 : (rem < half ? 1 : (rem == half ? 2 : 3));
    }
    int32_t qa;              // |Q|
    if (frac == 2) {
        if (m & 1) return 0;                                   // odd significand on a tie: one real step makes it even
        qa = (fl & 1) ? fl + 1 : fl;                           // from an even M the even neighbour wins every time
    } else {
        qa = fl + (frac == 3 ? 1 : 0);
    }
    *q_out = (int32_t)(dneg ? -qa : qa);
    // Every step must see its exact sum inside the binade (spacing u) and leave a normalised
    // significand: with |M| the magnitude before a step,
    //   growing  : |M| + |rho| < 2^24 and |M| + |Q| <= 2^24 - 1   <=  |M| <= 2^24 - 2 - floor|rho|
    //   shrinking: |M| - |rho| >= 2^23                             <=  |M| >= 2^23 + floor|rho| + 1
    const bool up = dneg == (m < 0);
    const int32_t room = up ? (0xfffffe - fl) - mag : mag - (0x800001 + fl);
    if (room < 0) return 0;
    if (qa == 0) return steps;                                 // |d| below half a spacing: phi is stuck
    const int32_t n = (int32_t)((uint32_t)room / (uint32_t)qa) + 1;
    return n > steps ? steps : n;
}

// One step of the decomposition: the sample at `pos` has phase `phi`; emits the run that starts
// there (covering n + 1 samples, n <= limit) and advances phi to the phase of the sample after it.
WSPR_PR_HD int phase_next_run(float& phi, float d, int pos, int limit, PhaseRun& r) {
    int32_t m = 0, q = 0, e = 0;
    const int n = phase_batch(phi, d, limit, &m, &q, &e);
    r.start = pos;
    if (n > 0) { r.m0 = m; r.q = q; r.e = e; }
    else       { r.m0 = (int32_t)pr_bits(phi); r.q = 0; r.e = kPhaseRawBits; }
    const float last = n > 0 ? phase_of(r, n) : phi;
    phi = last + d;
    return n;
}

// Decompose the walk of one signal: nsym symbols of sps samples, increment dphi[i] during symbol i,
// phi = 0 before the first sample.  runs[] receives the runs (ascending start), first_run[i] the index
// of the run containing sample i*sps (nsym + 1 entries, the last = number of runs).  Returns the
// number of runs, or -1 if max_runs is too small; sym_phi[i] (optional) = phase of sample i*sps, filled
// in either case, so that a caller can still evaluate an overflowing walk symbol by symbol
// (phase_from_symbol).  Written as one flat loop (a run per trip, symbol changes folded in) so that
// SIMT lanes working on different signals stay in step.
template <class DphiOf>
WSPR_PR_HD int phase_runs_build(const DphiOf& dphi_of, int nsym, int sps, PhaseRun* runs, int max_runs,
                                uint16_t* first_run, float* sym_phi = nullptr) {
    float phi = 0.0f, d = 0.0f;
    int nr = 0, i = -1, left = 0, pos = 0;
    bool full = false;
    for (;;) {
        if (left == 0) {
            if (++i == nsym) break;
            d = dphi_of(i);
            if (sym_phi) sym_phi[i] = phi;
            if (!full) first_run[i] = (uint16_t)nr;
            left = sps;
        }
        PhaseRun r;
        const int n = phase_next_run(phi, d, pos, left - 1, r);
        if (nr >= max_runs) full = true;
        if (!full) runs[nr++] = r;
        pos += n + 1;
        left -= n + 1;
    }
    if (full) return -1;
    first_run[nsym] = (uint16_t)nr;
    return nr;
}

// ---- symbol-level shortcut (what lets a whole wave work on one signal) -------------------------------------
// From a normal phase phi = m 2^(e-23) at a symbol's first sample, the symbol's `sps` additions of d are all
// regular (constant increment q, every sum inside the binade) iff phase_symbol_probe().ok and
// phase_symbol_room(): the symbol is then ONE run -- the run phase_next_run() would emit -- and the next symbol
// starts at significand m + sps q with the same exponent.  Between two such breaks the symbol-start
// significands are therefore a prefix sum of sps q_i, which lanes can form in parallel; a symbol that fails
// either test (binade crossing, odd significand on a tie, zero or subnormal phase, huge or subnormal
// increment) is walked with phase_next_run() as before.  The probe depends on the phase only through its
// exponent and the parity of m, and sps q is even, so the parity is the same for every symbol of a chain.
struct PhaseSymbolStep {
    int32_t q;          // signed increment per sample, units of 2^(e-23)
    int32_t fl;         // floor(|d| / 2^(e-23))
    bool dneg, ok;
};
WSPR_PR_HD PhaseSymbolStep phase_symbol_probe(int e, bool m_odd, float d) {
    PhaseSymbolStep r{0, 0, false, false};
    const uint32_t db = pr_bits(d);
    const int dbe = (int)((db >> 23) & 0xffu);
    r.dneg = (db >> 31) != 0;
    if (dbe == 0xff) return r;
    if ((db & 0x7fffffffu) == 0) { r.ok = true; return r; }       // d = +-0: the phase never changes
    if (dbe == 0) return r;                                        // subnormal increment: real steps
    const int sh = (dbe - 127) - e;
    if (sh > 0) return r;                                          // |d| / spacing >= 2^24
    const int32_t md = (int32_t)((db & 0x7fffffu) | 0x800000u);
    int32_t fl;
    int frac;                                                      // as in phase_batch()
    const int s = -sh;
    if (s == 0) { fl = md; frac = 0; }
    else if (s >= 25) { fl = 0; frac = 1; }
    else {
        fl = md >> s;
        const int32_t rem = md & ((1 << s) - 1), half = 1 << (s - 1);
        frac = rem == 0 ? 0 : (rem < half ? 1 : (rem == half ? 2 : 3));
    }
    int32_t qa;
    if (frac == 2) {
        if (m_odd) return r;                                       // one real step makes the significand even
        qa = (fl & 1) ? fl + 1 : fl;
    } else {
        qa = fl + (frac == 3 ? 1 : 0);
    }
    if (qa > (1 << 16)) return r;                                  // cannot take 256 steps inside a binade anyway
    r.q = r.dneg ? -qa : qa;
    r.fl = fl;
    r.ok = true;
    return r;
}
// all `sps` steps from significand m (2^23 <= |m| < 2^24) are regular
WSPR_PR_HD bool phase_symbol_room(int32_t m, const PhaseSymbolStep& s, int sps) {
    const int32_t mag = m < 0 ? -m : m;
    if (mag < 0x800000 || mag > 0xffffff) return false;
    const int32_t qa = s.q < 0 ? -s.q : s.q;
    const bool up = s.dneg == (m < 0);
    const int32_t room = up ? (0xfffffe - s.fl) - mag : mag - (0x800001 + s.fl);
    if (room < 0) return false;
    if (qa == 0) return true;                                      // |d| below half a spacing: the phase is stuck
    return (uint32_t)room / (uint32_t)qa + 1u >= (uint32_t)sps;
}
// float -> (m, e) of a normal phase; false for zero, subnormal and non-finite values
WSPR_PR_HD bool phase_split(float phi, int32_t* m, int* e) {
    const uint32_t b = pr_bits(phi);
    const int be = (int)((b >> 23) & 0xffu);
    if (be == 0 || be == 0xff) return false;
    const int32_t mag = (int32_t)((b & 0x7fffffu) | 0x800000u);
    *m = (b >> 31) ? -mag : mag;
    *e = be - 127;
    return true;
}
WSPR_PR_HD float phase_join(int32_t m, int e) {
    const PhaseRun r{0, m, 0, e};
    return phase_of(r, 0);
}

// The decomposition of phase_runs_build() formed chain by chain: `width` symbols are probed at once (a wave's
// lanes on the device; a loop here), the leading ones that pass become one run each with prefix-summed
// significands, the first that fails is walked serially.  Same tables as phase_runs_build(), bit for bit
// (tests/test_phase_runs.py compares them); this scalar form is what the tests run and what documents
// sub_runs_kernel's wave form.
template <class DphiOf>
WSPR_PR_HD int phase_runs_build_chained(const DphiOf& dphi_of, int nsym, int sps, PhaseRun* runs, int max_runs,
                                        uint16_t* first_run, float* sym_phi, int width) {
    float phi = 0.0f;
    int nr = 0, i = 0;
    bool full = false;
    while (i < nsym) {
        int32_t m = 0;
        int e = 0;
        int taken = 0;
        if (phase_split(phi, &m, &e)) {
            int32_t ms = m;                                        // significand at the start of symbol i + taken
            while (taken < width && i + taken < nsym) {
                const PhaseSymbolStep st = phase_symbol_probe(e, (m & 1) != 0, dphi_of(i + taken));
                if (!st.ok || !phase_symbol_room(ms, st, sps)) break;
                if (sym_phi) sym_phi[i + taken] = phase_join(ms, e);
                if (nr >= max_runs) full = true;
                if (!full) { first_run[i + taken] = (uint16_t)nr; runs[nr++] = PhaseRun{(i + taken) * sps, ms, st.q, e}; }
                ms += sps * st.q;
                ++taken;
            }
            if (taken) phi = phase_join(ms, e);
        }
        i += taken;
        if (taken == width || i >= nsym) continue;
        // symbol i does not pass: walk it run by run
        const float d = dphi_of(i);
        if (sym_phi) sym_phi[i] = phi;
        if (!full) first_run[i] = (uint16_t)nr;
        int left = sps, pos = i * sps;
        while (left > 0) {
            PhaseRun r;
            const int n = phase_next_run(phi, d, pos, left - 1, r);
            if (nr >= max_runs) full = true;
            if (!full) runs[nr++] = r;
            pos += n + 1;
            left -= n + 1;
        }
        ++i;
    }
    if (full) return -1;
    first_run[nsym] = (uint16_t)nr;
    return nr;
}

// fallback for a walk whose runs did not fit: step from the symbol's first phase
WSPR_PR_HD float phase_from_symbol(float sym_phi, float d, int j) {
    float phi = sym_phi;
    for (int k = 0; k < j; ++k) phi = phi + d;
    return phi;
}

// phase of sample n (0 <= n < nsym*sps)
WSPR_PR_HD float phase_at(const PhaseRun* runs, const uint16_t* first_run, int sps_log2, int n) {
    const int i = n >> sps_log2;
    int r = first_run[i];
    const int r_end = first_run[i + 1];
    while (r + 1 < r_end && runs[r + 1].start <= n) ++r;
    return phase_of(runs[r], n - runs[r].start);
}

}  // namespace wspr

// Test helper: the wave-parallel Fano search (csrc/kernels/fano_wave.h) emulated on the host, lane by
// lane, with the same step structure as the kernel k6_fano_wave.hip (pop the earliest visits, expand,
// cut at a completed frame, slot/ledger prefix sums, push in walk order).  Callable from ctypes with the
// reference fano() argument meaning; `width` = lanes per step, `cap` = pending-visit capacity.
#include <algorithm>
#include <cstring>
#include <vector>
#include "../../rtlsdr-wsprd_amd/csrc/kernels/fano_wave.h"

using namespace wspr::fano_wave;

extern "C" int fano_wave_host(unsigned* metric, unsigned* cycles, unsigned* maxnp, unsigned char* data,
                              const unsigned char* symbols, unsigned nbits, const int mettab[2][256], int delta,
                              unsigned maxcycles, int width, int cap, unsigned* steps_out, unsigned* maxsize_out) {
    if (nbits != (unsigned)kBits || delta != kDelta || width < 1 || width > 64) return -3;
    uint32_t bm_lo[kBits], bm_hi[kBits];
    for (int k = 0; k < kBits; ++k) {
        const int a0 = mettab[0][symbols[2 * k]], a1 = mettab[1][symbols[2 * k]];
        const int b0 = mettab[0][symbols[2 * k + 1]], b1 = mettab[1][symbols[2 * k + 1]];
        bm_lo[k] = (uint32_t)(uint16_t)(short)(a0 + b0) | ((uint32_t)(uint16_t)(short)(a0 + b1) << 16);
        bm_hi[k] = (uint32_t)(uint16_t)(short)(a1 + b0) | ((uint32_t)(uint16_t)(short)(a1 + b1) << 16);
    }
    std::vector<Visit> pool((size_t)cap);
    pool[0] = Visit{0u, 0u, pack_meta(0u, 0, true), pack_gt(0, 0), 0u};
    int size = 1;
    unsigned settled = 0, steps = 0, maxsize = 1;
    const unsigned budget = maxcycles * (unsigned)kBits;
    int rc = -2;
    unsigned out_cycles = 0, out_metric = 0;
    uint32_t out_dlo = 0, out_dhi = 0;
    for (;;) {
        if (settled >= budget) { rc = -1; out_cycles = budget + 2; break; }
        const int room = cap - size;
        const int wide = size <= cap / 2 ? width : (room > 256 ? std::min(width, 8) : 1);
        const int take = std::min(std::min(wide, size), room >> 1);
        if (take < 1 || steps > 4u * budget + 1024u) { rc = -2; break; }
        ++steps;
        Visit x[64]; Expansion e[64];
        bool donev[64], live[64], has0[64], has1[64], again[64], keep[64];
        uint32_t back[64], pre[64];
        for (int l = 0; l < take; ++l) x[l] = pool[size - 1 - l];
        if (v_pos(x[0]) == kPosDone) {
            const unsigned looks = settled + 1;
            rc = looks >= budget ? -1 : 0;
            out_cycles = looks + 1;
            out_metric = (unsigned)v_gamma(x[0]);
            out_dlo = x[0].dlo; out_dhi = x[0].meta & 0x3ffffu;
            break;
        }
        int jc = -1;
        for (int l = 0; l < take; ++l) {
            const int pos = v_pos(x[l]);
            donev[l] = pos == kPosDone;
            const int pc = std::min(pos, kLast);
            expand(x[l], bm_lo[pc], bm_hi[pc], e[l]);
            if (jc < 0 && (donev[l] || e[l].done)) jc = l;
        }
        int base = size - take;
        if (jc >= 0) base = 0;
        int total = 0;
        bool low = false;
        for (int l = 0; l < take; ++l) {
            live[l] = jc < 0 || l <= jc;
            keep[l] = live[l] && (donev[l] || e[l].done);
            has0[l] = live[l] && !keep[l] && e[l].has0;
            has1[l] = live[l] && !keep[l] && e[l].has1;
            again[l] = live[l] && (keep[l] || e[l].again);
            const int c = has0[l] + has1[l] + again[l];
            back[l] = (live[l] && !keep[l]) ? 1u + (c == 0 ? x[l].led : 0u) : 0u;
            pre[l] = back[l] + (l ? pre[l - 1] : 0u);
            total += c;
            low |= again[l] && !keep[l] && v_thr(e[l].self) < -32000;
        }
        if (low) { rc = -2; break; }
        int first = -1;
        for (int l = 0; l < take; ++l) if (has0[l] || has1[l] || again[l]) { first = l; break; }
        settled += first >= 0 ? pre[first] : pre[take - 1];
        const int top = base + total - 1;
        int before = 0;
        for (int l = 0; l < take; ++l) {
            int nxt = take - 1;
            for (int m = l + 1; m < take; ++m) if (has0[m] || has1[m] || again[m]) { nxt = m; break; }
            const uint32_t tail_led = x[l].led + (pre[nxt] - pre[l]);
            if (has0[l]) { Visit k = e[l].kid0; k.led = e[l].look1 + ((!has1[l] && !again[l]) ? tail_led : 0u); pool[top - before] = k; }
            if (has1[l]) { Visit k = e[l].kid1; k.led = !again[l] ? tail_led : 0u; pool[top - before - has0[l]] = k; }
            if (again[l]) { Visit s = donev[l] ? x[l] : e[l].self; s.led = keep[l] ? 0u : tail_led; pool[top - before - has0[l] - has1[l]] = s; }
            before += has0[l] + has1[l] + again[l];
        }
        size = base + total;
        maxsize = std::max(maxsize, (unsigned)size);
    }
    *cycles = out_cycles;
    *metric = rc == 0 ? out_metric : 0u;
    *maxnp = rc == 0 ? (unsigned)kLast : 0u;
    decisions_to_bytes(out_dlo, out_dhi, data);
    if (steps_out) *steps_out = steps;
    if (maxsize_out) *maxsize_out = maxsize;
    return rc;
}

/* ============================================================================
 * orc_frontend.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Front end of the reference receiver, restated:
 *   fs/4 mixer + 2-stage CIC (R = 6401) + 33-tap FIR    rtlsdr_wsprd.c:126-244
 *   tail zero-fill + max-abs normalisation to 0.5        rtlsdr_wsprd.c:284-305
 *   .iq file convention (Q negated, normalise)           rtlsdr_wsprd.c:555-592
 *
 * PARITY UNPINNED for the decimator: rtlsdr_wsprd.c needs <rtl-sdr.h>, libusb and
 * libcurl headers that are absent from this image (no stand-ins are written), the
 * callback is `static`, and no reference test or fixture holds decimator output.
 * The restatement is checked by construction properties only (tests/).
 * ==========================================================================*/
#include "wspr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* rtlsdr_wsprd.c:142-152: CIC compensation FIR, 33 symmetric taps; stored as the
 * 16 distinct side taps + centre and mirrored at start-up. */
static const float fir_half[17] = {
    -0.0027772683f, -0.0005058826f,  0.0049745750f, -0.0034059318f,
    -0.0077557814f,  0.0139375423f,  0.0039896935f, -0.0299394142f,
     0.0162250643f,  0.0405130860f, -0.0580746013f, -0.0272104968f,
     0.1183705475f, -0.0306029022f, -0.2011241667f,  0.1615898423f,
     0.5000000000f
};

#define ORC_DOWNSAMPLING (2400000u / 375u)     /* SAMPLING_RATE / SIGNAL_SAMPLE_RATE, rtlsdr_wsprd.c:36-41 */

struct orc_decim_state {
    uint32_t i1, i2, q1, q2;            /* integrators (mod 2^32)           */
    uint32_t ic1[2], ic2[2];            /* comb delay lines, I              */
    uint32_t qc1[2], qc2[2];            /* comb delay lines, Q              */
    uint32_t phase;                     /* samples since last output        */
    float    fi[32], fq[32];            /* FIR history, oldest first        */
    float    taps[33];
};

orc_decim_state *orc_decim_new(void) {
    orc_decim_state *s = (orc_decim_state *)calloc(1, sizeof *s);
    if (!s) return NULL;
    for (int i = 0; i < 17; i++) { s->taps[i] = fir_half[i]; s->taps[32 - i] = fir_half[i]; }
    return s;
}
void orc_decim_free(orc_decim_state *s) { free(s); }

/* the constants of the front end as this restatement uses them: 33 FIR taps (rtlsdr_wsprd.c:142-152) and the
 * number of input samples per output (rtlsdr_wsprd.c:198-202: DOWNSAMPLING + 1) */
void orc_front_end_constants(float *taps33, int *samples_per_output) {
    for (int i = 0; i < 17; i++) { taps33[i] = fir_half[i]; taps33[32 - i] = fir_half[i]; }
    *samples_per_output = (int)ORC_DOWNSAMPLING + 1;
}

static inline int8_t s8(unsigned char b) { return (int8_t)(b ^ 0x80); }
static inline int8_t neg8(int8_t v) { return (int8_t)(uint8_t)(0u - (uint8_t)v); } /* -(-128) = -128 */

uint32_t orc_decim_feed(orc_decim_state *s, const unsigned char *iq, size_t nbytes,
                        float *I, float *Q, uint32_t fill, uint32_t cap) {
    for (size_t n = 0; n + 1 < nbytes; n += 2) {
        /* mixer, rtlsdr_wsprd.c:170-182: multiply sample m by (1, j, -1, -j)[m & 3] */
        int8_t a = s8(iq[n]), b = s8(iq[n + 1]), xi, xq;
        switch ((n >> 1) & 3) {
            case 0:  xi = a;        xq = b;        break;
            case 1:  xi = neg8(b);  xq = a;        break;
            case 2:  xi = neg8(a);  xq = neg8(b);  break;
            default: xi = b;        xq = neg8(a);  break;
        }
        /* integrators, :190-195 */
        s->i1 += (uint32_t)(int32_t)xi;  s->q1 += (uint32_t)(int32_t)xq;
        s->i2 += s->i1;                  s->q2 += s->q1;
        /* decimate by 6401, :197-202 */
        if (++s->phase <= ORC_DOWNSAMPLING) continue;      /* decimationIndex <= DOWNSAMPLING, then reset */
        s->phase = 0;
        /* two combs with a two-output delay, :204-218 */
        uint32_t iy1 = s->i2 - s->ic1[1];  s->ic1[1] = s->ic1[0];  s->ic1[0] = s->i2;
        uint32_t qy1 = s->q2 - s->qc1[1];  s->qc1[1] = s->qc1[0];  s->qc1[0] = s->q2;
        uint32_t iy2 = iy1 - s->ic2[1];    s->ic2[1] = s->ic2[0];  s->ic2[0] = iy1;
        uint32_t qy2 = qy1 - s->qc2[1];    s->qc2[1] = s->qc2[0];  s->qc2[0] = qy1;
        /* FIR, :220-234: 32 old outputs (oldest first) then the new one on tap 32 */
        float si = 0.0f, sq = 0.0f;
        for (int j = 0; j < 32; j++) {
            float pi = s->fi[j] * s->taps[j], pq = s->fq[j] * s->taps[j];
            si += pi;
            sq += pq;
        }
        memmove(s->fi, s->fi + 1, 31 * sizeof(float));
        memmove(s->fq, s->fq + 1, 31 * sizeof(float));
        s->fi[31] = (float)(int32_t)iy2;
        s->fq[31] = (float)(int32_t)qy2;
        float pi = s->fi[31] * s->taps[32], pq = s->fq[31] * s->taps[32];
        si += pi;
        sq += pq;
        if (fill < cap) { I[fill] = si; Q[fill] = sq; fill++; }     /* :236-242 */
    }
    return fill;
}

/* rtlsdr_wsprd.c:284-305 (also :574-589) */
void orc_normalise(float *I, float *Q, int n_valid, int n_total) {
    for (int i = n_valid; i < n_total; i++) { I[i] = 0.0f; Q[i] = 0.0f; }
    float peak = 1e-24f;
    for (int i = 0; i < n_total; i++) {
        float a = fabs(I[i]), b = fabs(Q[i]);
        if (a > peak) peak = a;
        if (b > peak) peak = b;
    }
    float scale = 0.5 / peak;
    for (int i = 0; i < n_total; i++) { I[i] *= scale; Q[i] *= scale; }
}

/* rtlsdr_wsprd.c:555-592: interleaved float32 (I, Q) pairs, Q sign flipped */
int orc_iq_from_interleaved(const float *f, int nfloats, float *I, float *Q) {
    int n = nfloats / 2;
    if (n > ORC_MAXSAMPLES) n = ORC_MAXSAMPLES;
    for (int i = 0; i < n; i++) { I[i] = f[2 * i]; Q[i] = -f[2 * i + 1]; }
    orc_normalise(I, Q, n, n);
    return n;
}

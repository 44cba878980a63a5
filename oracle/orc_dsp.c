/* ============================================================================
 * orc_dsp.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Signal-processing stages of the reference decoder, restated stage by stage
 * with the reference's float/double evaluation order (x86-64 SSE, no FMA):
 *   FFT bank + power spectrogram   wsprd/wsprd.c:509-553
 *   peak picker                    wsprd/wsprd.c:555-631
 *   coarse (freq, lag, drift) sync wsprd/wsprd.c:646-678
 *   fine sync / soft demodulator   wsprd/wsprd.c:101-259  (sync_and_demodulate)
 *   coherent subtraction           wsprd/wsprd.c:316-413  (subtract_signal2)
 *   decode orchestration           wsprd/wsprd.c:416-855  (wspr_decode)
 *
 * PARITY PIN.  wsprd/wsprd.c cannot be compiled in this image: it needs
 * <fftw3.h>/libfftw3f (un-vendored, unpinned system library, Makefile:3) which
 * is absent, and no stand-in is written for it.  This restatement is therefore
 * pinned by (tests/test_oracle_golden.py):
 *   - the reference's documented spot lines for signals/refSignalSnr0dB.iq and
 *     for the -t self-test (documentation/bug-fix/REPORT.md:198,202),
 *   - the self-test acceptance rule of rtlsdr_wsprd.c:782-788,
 *   - the per-stage anchor values recorded from the reference in SURVEY.md §8(c).
 * The FFT itself (FFTW codelets in the reference) is float32 radix-2 here and is
 * only tolerance-comparable (no reference test pins FFT output).
 *
 * Build with -ffp-contract=off (oracle/Makefile).
 * ==========================================================================*/
#include "wspr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* macro expansions of wsprd.c:59-69, evaluated left to right as C does */
static const double kTwoPiDt = 2.0 * M_PI * 1.0 / 375.0;    /* TWOPIDT          */
static const double kDf      = 375.0 / 256.0;               /* DF  (as a value) */
static const double kDf05    = 375.0 / 256.0 * 0.5;         /* DF05             */
static const double kDf15    = 375.0 / 256.0 * 1.5;         /* DF15             */
static const double kHalfDf  = 375.0 / 256.0 / 2.0;         /* (DF / 2.0)       */

/* ------------------------------------------------------------------ FFT -- */
static float tw_re[256], tw_im[256];
static int   tw_ready = 0;
static void tw_init(void) {
    for (int k = 0; k < 256; k++) {
        double a = 2.0 * M_PI * (double)k / 512.0;
        tw_re[k] = (float)cos(a);
        tw_im[k] = (float)(-sin(a));
    }
    tw_re[0] = 1.0f;   tw_im[0] = 0.0f;      /* exact trivial twiddles */
    tw_re[128] = 0.0f; tw_im[128] = -1.0f;
    tw_ready = 1;
}

static inline unsigned rev9(unsigned v) {
    unsigned r = 0;
    for (int b = 0; b < 9; b++) r |= ((v >> b) & 1u) << (8 - b);
    return r;
}

/* Decimation-in-frequency radix-2, 9 stages, natural-order in, natural-order out.
 * Butterfly (u,v) -> (u+v, (u-v)*w), w = exp(-2*pi*i*j*2^s/512), complex multiply
 * as (dr*wr - di*wi, dr*wi + di*wr) with separate roundings.  The MI355X kernel
 * performs the same butterflies, so both produce identical bits. */
void orc_fft512(float *re, float *im) {
    if (!tw_ready) tw_init();
    for (int s = 0; s < 9; s++) {
        int half = 256 >> s;
        for (int base = 0; base < 512; base += 2 * half) {
            for (int j = 0; j < half; j++) {
                int a = base + j, b = a + half;
                float ur = re[a], ui = im[a], vr = re[b], vi = im[b];
                float dr = ur - vr, di = ui - vi;
                float wr = tw_re[j << s], wi = tw_im[j << s];
                re[a] = ur + vr;
                im[a] = ui + vi;
                float t1 = dr * wr, t2 = di * wi, t3 = dr * wi, t4 = di * wr;
                re[b] = t1 - t2;
                im[b] = t3 + t4;
            }
        }
    }
    float tr[512], ti[512];
    memcpy(tr, re, sizeof tr);
    memcpy(ti, im, sizeof ti);
    for (unsigned n = 0; n < 512; n++) {
        unsigned k = rev9(n);
        re[k] = tr[n];
        im[k] = ti[n];
    }
}

/* wsprd.c:516 */
int orc_blocks_for(int samples) { return 4 * (samples / ORC_FFT) - 1; }

/* wsprd.c:509-513 (window) and :536-553 (blocks, fft-shift, |.|^2) */
void orc_fft_bank(const float *idat, const float *qdat, int samples, float *ps) {
    const int blocks = orc_blocks_for(samples);
    float win[ORC_FFT];
    for (int j = 0; j < ORC_FFT; j++) win[j] = sinf(0.006147931 * j);
    float xr[ORC_FFT], xi[ORC_FFT];
    for (int t = 0; t < blocks; t++) {
        for (int j = 0; j < ORC_FFT; j++) {
            int k = t * 128 + j;
            xr[j] = idat[k] * win[j];
            xi[j] = qdat[k] * win[j];
        }
        if (orc_get_fft_variant() == 0) orc_fft512(xr, xi);
        else                            orc_fft512_variant(xr, xi);   /* robustness study only: orc_fft_alt.c */
        for (int j = 0; j < ORC_FFT; j++) {
            int k = (j + ORC_FFT / 2) & (ORC_FFT - 1);
            float a = xr[k] * xr[k], b = xi[k] * xi[k];
            ps[(size_t)j * blocks + t] = a + b;
        }
    }
}

/* ---------------------------------------------------------- peak picker -- */
static int cmp_float_asc(const void *a, const void *b) {   /* wsprd_utils.c:222 */
    float x = *(const float *)a, y = *(const float *)b;
    if (x < y) return -1;
    return x > y;
}
static int cmp_cand_snr_desc(const void *a, const void *b) {   /* wsprd.c:47-51 */
    float x = ((const orc_cand *)a)->snr, y = ((const orc_cand *)b)->snr;
    return (x < y) - (x > y);
}
static int cmp_spot_snr_desc(const void *a, const void *b) {   /* wsprd.c:53-57 */
    float x = ((const orc_spot *)a)->snr, y = ((const orc_spot *)b)->snr;
    return (x < y) - (x > y);
}

/* wsprd.c:555-631 */
int orc_pick_peaks(const float *ps, int blocks, orc_cand *cand,
                   float *noise_level, float *smspec_raw, float *smspec_norm) {
    float psavg[ORC_FFT];
    for (int j = 0; j < ORC_FFT; j++) psavg[j] = 0.0f;
    for (int t = 0; t < blocks; t++)
        for (int j = 0; j < ORC_FFT; j++) psavg[j] += ps[(size_t)j * blocks + t];

    float sm[411], sorted[411];
    for (int i = 0; i < 411; i++) {
        float acc = 0.0f;
        for (int d = -3; d <= 3; d++) acc += 1 * psavg[256 - 205 + i + d];
        sm[i] = acc;
    }
    if (smspec_raw) memcpy(smspec_raw, sm, sizeof sm);
    memcpy(sorted, sm, sizeof sm);
    qsort(sorted, 411, sizeof(float), cmp_float_asc);
    float noise = sorted[122];
    if (noise_level) *noise_level = noise;

    float min_snr = powf(10.0, -8.0 / 10.0);
    float snr_off = 26.3;
    for (int j = 0; j < 411; j++) {
        sm[j] = sm[j] / noise - 1.0;
        if (sm[j] < min_snr) sm[j] = 0.1 * min_snr;
    }
    if (smspec_norm) memcpy(smspec_norm, sm, sizeof sm);

    for (int i = 0; i < ORC_MAXCAND; i++) {
        cand[i].freq = 0.0f; cand[i].snr = 0.0f; cand[i].drift = 0.0f;
        cand[i].shift = 0;   cand[i].sync = 0.0f;
    }
    int npk = 0;
    for (int j = 1; j < 410; j++) {
        if (sm[j] > sm[j - 1] && sm[j] > sm[j + 1] && npk < ORC_MAXCAND) {
            cand[npk].freq = (j - 205) * kHalfDf;
            cand[npk].snr  = 10.0 * log10f(sm[j]) - snr_off;
            npk++;
        }
    }
    const float fmin = -110.0f, fmax = 110.0f;
    int kept = 0;
    for (int j = 0; j < npk; j++)
        if (cand[j].freq >= fmin && cand[j].freq <= fmax) cand[kept++] = cand[j];
    npk = kept;
    qsort(cand, npk, sizeof(orc_cand), cmp_cand_snr_desc);
    return npk;
}

/* ---------------------------------------------------------- coarse sync -- */
/* wsprd.c:646-678.  ps is addressed as one flat row-major array exactly as the
 * reference's VLA is laid out, so a negative time index lands in the previous
 * bin's row; "/ DF" is the unparenthesised macro, i.e. /375.0/256.0
 * (SURVEY Q1). */
void orc_coarse_sync(const float *ps, int blocks, orc_cand *cand, int npk, int maxdrift) {
    const unsigned char *pr3 = orc_sync_vector;
    for (int c = 0; c < npk; c++) {
        float sync = 0.0f, best = -1e30f;
        int if0 = cand[c].freq / kHalfDf + ORC_SPS;
        for (int ifr = if0 - 1; ifr <= if0 + 1; ifr++) {
            for (int k0 = -10; k0 < 22; k0++) {
                for (int idr = -maxdrift; idr <= maxdrift; idr++) {
                    float ss = 0.0f, pw = 0.0f;
                    for (int k = 0; k < ORC_NSYM; k++) {
                        int ifd = ifr + ((float)k - (float)ORC_NBITS) / (float)ORC_NBITS
                                            * ((float)idr) / 375.0 / 256.0;
                        int kidx = k0 + 2 * k;
                        if (kidx < blocks) {
                            long o = (long)kidx;
                            float p0 = sqrtf(ps[(long)(ifd - 3) * blocks + o]);
                            float p1 = sqrtf(ps[(long)(ifd - 1) * blocks + o]);
                            float p2 = sqrtf(ps[(long)(ifd + 1) * blocks + o]);
                            float p3 = sqrtf(ps[(long)(ifd + 3) * blocks + o]);
                            ss = ss + (2 * pr3[k] - 1) * ((p1 + p3) - (p0 + p2));
                            pw = pw + p0 + p1 + p2 + p3;
                            sync = ss / pw;
                        }
                    }
                    if (sync > best) {
                        best = sync;
                        cand[c].shift = 128 * (k0 + 1);
                        cand[c].drift = idr;
                        cand[c].freq  = (ifr - ORC_SPS) * kHalfDf;
                        cand[c].sync  = sync;
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------ fine sync / demodulator -- */
static inline unsigned char soft_to_u8(float v) {
    if (v != v) return 0;              /* NaN: x86 cvttss2si -> 0x80000000 -> low byte 0 */
    return (unsigned char)(int)v;
}

/* wsprd.c:101-259.  mode 0: scan lags; mode 1: scan frequencies; mode 2: soft
 * symbols at the given (freq, lag). */
void orc_sync_demod(const float *id, const float *qd, long np, unsigned char *symbols,
                    float *freq, int ifmin, int ifmax, float fstep,
                    int *shift, int lagmin, int lagmax, int lagstep,
                    const float *drift, int symfac, float *sync, int mode) {
    const unsigned char *pr3 = orc_sync_vector;
    float ct[4][ORC_SPS], st[4][ORC_SPS];
    float fsymb[ORC_NSYM];
    float syncmax = -1e30f, fbest = 0.0f;
    int   best_shift = 0;

    if (mode == 0) { ifmin = 0; ifmax = 0; fstep = 0.0f; }
    else if (mode == 1) { lagmin = *shift; lagmax = *shift; }
    else if (mode == 2) { lagmin = *shift; lagmax = *shift; ifmin = 0; ifmax = 0; }

    for (int ifreq = ifmin; ifreq <= ifmax; ifreq++) {
        float f0 = *freq + ifreq * fstep;
        for (int lag = lagmin; lag <= lagmax; lag += lagstep) {
            float ss = 0.0f, totp = 0.0f;
            float fplast = 0.0f;
            for (int i = 0; i < ORC_NSYM; i++) {
                float fp = f0 + (*drift / 2.0) * ((float)i - (float)ORC_NBITS) / (float)ORC_NBITS;
                if (i == 0 || fp != fplast) {
                    /* tone t sits at fp + (t - 1.5) * DF */
                    float dphi[4];
                    dphi[0] = kTwoPiDt * (fp - kDf15);
                    dphi[1] = kTwoPiDt * (fp - kDf05);
                    dphi[2] = kTwoPiDt * (fp + kDf05);
                    dphi[3] = kTwoPiDt * (fp + kDf15);
                    for (int t = 0; t < 4; t++) {
                        float cd = cosf(dphi[t]), sd = sinf(dphi[t]);
                        ct[t][0] = 1.0f; st[t][0] = 0.0f;
                        for (int j = 1; j < ORC_SPS; j++) {
                            float a = ct[t][j - 1] * cd, b = st[t][j - 1] * sd;
                            float c = ct[t][j - 1] * sd, d = st[t][j - 1] * cd;
                            ct[t][j] = a - b;
                            st[t][j] = c + d;
                        }
                    }
                    fplast = fp;
                }
                float ai[4] = {0, 0, 0, 0}, aq[4] = {0, 0, 0, 0};
                for (int j = 0; j < ORC_SPS; j++) {
                    int k = lag + i * ORC_SPS + j;
                    if (k > 0 && k < np) {
                        float x = id[k], y = qd[k];
                        for (int t = 0; t < 4; t++) {
                            float m1 = x * ct[t][j], m2 = y * st[t][j];
                            float m3 = x * st[t][j], m4 = y * ct[t][j];
                            ai[t] = (ai[t] + m1) + m2;
                            aq[t] = (aq[t] - m3) + m4;
                        }
                    }
                }
                float p[4];
                for (int t = 0; t < 4; t++) {
                    float e1 = ai[t] * ai[t], e2 = aq[t] * aq[t];
                    float e = e1 + e2;
                    p[t] = sqrt(e);
                }
                totp = totp + p[0] + p[1] + p[2] + p[3];
                float cmet = (p[1] + p[3]) - (p[0] + p[2]);
                ss = (pr3[i] == 1) ? ss + cmet : ss - cmet;
                if (mode == 2) fsymb[i] = (pr3[i] == 1) ? p[3] - p[1] : p[2] - p[0];
            }
            ss = ss / totp;
            if (ss > syncmax) { syncmax = ss; best_shift = lag; fbest = f0; }
        }
    }

    if (mode <= 1) {
        *sync = syncmax; *shift = best_shift; *freq = fbest;
        return;
    }
    /* mode 2: wsprd.c:243-256 */
    *sync = syncmax;
    float fsum = 0.0f, f2sum = 0.0f;
    for (int i = 0; i < ORC_NSYM; i++) {
        fsum  += fsymb[i] / ORC_NSYM;
        f2sum += fsymb[i] * fsymb[i] / ORC_NSYM;
    }
    float m2 = fsum * fsum;
    float var = f2sum - m2;
    float fac = sqrt(var);
    for (int i = 0; i < ORC_NSYM; i++) {
        float v = symfac * fsymb[i] / fac;
        if (v > 127) v = 127.0f;
        if (v < -128) v = -128.0f;
        symbols[i] = soft_to_u8(v + 128);
    }
}

/* --------------------------------------------------- coherent subtraction -- */
/* wsprd.c:316-413 */
void orc_subtract(float *id, float *qd, long np, float f0, int shift, float drift,
                  const unsigned char *cs) {
    enum { NF = 360, NS = ORC_MAXSAMPLES, NSIG = ORC_NSYM * ORC_SPS };
    float *buf = (float *)calloc((size_t)6 * NS, sizeof(float));
    if (!buf) return;
    float *refi = buf, *refq = buf + NS, *ci = buf + 2 * NS, *cq = buf + 3 * NS,
          *cfi = buf + 4 * NS, *cfq = buf + 5 * NS;

    float phi = 0.0f;
    for (int i = 0; i < ORC_NSYM; i++) {
        float s = (float)cs[i];
        float dphi = kTwoPiDt * (f0 + (drift / 2.0) * ((float)i - (float)ORC_NSYM / 2.0)
                                      / ((float)ORC_NSYM / 2.0) + (s - 1.5) * 375.0 / 256.0);
        for (int j = 0; j < ORC_SPS; j++) {
            int n = ORC_SPS * i + j;
            refi[n] = cosf(phi);
            refq[n] = sinf(phi);
            phi = phi + dphi;
        }
    }

    float w[NF], part[NF], norm = 0.0f;
    for (int i = 0; i < NF; i++) {
        w[i] = sinf(M_PI * (float)i / (float)(NF - 1));
        norm = norm + w[i];
    }
    for (int i = 0; i < NF; i++) w[i] = w[i] / norm;
    part[0] = 0.0f;
    for (int i = 1; i < NF; i++) part[i] = part[i - 1] + w[i];

    for (int i = 0; i < NSIG; i++) {
        int k = shift + i;
        if (k > 0 && k < np) {
            float a = id[k] * refi[i], b = qd[k] * refq[i];
            float c = qd[k] * refi[i], d = id[k] * refq[i];
            ci[i + NF] = a + b;
            cq[i + NF] = c - d;
        }
    }
    for (int i = NF / 2; i < NS - NF / 2; i++) {
        float si = 0.0f, sq = 0.0f;
        for (int j = 0; j < NF; j++) {
            float a = w[j] * ci[i - NF / 2 + j], b = w[j] * cq[i - NF / 2 + j];
            si = si + a;
            sq = sq + b;
        }
        cfi[i] = si;
        cfq[i] = sq;
    }
    for (int i = 0; i < NSIG; i++) {
        if (i < NF / 2)                  norm = part[NF / 2 + i];
        else if (i > NSIG - 1 - NF / 2)  norm = part[NF / 2 + NSIG - 1 - i];
        else                             norm = 1.0f;
        int k = shift + i, j = i + NF;
        if (k > 0 && k < np) {
            float a = cfi[j] * refi[i], b = cfq[j] * refq[i];
            float c = cfi[j] * refq[i], d = cfq[j] * refi[i];
            float ri = a - b, rq = c + d;
            id[k] = id[k] - ri / norm;
            qd[k] = qd[k] - rq / norm;
        }
    }
    free(buf);
}

/* wsprd.c:263-312: symbol-by-symbol subtraction (exported by wsprd.h:83-89, not called by wspr_decode) */
void orc_subtract_simple(float *id, float *qd, long np, float f0, int shift, float drift,
                         const unsigned char *cs) {
    float c0[ORC_SPS], s0[ORC_SPS];
    for (int i = 0; i < ORC_NSYM; i++) {
        float fp = f0 + ((float)drift / 2.0) * ((float)i - (float)ORC_NBITS) / (float)ORC_NBITS;
        float dphi = kTwoPiDt * (fp + ((float)cs[i] - 1.5) * 375.0 / 256.0);
        float cdphi = cosf(dphi), sdphi = sinf(dphi);
        c0[0] = 1; s0[0] = 0;
        for (int j = 1; j < ORC_SPS; j++) {
            c0[j] = c0[j - 1] * cdphi - s0[j - 1] * sdphi;
            s0[j] = c0[j - 1] * sdphi + s0[j - 1] * cdphi;
        }
        float i0 = 0.0, q0 = 0.0;
        for (int j = 0; j < ORC_SPS; j++) {
            int k = shift + i * ORC_SPS + j;
            if ((k > 0) && (k < np)) {
                i0 = i0 + id[k] * c0[j] + qd[k] * s0[j];
                q0 = q0 - id[k] * s0[j] + qd[k] * c0[j];
            }
        }
        i0 = i0 / (float)ORC_SPS;
        q0 = q0 / (float)ORC_SPS;
        for (int j = 0; j < ORC_SPS; j++) {
            int k = shift + i * ORC_SPS + j;
            if ((k > 0) && (k < np)) {
                id[k] = id[k] - (i0 * c0[j] - q0 * s0[j]);
                qd[k] = qd[k] - (q0 * c0[j] + i0 * s0[j]);
            }
        }
    }
}

/* ----------------------------------------------------------- orchestration -- */
/* wsprd.c:416-855, including the hashtable.txt persistence of :481-494 / :842-852 when
 * options.usehashtable is set.  The fftw_wisdom.dat side effect is FFTW-specific and not restated. */
int orc_wspr_decode(float *idat, float *qdat, int samples, orc_options opt,
                    orc_spot *spots, int *n_results, orc_trace *tr) {
    const float minsync1 = 0.10f;
    float minsync2 = 0.12f;
    const int iifac = 3, symfac = 50;
    int   maxdrift = 4;
    const float minrms = 52.0 * (symfac / 64.0);
    const int delta = 60;
    const unsigned maxcycles = 10000;

    int mettab[2][256];
    orc_build_mettab(mettab);

    char *hashtab = (char *)calloc((size_t)ORC_HASH_N * ORC_HASH_W, 1);
    char *loctab  = (char *)calloc((size_t)ORC_HASH_N * ORC_LOC_W, 1);
    if (opt.usehashtable) {                                   /* wsprd.c:481-494 */
        FILE *fh = fopen("hashtable.txt", "r+");
        if (fh) {
            char line[80], hcall[13], hgrid[5];
            int nh;
            while (fgets(line, sizeof line, fh) != NULL) {
                hgrid[0] = '\0';
                hcall[0] = '\0';
                if (sscanf(line, "%d %12s %4s", &nh, hcall, hgrid) < 2) continue;
                if (nh >= 0 && nh < ORC_HASH_N) {
                    snprintf(hashtab + nh * ORC_HASH_W, ORC_HASH_W, "%s", hcall);
                    if (strlen(hgrid) > 0) snprintf(loctab + nh * ORC_LOC_W, ORC_LOC_W, "%s", hgrid);
                }
            }
            fclose(fh);
        }
    }
    const int blocks = orc_blocks_for(samples);
    float *ps = (float *)calloc((size_t)ORC_FFT * (blocks > 0 ? blocks : 1), sizeof(float));
    orc_cand cand[ORC_MAXCAND];
    float allfreqs[ORC_MAXUNIQ];
    char  allcalls[ORC_MAXUNIQ][ORC_HASH_W];
    memset(allfreqs, 0, sizeof allfreqs);
    memset(allcalls, 0, sizeof allcalls);
    int uniques = 0;
    unsigned metric = 0, cycles = 0, maxnp = 0;
    unsigned char symbols[ORC_NSYM], decdata[11];
    signed char message[12];
    memset(symbols, 0, sizeof symbols);
    memset(decdata, 0, sizeof decdata);
    memset(message, 0, sizeof message);
    if (tr) { memset(tr, 0, sizeof *tr); tr->blocks = blocks; }

    for (int ipass = 0; ipass < opt.npasses; ipass++) {
        if (ipass == 1 && uniques == 0) break;
        if (ipass < 2) { maxdrift = 4; minsync2 = 0.12f; }
        if (ipass == 2) { maxdrift = 0; minsync2 = 0.10f; }

        orc_fft_bank(idat, qdat, samples, ps);
        float noise;
        int npk = orc_pick_peaks(ps, blocks, cand, &noise,
                                 (tr && ipass < ORC_TRACE_PASSES) ? tr->smspec_raw[ipass] : NULL, NULL);
        if (tr && ipass < ORC_TRACE_PASSES) {
            tr->passes_run = ipass + 1;
            tr->noise_level[ipass] = noise;
            tr->npk[ipass] = npk;
            memcpy(tr->cand_peaks[ipass], cand, sizeof cand);
        }
        orc_coarse_sync(ps, blocks, cand, npk, maxdrift);
        if (tr && ipass < ORC_TRACE_PASSES) memcpy(tr->cand_coarse[ipass], cand, sizeof cand);

        int stop = 0;
        for (int j = 0; j < npk && !stop; j++) {
            char callsign[ORC_HASH_W], call_loc_pow[23], call[ORC_HASH_W], loc[7], pwr[3];
            memset(callsign, 0, sizeof callsign);
            memset(call_loc_pow, 0, sizeof call_loc_pow);
            memset(call, 0, sizeof call);
            memset(loc, 0, sizeof loc);
            memset(pwr, 0, sizeof pwr);

            float freq = cand[j].freq, drift = cand[j].drift, sync = cand[j].sync;
            int   shift = cand[j].shift;
            int   lagmin = shift - 128, lagmax = shift + 128;
            int   lagstep = opt.quickmode ? 16 : 8;

            orc_sync_demod(idat, qdat, samples, symbols, &freq, 0, 0, 0.0f, &shift,
                           lagmin, lagmax, lagstep, &drift, symfac, &sync, 0);
            if (tr && ipass < ORC_TRACE_PASSES) {
                tr->n_visited[ipass] = j + 1;
                tr->mode0_shift[ipass][j] = shift;
                tr->mode0_sync[ipass][j] = sync;
            }
            float fstep = 0.1;
            orc_sync_demod(idat, qdat, samples, symbols, &freq, -2, 2, fstep, &shift,
                           lagmin, lagmax, lagstep, &drift, symfac, &sync, 1);
            cand[j].freq = freq; cand[j].shift = shift; cand[j].drift = drift; cand[j].sync = sync;
            if (tr && ipass < ORC_TRACE_PASSES) tr->cand_fine[ipass][j] = cand[j];

            int worth = (sync > minsync1);
            int idt = 0, ii = 0, not_decoded = 1;
            while (worth && not_decoded && idt <= (128 / iifac)) {
                ii = (idt + 1) / 2;
                if (idt % 2 == 1) ii = -ii;
                ii = iifac * ii;
                int jig = shift + ii;
                orc_sync_demod(idat, qdat, samples, symbols, &freq, -2, 2, fstep, &jig,
                               lagmin, lagmax, lagstep, &drift, symfac, &sync, 2);
                float sq = 0.0f;
                for (int i = 0; i < ORC_NSYM; i++) {
                    float y = (float)symbols[i] - 128.0;
                    sq += y * y;
                }
                float rms = sqrtf(sq / (float)ORC_NSYM);
                if (tr && ipass < ORC_TRACE_PASSES) {
                    if (idt == 0) {
                        tr->first_rms[ipass][j] = rms;
                        tr->first_sync2[ipass][j] = sync;
                        memcpy(tr->first_symbols[ipass][j], symbols, ORC_NSYM);
                    }
                    tr->attempts[ipass][j]++;
                }
                if (sync > minsync2 && rms > minrms) {
                    orc_deinterleave(symbols);
                    not_decoded = orc_fano(&metric, &cycles, &maxnp, decdata, symbols, ORC_NBITS,
                                           (const int (*)[256])mettab, delta, maxcycles);
                    if (tr) {
                        tr->fano_cycles_total += cycles;
                        if (ipass < ORC_TRACE_PASSES) tr->fano_calls[ipass][j]++;
                    }
                }
                idt++;
                if (opt.quickmode) break;
            }

            if (worth && !not_decoded) {
                for (int i = 0; i < 11; i++)
                    message[i] = (decdata[i] > 127) ? (signed char)(decdata[i] - 256) : (signed char)decdata[i];
                if (tr && ipass < ORC_TRACE_PASSES) {
                    tr->decoded[ipass][j] = 1;
                    tr->fano_metric[ipass][j] = metric;
                    tr->fano_cycles[ipass][j] = cycles;
                    tr->fano_maxnp[ipass][j] = maxnp;
                    memcpy(tr->decdata[ipass][j], decdata, 11);
                }
                int noprint = orc_unpk(message, hashtab, loctab, call_loc_pow, call, loc, pwr, callsign);
                if (opt.subtraction && ipass == 0 && !noprint) {
                    unsigned char chan[ORC_NSYM];
                    if (orc_channel_symbols(call_loc_pow, hashtab, loctab, chan)) {
                        orc_subtract(idat, qdat, samples, freq, shift, drift, chan);
                        if (tr) tr->subtracted[ipass][j] = 1;
                    } else {
                        stop = 1;         /* wsprd.c:787 leaves the candidate loop */
                        continue;
                    }
                }
                if (!strcmp(loc, "A000AA")) { stop = 1; continue; }   /* wsprd.c:792 */

                int dupe = 0;
                for (int i = 0; i < uniques; i++)
                    if (!strcmp(callsign, allcalls[i]) && fabs(freq - allfreqs[i]) < 3.0) dupe = 1;
                if (!dupe && uniques < ORC_MAXUNIQ) {
                    snprintf(allcalls[uniques], sizeof allcalls[0], "%s", callsign);
                    allfreqs[uniques] = freq;
                    uniques++;
                    double dial = (double)opt.freq / 1e6;
                    orc_spot *o = &spots[uniques - 1];
                    o->sync   = cand[j].sync;
                    o->snr    = cand[j].snr;
                    o->dt     = shift * 1.0 / 375.0 - 2.0;
                    o->freq   = dial + (1500.0 + freq) / 1e6;
                    o->drift  = drift;
                    o->cycles = (int)cycles;
                    o->jitter = ii;
                    snprintf(o->message, sizeof o->message, "%s", call_loc_pow);
                    snprintf(o->call, sizeof o->call, "%s", call);
                    snprintf(o->loc, sizeof o->loc, "%s", loc);
                    snprintf(o->pwr, sizeof o->pwr, "%s", pwr);
                }
            }
        }
    }
    qsort(spots, uniques, sizeof(orc_spot), cmp_spot_snr_desc);
    *n_results = uniques;
    if (opt.usehashtable) {                                   /* wsprd.c:842-852 */
        FILE *fh = fopen("hashtable.txt", "w");
        if (fh) {
            for (int i = 0; i < ORC_HASH_N; i++)
                if (hashtab[i * ORC_HASH_W] != '\0')
                    fprintf(fh, "%5d %s %s\n", i, hashtab + i * ORC_HASH_W, loctab + i * ORC_LOC_W);
            fclose(fh);
        }
    }
    free(ps); free(hashtab); free(loctab);
    return 0;
}

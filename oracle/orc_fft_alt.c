/* ============================================================================
 * orc_fft_alt.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * ALTERNATIVE 512-point FFTs for the robustness study of DESIGN.md section 2
 * (tools/fft_robustness.py, tests/test_fft_robustness.py).
 *
 * Why: the reference's spectrogram comes from FFTW single precision
 * (wsprd/wsprd.c:496-500 plan, :544 execute), an un-vendored system library
 * that is absent here and whose codelet choice (and hence rounding) depends on
 * its version, build flags and CPU.  The product and the default oracle use one
 * particular float32 radix-2 DIF.  Every float32 FFT differs from every other in
 * the last bits of `ps`; the decoder then takes DECISIONS on `ps` (the noise
 * quantile and local maxima of wsprd.c:590-631, the strict-> argmax of :654-667).
 * These variants let the oracle decode the same input with a different FFT's
 * rounding so that the effect on the SPOTS can be counted instead of assumed:
 *
 *   0  radix-2 DIF, float32                    (default: the product's arithmetic)
 *   1  float64 FFT, re/im rounded to float32   (the correctly rounded answer)
 *   2  mixed-radix DIT 4,4,4,4,2, float32
 *   3  mixed-radix DIT 8,8,8, float32
 *   4  mixed-radix DIT 16,32, float32       (the two-codelet shape FFTW favours)
 *   5  mixed-radix DIT 32,16, float32
 *   6  as 2 with fused multiply-add twiddles   (FFTW builds with AVX2/FMA codelets)
 *   7  mixed-radix DIT 2 x 9, float32
 *
 * The small r-point DFT inside a radix-r pass is a radix-2 DIF on r values with
 * the exact r-th roots (+-1, +-i exact; the others rounded from double), which
 * is the arithmetic shape (not the instruction order) of a hard-coded codelet.
 * ==========================================================================*/
#include "wspr_oracle.h"

#include <math.h>
#include <string.h>

static int g_variant = 0;
int  orc_get_fft_variant(void) { return g_variant; }

static float  wf_re[512], wf_im[512];      /* exp(-2 pi i k / 512), rounded from double */
static double wd_re[512], wd_im[512];
static int    tables_ready = 0;

static void tables_init(void) {
    for (int k = 0; k < 512; k++) {
        double a = 2.0 * M_PI * (double)k / 512.0;
        wd_re[k] = cos(a);  wd_im[k] = -sin(a);
    }
    /* exact values on the axes and the diagonals' symmetry, as any table generator gives */
    wd_re[0] = 1;   wd_im[0] = 0;    wd_re[128] = 0;  wd_im[128] = -1;
    wd_re[256] = -1; wd_im[256] = 0; wd_re[384] = 0;  wd_im[384] = 1;
    for (int k = 0; k < 512; k++) { wf_re[k] = (float)wd_re[k]; wf_im[k] = (float)wd_im[k]; }
    tables_ready = 1;
}

int orc_set_fft_variant(int v) {
    if (v < 0 || v > 7) return -1;
    if (!tables_ready) tables_init();
    g_variant = v;
    return 0;
}

/* ---- variant 1: float64 radix-2 DIF, output rounded once ------------------ */
static void fft512_f64(float *re, float *im) {
    double xr[512], xi[512];
    for (int n = 0; n < 512; n++) { xr[n] = re[n]; xi[n] = im[n]; }
    for (int s = 0; s < 9; s++) {
        int half = 256 >> s;
        for (int base = 0; base < 512; base += 2 * half)
            for (int j = 0; j < half; j++) {
                int a = base + j, b = a + half;
                double ur = xr[a], ui = xi[a], vr = xr[b], vi = xi[b];
                double dr = ur - vr, di = ui - vi, wr = wd_re[j << s], wi = wd_im[j << s];
                xr[a] = ur + vr;  xi[a] = ui + vi;
                xr[b] = dr * wr - di * wi;
                xi[b] = dr * wi + di * wr;
            }
    }
    for (unsigned n = 0; n < 512; n++) {
        unsigned k = 0;
        for (int b = 0; b < 9; b++) k |= ((n >> b) & 1u) << (8 - b);
        re[k] = (float)xr[n];  im[k] = (float)xi[n];
    }
}

/* ---- float32 mixed-radix decimation in time, arbitrary power-of-two radices ---- */
/* r-point DFT of (xr, xi) in place, natural order out: radix-2 DIF with the exact r-th roots */
static void small_dft(float *xr, float *xi, int r) {
    int lg = 0;
    while ((1 << lg) < r) lg++;
    for (int s = 0; s < lg; s++) {
        int half = (r >> 1) >> s;
        for (int base = 0; base < r; base += 2 * half)
            for (int j = 0; j < half; j++) {
                int a = base + j, b = a + half;
                float ur = xr[a], ui = xi[a], vr = xr[b], vi = xi[b];
                float dr = ur - vr, di = ui - vi;
                int   t = (j << s) * (512 / r);            /* index into the 512-th roots */
                xr[a] = ur + vr;  xi[a] = ui + vi;
                if (t == 0)        { xr[b] = dr;  xi[b] = di; }
                else if (t == 128) { xr[b] = di;  xi[b] = -dr; }          /* times -i */
                else {
                    float wr = wf_re[t], wi = wf_im[t];
                    float t1 = dr * wr, t2 = di * wi, t3 = dr * wi, t4 = di * wr;
                    xr[b] = t1 - t2;  xi[b] = t3 + t4;
                }
            }
    }
    float tr[32], ti[32];
    memcpy(tr, xr, sizeof(float) * r);  memcpy(ti, xi, sizeof(float) * r);
    for (int n = 0; n < r; n++) {
        int k = 0;
        for (int b = 0; b < lg; b++) k |= ((n >> b) & 1) << (lg - 1 - b);
        xr[k] = tr[n];  xi[k] = ti[n];
    }
}

/* Mixed-radix decimation in time, out of place (Stockham-style ping-pong), natural order in and out.
 * Pass p combines r = radix[p] finished transforms of length L (L = product of the earlier radices) into one of
 * length L*r:   X[k + L*j] = sum_c W_r^(c j) * ( W_(L r)^(c k) * Y_c[k] ),   Y_c = DFT_L of the c-th interleaved
 * sub-sequence.  A "problem" is the sample set {x[off + stride*n]}; with M = 512/(L r) problems after the pass,
 * problem q (offset q, stride M) is made of the source problems of offsets q + M*c, c < r.  Problems are stored one
 * after the other, bins contiguous: src[(q + M c)*L + k] -> dst[q*(L r) + k + L j].  At the start (L = 1) problem P
 * is the single sample x[P], i.e. the input in natural order: no digit-reversal pass is needed. */
static void mixed_radix_dit(float *re, float *im, const int *radix, int npass, int use_fma) {
    float ar[512], ai[512], br[512], bi[512];
    memcpy(ar, re, sizeof ar);  memcpy(ai, im, sizeof ai);
    float *sr = ar, *si = ai, *dr = br, *di = bi;
    int L = 1;
    for (int p = 0; p < npass; p++) {
        int r = radix[p], M = 512 / (L * r);
        for (int q = 0; q < M; q++)
            for (int k = 0; k < L; k++) {
                float xr[32], xi[32];
                for (int c = 0; c < r; c++) {
                    int   s_idx = (q + M * c) * L + k;
                    float vr = sr[s_idx], vi = si[s_idx];
                    int   t = (c * k * M) & 511;               /* exp(-2 pi i c k / (L r)) = W_512^(c k M) */
                    if (t == 0) { xr[c] = vr; xi[c] = vi; }
                    else {
                        float wr = wf_re[t], wi = wf_im[t];
                        if (use_fma) {
                            xr[c] = fmaf(vr, wr, -(vi * wi));
                            xi[c] = fmaf(vr, wi, vi * wr);
                        } else {
                            float t1 = vr * wr, t2 = vi * wi, t3 = vr * wi, t4 = vi * wr;
                            xr[c] = t1 - t2;  xi[c] = t3 + t4;
                        }
                    }
                }
                small_dft(xr, xi, r);
                for (int j = 0; j < r; j++) {
                    int d_idx = q * (L * r) + k + L * j;
                    dr[d_idx] = xr[j];  di[d_idx] = xi[j];
                }
            }
        float *t;
        t = sr; sr = dr; dr = t;
        t = si; si = di; di = t;
        L *= r;
    }
    memcpy(re, sr, sizeof ar);  memcpy(im, si, sizeof ai);
}

static const int kRadix[8][9] = {
    {0}, {0},
    {4, 4, 4, 4, 2}, {8, 8, 8}, {16, 32}, {32, 16}, {4, 4, 4, 4, 2}, {2, 2, 2, 2, 2, 2, 2, 2, 2}};
static const int kNpass[8] = {0, 0, 5, 3, 2, 2, 5, 9};

/* Called by orc_fft_bank (orc_dsp.c) when the variant is not 0. */
void orc_fft512_variant(float *re, float *im) {
    if (g_variant == 1) fft512_f64(re, im);
    else                mixed_radix_dit(re, im, kRadix[g_variant], kNpass[g_variant], g_variant == 6);
}

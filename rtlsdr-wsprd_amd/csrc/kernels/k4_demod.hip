// K4/K5 -- fine time/frequency sync search and soft-symbol demodulator.
//
// Replaces reference sync_and_demodulate(), wsprd/wsprd.c:101-259, in its three
// modes (0: lag scan, 1: frequency scan, 2: soft symbols at a jittered lag).
//
// Mapping (north star: "one workgroup per (lag, df) candidate"): grid =
// (hypothesis, candidate); one lane per WSPR symbol (162 of 192 lanes).  A lane
// runs the four tone phasors as float recurrences and the 256-sample matched-
// filter accumulation serially, in the reference's operation order, so each of
// its eight accumulators is bit-identical to the reference's.  The cross-symbol
// sums (sync metric, soft-symbol normalisation) are formed by one lane in symbol
// order.  Phasor seeds come from glibc-exact sinf/cosf (glibc_sincosf.h).
// Bound: fp32 VALU (AI > 100 flop/B); no MFMA (separately rounded mul/add chains).
#include "wspr_device.h"
#include "glibc_sincosf.h"

#pragma clang fp contract(off)

namespace wspr {
namespace {

constexpr double kTwoPiDt = 2.0 * 3.14159265358979323846 * 1.0 / 375.0;   // TWOPIDT
constexpr double kDf05 = 375.0 / 256.0 * 0.5;
constexpr double kDf15 = 375.0 / 256.0 * 1.5;

__global__ __launch_bounds__(192)
void demod_kernel(const float* __restrict__ dI, const float* __restrict__ dQ, int np,
                  const FineState* __restrict__ items, int mode, int nlag, int lagstep,
                  int ifmin, float fstep, const int* __restrict__ jitter, float minsync1,
                  float* __restrict__ sync_out, unsigned char* __restrict__ sym_out,
                  float* __restrict__ rms_out, const unsigned char* __restrict__ pr3) {
    __shared__ float pw[kNSymD][4];
    const int item = blockIdx.y, hyp = blockIdx.x;
    const FineState st = items[item];

    float f0;
    int lag;
    if (mode == 0) {
        f0 = st.freq_coarse;
        lag = st.shift_coarse - 128 + lagstep * hyp;
    } else if (mode == 1) {
        f0 = st.freq + (float)(ifmin + hyp) * fstep;      // *freq + ifreq * fstep, wsprd.c:151
        lag = st.shift;
    } else {
        if (!(st.sync > minsync1)) return;              // not worth a try (wsprd.c:733-737)
        f0 = st.freq;
        lag = st.shift + jitter[hyp];
    }

    const int i = threadIdx.x;
    if (i < kNSymD) {
        const float* __restrict__ xi = dI + (size_t)st.seg * kIqStride;
        const float* __restrict__ xq = dQ + (size_t)st.seg * kIqStride;
        const float fp = (float)((double)f0 + ((double)st.drift / 2.0) * (double)((float)i - 81.0f) / (double)81.0f);
        const double fpd = (double)fp;
        const float dphi[4] = {(float)(kTwoPiDt * (fpd - kDf15)), (float)(kTwoPiDt * (fpd - kDf05)),
                               (float)(kTwoPiDt * (fpd + kDf05)), (float)(kTwoPiDt * (fpd + kDf15))};
        float cd[4], sd[4], c[4], s[4], ai[4], aq[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            cd[t] = glibc_cosf(dphi[t]);
            sd[t] = glibc_sinf(dphi[t]);
            c[t] = 1.0f; s[t] = 0.0f; ai[t] = 0.0f; aq[t] = 0.0f;
        }
        const int base = lag + kSps * i;
        for (int j0 = 0; j0 < kSps; j0 += 4) {
            float x[4], y[4];
            const int k0 = base + j0;
            if (k0 > 0 && k0 + 3 < np) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { x[u] = xi[k0 + u]; y[u] = xq[k0 + u]; }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + u;
                    const bool ok = (k > 0) && (k < np);
                    x[u] = ok ? xi[k] : 0.0f;
                    y[u] = ok ? xq[k] : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u;
                if (j0 + u > 0) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float a = c[t] * cd[t], b = s[t] * sd[t];
                        const float e = c[t] * sd[t], d = s[t] * cd[t];
                        c[t] = a - b;
                        s[t] = e + d;
                    }
                }
                if (k > 0 && k < np) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float m1 = x[u] * c[t], m2 = y[u] * s[t];
                        const float m3 = x[u] * s[t], m4 = y[u] * c[t];
                        ai[t] = (ai[t] + m1) + m2;
                        aq[t] = (aq[t] - m3) + m4;
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float e1 = ai[t] * ai[t], e2 = aq[t] * aq[t];
            pw[i][t] = sqrtf(e1 + e2);
        }
    }
    __syncthreads();

    if (threadIdx.x == 0) {
        float ss = 0.0f, totp = 0.0f;
        for (int k = 0; k < kNSymD; ++k) {
            const float p0 = pw[k][0], p1 = pw[k][1], p2 = pw[k][2], p3 = pw[k][3];
            totp = totp + p0 + p1 + p2 + p3;
            const float cmet = (p1 + p3) - (p0 + p2);
            ss = pr3[k] ? ss + cmet : ss - cmet;
        }
        ss = ss / totp;
        const size_t o = (size_t)item * nlag + hyp;
        if (mode != 2) {
            sync_out[o] = ss;
        } else {
            sync_out[o] = (ss > -1e30f) ? ss : -1e30f;
            // soft symbols, wsprd.c:219-225 and :243-256
            float fsum = 0.0f, f2sum = 0.0f;
            for (int k = 0; k < kNSymD; ++k) {
                const float f = pr3[k] ? pw[k][3] - pw[k][1] : pw[k][2] - pw[k][0];
                fsum += f / 162.0f;
                const float ff = f * f;
                f2sum += ff / 162.0f;
            }
            const float m2 = fsum * fsum;
            const float fac = sqrtf(f2sum - m2);
            float sq = 0.0f;
            unsigned char* __restrict__ so = sym_out + o * kNSymD;
            for (int k = 0; k < kNSymD; ++k) {
                const float f = pr3[k] ? pw[k][3] - pw[k][1] : pw[k][2] - pw[k][0];
                float v = 50.0f * f / fac;
                if (v > 127.0f) v = 127.0f;
                if (v < -128.0f) v = -128.0f;
                const float w = v + 128.0f;
                const unsigned char b = (w == w) ? (unsigned char)(int)w : (unsigned char)0;
                so[k] = b;
                const float y = (float)b - 128.0f;
                sq += y * y;
            }
            rms_out[o] = sqrtf(sq / 162.0f);
        }
    }
}

// mode 0 epilogue: first lag (in scan order) with the strictly largest metric
__global__ void pick_lag_kernel(FineState* __restrict__ items, int nitems,
                                const float* __restrict__ sync_in, int nlag, int lagstep) {
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= nitems) return;
    FineState st = items[it];
    float best = -1e30f, fbest = 0.0f;
    int bshift = 0;
    for (int m = 0; m < nlag; ++m) {
        const float v = sync_in[(size_t)it * nlag + m];
        if (v > best) { best = v; bshift = st.shift_coarse - 128 + lagstep * m; fbest = st.freq_coarse; }
    }
    st.shift = bshift;
    st.freq = fbest;
    st.sync = best;
    items[it] = st;
}

// mode 1 epilogue: first frequency with the strictly largest metric
__global__ void pick_freq_kernel(FineState* __restrict__ items, int nitems,
                                 const float* __restrict__ sync_in, int nfreq, int ifmin, float fstep) {
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= nitems) return;
    FineState st = items[it];
    const float fin = st.freq;                 // == freq_coarse unless mode 0 found nothing
    float best = -1e30f, fbest = 0.0f;
    int bshift = 0;
    for (int q = 0; q < nfreq; ++q) {
        const float v = sync_in[(size_t)it * nfreq + q];
        if (v > best) { best = v; fbest = fin + (float)(ifmin + q) * fstep; bshift = st.shift; }
    }
    st.freq = fbest;
    st.shift = bshift;
    st.sync = best;
    items[it] = st;
}
}  // namespace

void launch_demod(const float* dI, const float* dQ, int samples, const FineState* items, int nitems,
                  int mode, int nlag, int lagstep, int ifmin, float fstep, const int* jitter,
                  float minsync1, float* sync_out, unsigned char* sym_out, float* rms_out,
                  const DeviceTables& t, hipStream_t st) {
    if (nitems <= 0 || nlag <= 0) return;
    hipLaunchKernelGGL(demod_kernel, dim3(nlag, nitems), dim3(192), 0, st, dI, dQ, samples, items, mode,
                       nlag, lagstep, ifmin, fstep, jitter, minsync1, sync_out, sym_out, rms_out, t.sync);
}
void launch_pick_lag(FineState* items, int nitems, const float* sync_in, int nlag, int lagstep, hipStream_t st) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(pick_lag_kernel, dim3((nitems + 63) / 64), dim3(64), 0, st, items, nitems, sync_in, nlag, lagstep);
}
void launch_pick_freq(FineState* items, int nitems, const float* sync_in, int nfreq, int ifmin,
                      float fstep, hipStream_t st) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(pick_freq_kernel, dim3((nitems + 63) / 64), dim3(64), 0, st, items, nitems, sync_in,
                       nfreq, ifmin, fstep);
}

}  // namespace wspr

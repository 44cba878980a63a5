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
//
// Kernels in this file, by the stage they serve (every one performs demod_kernel's operations per accumulator):
//   general               demod_kernel                 any mode, any drift; exported sync_and_demodulate()
//   mode 0, no drift      demod_lagsys_kernel          a strided correlation, samples in registers and handed lane to lane
//   mode 0, drift         demod_drift_kernel           three lags per lane, per-symbol tables in an LDS ring
//   mode 0 (quick mode), ladder rungs (mode 2, 43 lags)
//                         phasor_table_kernel + demod_tile_kernel<STEP, shared> + demod_metric_kernel
//   mode 1 + rung 0, no drift   phasor_freq_kernel + freq_scalar_kernel (freq_tile_kernel: tables in LDS) + freq_metric_kernel
//   mode 1 + rung 0, drift      freq_drift_kernel + freq_metric_kernel
//   epilogues             pick_lag_kernel, pick_freq_kernel;  calibration: calib_valu_kernel
#include "wspr_device.h"
#include "glibc_sincosf.h"
#include <cstdlib>

#pragma clang fp contract(off)

namespace wspr {
namespace {

constexpr double kTwoPiDt = 2.0 * 3.14159265358979323846 * 1.0 / 375.0;   // TWOPIDT
constexpr double kDf05 = 375.0 / 256.0 * 0.5;
constexpr double kDf15 = 375.0 / 256.0 * 1.5;

constexpr int kGenThreads = 192;
constexpr int kGenChunk = 32;
constexpr int kGenPerThread = kNSymD * kGenChunk / kGenThreads;     // 27 samples staged per thread and chunk
static_assert(kNSymD * kGenChunk % kGenThreads == 0, "chunk must split evenly over the workgroup");

__global__ __launch_bounds__(kGenThreads)
void demod_kernel(const float* __restrict__ dI, const float* __restrict__ dQ, int np,
                  const FineState* __restrict__ items, const int* __restrict__ item_list, int mode,
                  int nlag, int lagstep, int ifmin, float fstep, const int* __restrict__ jitter,
                  float minsync1, float* __restrict__ sync_out, unsigned char* __restrict__ sym_out,
                  float* __restrict__ rms_out, const unsigned char* __restrict__ pr3, float symfac) {
    __shared__ float pw[kNSymD][4];
    __shared__ float2 tile[kNSymD][kGenChunk + 1];
    const int item = item_list ? item_list[blockIdx.y] : (int)blockIdx.y, hyp = blockIdx.x;
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

    // lane = symbol.  The samples stream through LDS in chunks of 32 per symbol (coalesced 128-byte row
    // segments from HBM/L2, transposed so that lane = symbol reads conflict-free), the next chunk in flight
    // in registers while the current one is consumed; the four tone phasors are the reference's float
    // recurrences, run inline.
    const int i = threadIdx.x, tid = threadIdx.x;
    const float* __restrict__ xi = dI + (size_t)st.seg * kIqStride;
    const float* __restrict__ xq = dQ + (size_t)st.seg * kIqStride;
    float cd[4], sd[4], c[4], s[4], ai[4], aq[4];
    if (i < kNSymD) {
        const float fp = (float)((double)f0 + ((double)st.drift / 2.0) * (double)((float)i - 81.0f) / (double)81.0f);
        const double fpd = (double)fp;
        const float dphi[4] = {(float)(kTwoPiDt * (fpd - kDf15)), (float)(kTwoPiDt * (fpd - kDf05)),
                               (float)(kTwoPiDt * (fpd + kDf05)), (float)(kTwoPiDt * (fpd + kDf15))};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            cd[t] = glibc_cosf(dphi[t]);
            sd[t] = glibc_sinf(dphi[t]);
            c[t] = 1.0f; s[t] = 0.0f; ai[t] = 0.0f; aq[t] = 0.0f;
        }
    }
    float2 nxt[kGenPerThread];
    auto fetch = [&](int ch) {
#pragma unroll
        for (int u = 0; u < kGenPerThread; ++u) {
            const int e = u * kGenThreads + tid, row = e >> 5, col = e & (kGenChunk - 1);
            const int k = lag + kSps * row + kGenChunk * ch + col;
            const bool ok = (k > 0) && (k < np);
            nxt[u] = ok ? make_float2(xi[k], xq[k]) : make_float2(0.0f, 0.0f);
        }
    };
    fetch(0);
    for (int ch = 0; ch < kSps / kGenChunk; ++ch) {
        __syncthreads();                                             // the previous chunk has been consumed
#pragma unroll
        for (int u = 0; u < kGenPerThread; ++u) {
            const int e = u * kGenThreads + tid;
            tile[e >> 5][e & (kGenChunk - 1)] = nxt[u];
        }
        __syncthreads();
        if (ch + 1 < kSps / kGenChunk) fetch(ch + 1);
        if (i < kNSymD) {
            const int base = lag + kSps * i + kGenChunk * ch;
#pragma unroll 4
            for (int jj = 0; jj < kGenChunk; ++jj) {
                const int k = base + jj;
                if (kGenChunk * ch + jj > 0) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float a = c[t] * cd[t], b = s[t] * sd[t];
                        const float e = c[t] * sd[t], d = s[t] * cd[t];
                        c[t] = a - b;
                        s[t] = e + d;
                    }
                }
                if (k > 0 && k < np) {
                    const float2 xy = tile[i][jj];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float m1 = xy.x * c[t], m2 = xy.y * s[t];
                        const float m3 = xy.x * s[t], m4 = xy.y * c[t];
                        ai[t] = (ai[t] + m1) + m2;
                        aq[t] = (aq[t] - m3) + m4;
                    }
                }
            }
        }
    }
    if (i < kNSymD) {
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
                float v = symfac * f / fac;                 // symfac * fsymb[i] / fac, wsprd.c:250 (int -> float)
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

// =============================================================================
// Tiled variant for the two wide searches (mode 0: 33/17 lags; ladder rungs: 43 lags).
//
// The per-(candidate, symbol) phasor tables do not depend on the lag, and adjacent
// lags read overlapping samples, so:
//   phasor_table_kernel  builds each distinct table once (4 tones x 256 steps of the
//                        reference's float recurrence) -> HBM, 8 KB per table;
//   demod_tile_kernel    one workgroup per (candidate, 6 consecutive symbols): stages
//                        the tables and the 6*256 + span samples it needs in LDS (samples
//                        transposed by the lag step so that the lanes of a symbol read
//                        consecutive LDS words), then lane = (symbol, lag) runs the same
//                        256-step matched-filter sums as demod_kernel, in the same order;
//   demod_metric_kernel  one lane per (candidate, lag) folds the 162 symbols in order.
// Same arithmetic per accumulator as demod_kernel => identical bits; ~8x less time
// because loads are coalesced/LDS-served and tables are not recomputed per lag.
// Packed-pair arithmetic: gfx950 executes v_pk_mul_f32 / v_pk_add_f32 on register pairs; each
// half is an ordinary IEEE multiply or add (no fusion), so the per-accumulator operation
// sequence -- and therefore every bit -- is unchanged.  Tones (0,1) and (2,3) share a pair.
typedef float v2f __attribute__((ext_vector_type(2)));

struct ToneAcc {
    v2f i01, i23, q01, q23;
    __device__ __forceinline__ void clear() { i01 = i23 = q01 = q23 = (v2f){0.0f, 0.0f}; }
    // ai = (ai + x*c) + y*s ; aq = (aq - x*s) + y*c   (wsprd.c:200-207)
    __device__ __forceinline__ void step(const float2 d, const float4 c4, const float4 s4) {
        const v2f xx = {d.x, d.x}, yy = {d.y, d.y};
        const v2f c01 = {c4.x, c4.y}, c23 = {c4.z, c4.w}, s01 = {s4.x, s4.y}, s23 = {s4.z, s4.w};
        i01 = (i01 + xx * c01) + yy * s01;
        i23 = (i23 + xx * c23) + yy * s23;
        q01 = (q01 - xx * s01) + yy * c01;
        q23 = (q23 - xx * s23) + yy * c23;
    }
    __device__ __forceinline__ float4 amplitudes() const {
        const v2f e01 = i01 * i01 + q01 * q01, e23 = i23 * i23 + q23 * q23;
        return make_float4(sqrtf(e01.x), sqrtf(e01.y), sqrtf(e23.x), sqrtf(e23.y));
    }
};

constexpr int kTileSymsShared = 9;    // 9 x 33 lags = 297 of 320 lanes; 28 KB of LDS
constexpr int kTileSymsOwn = 6;       // per-symbol tables: 6 x 8 KB + tile
// A wave of the per-symbol-table variant spans two or three symbols, whose tables are read at the same step
// j: with a pitch of exactly 8 KB those reads fall on the same LDS banks (measured: bank-conflict cycles 1.5x
// the LDS cycles of the kernel).  Two extra 16-byte words per table move consecutive symbols 8 banks apart.
constexpr int kOwnTabPitch = 512 + 2;           // float4 words per table in LDS

__global__ __launch_bounds__(64)
void phasor_table_kernel(const FineState* __restrict__ items, int mode, float* __restrict__ tabs) {
    const int item = blockIdx.y;
    const FineState st = items[item];
    const int lane = threadIdx.x;
    const int sym = blockIdx.x * 16 + (lane >> 2), tone = lane & 3;
    // drifting candidates build their per-symbol tables inside demod_tile_kernel<., false>
    if (st.drift != 0.0f || sym != 0) return;
    const float f0 = (mode == 0) ? st.freq_coarse : st.freq;
    const float fp = (float)((double)f0 + ((double)st.drift / 2.0) * (double)((float)sym - 81.0f) / (double)81.0f);
    const double off = (tone == 0) ? -kDf15 : (tone == 1) ? -kDf05 : (tone == 2) ? kDf05 : kDf15;
    const float dphi = (float)(kTwoPiDt * ((double)fp + off));
    const float cd = glibc_cosf(dphi), sd = glibc_sinf(dphi);
    float* __restrict__ t = tabs + (size_t)st.pad * 2048;   // [256][8]
    float c = 1.0f, s = 0.0f;
    for (int j = 0; j < kSps; ++j) {
        if (j > 0) {
            const float a = c * cd, b = s * sd, e = c * sd, d = s * cd;
            c = a - b;
            s = e + d;
        }
        t[8 * j + tone] = c;
        t[8 * j + 4 + tone] = s;
    }
}

template <int STEP, bool SHARED>
__global__ __launch_bounds__(448)
void demod_tile_kernel(const float* __restrict__ dI, const float* __restrict__ dQ, int np,
                       const FineState* __restrict__ items, const int* __restrict__ item_list, int mode,
                       int nlag, float minsync1, const float* __restrict__ tabs, float4* __restrict__ pw_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int item = item_list[blockIdx.y];
    const FineState st = items[item];
    if (mode == 2 && !(st.sync > minsync1)) return;
    constexpr int kTileSyms = SHARED ? kTileSymsShared : kTileSymsOwn;
    const int i0 = blockIdx.x * kTileSyms, tid = threadIdx.x;
    const int lag0 = (mode == 0) ? st.shift_coarse - 128 : st.shift - 63;
    float4* tab = reinterpret_cast<float4*>(smem);
    float2* tile = reinterpret_cast<float2*>(smem + (SHARED ? 0 : kTileSyms * kOwnTabPitch * 16));
    const int span = kSps * kTileSyms + STEP * (nlag - 1);
    const int pitch = (span + STEP - 1) / STEP + 1;

    const float* __restrict__ xi = dI + (size_t)st.seg * kIqStride;
    const float* __restrict__ xq = dQ + (size_t)st.seg * kIqStride;
    const int kbase = lag0 + kSps * i0;
    const int nthr = blockDim.x;
    // the candidate's one table (built by phasor_table_kernel): every lane of the workgroup reads the same
    // entry at every step, so it is read through the scalar cache straight into SGPR operands of the packed
    // multiplies (an LDS copy costs two 16-byte LDS reads per lane and step: 2.00 vs 1.91 ms per 2 048 candidates)
    const float4* __restrict__ gtab = reinterpret_cast<const float4*>(tabs) +
                                      (size_t)__builtin_amdgcn_readfirstlane(st.pad) * 512;
    if constexpr (!SHARED) {
        // a drifting candidate has one table per symbol (162 x 8 KB): its 6 x 4 phasor recurrences are run
        // here by the first 24 lanes, straight into LDS, instead of travelling through HBM (2.6 MB per
        // candidate written and read back); same operations as phasor_table_kernel (wsprd.c:158-188)
        if (tid < kTileSyms * 4) {
            const int il = tid >> 2, tone = tid & 3, sym = i0 + il;
            const float f0 = (mode == 0) ? st.freq_coarse : st.freq;
            const float fp = (float)((double)f0 + ((double)st.drift / 2.0) * (double)((float)sym - 81.0f) / (double)81.0f);
            const double off = (tone == 0) ? -kDf15 : (tone == 1) ? -kDf05 : (tone == 2) ? kDf05 : kDf15;
            const float dphi = (float)(kTwoPiDt * ((double)fp + off));
            const float cd = glibc_cosf(dphi), sd = glibc_sinf(dphi);
            float* __restrict__ t = reinterpret_cast<float*>(tab + il * kOwnTabPitch);     // [256][8]
            float c = 1.0f, s = 0.0f;
            for (int j = 0; j < kSps; ++j) {
                if (j > 0) {
                    const float a = c * cd, b = s * sd, e = c * sd, d = s * cd;
                    c = a - b;
                    s = e + d;
                }
                t[8 * j + tone] = c;
                t[8 * j + 4 + tone] = s;
            }
        }
    }
    for (int e0 = tid; e0 < span; e0 += 4 * nthr) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * nthr, k = kbase + e;
            const bool ok = (e < span) && (k > 0) && (k < np);   // wsprd.c:199; zero-fill == skip (x*c = 0 adds exactly)
            v[u] = ok ? make_float2(xi[k], xq[k]) : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * nthr;
            if (e < span) tile[(e % STEP) * pitch + e / STEP] = v[u];
        }
    }
    __syncthreads();

    const int il = tid / nlag, m = tid - il * nlag;
    if (il >= kTileSyms) return;
    const float4* __restrict__ tb = tab + (SHARED ? 0 : il * kOwnTabPitch);
    ToneAcc acc;
    acc.clear();
    if constexpr (STEP == 8 || STEP == 16) {
        // e = STEP*m + 256*il + j with STEP | 256: row = j % STEP, column = m + (256/STEP)*il + j/STEP
        const float2* __restrict__ col = tile + m + (kSps / STEP) * il;
        if (SHARED) {
            for (int j0 = 0; j0 < kSps; j0 += STEP) {
                const float2* __restrict__ t0 = col + j0 / STEP;
#pragma unroll
                for (int u = 0; u < STEP; ++u) acc.step(t0[u * pitch], gtab[2 * (j0 + u)], gtab[2 * (j0 + u) + 1]);
            }
        } else {
            for (int j0 = 0; j0 < kSps; j0 += STEP) {
                const float2* __restrict__ t0 = col + j0 / STEP;
#pragma unroll
                for (int u = 0; u < STEP; ++u) acc.step(t0[u * pitch], tb[2 * (j0 + u)], tb[2 * (j0 + u) + 1]);
            }
        }
    } else {
        const int e0 = STEP * m + kSps * il;
        if (SHARED) {
#pragma unroll 8
            for (int j = 0; j < kSps; ++j) {
                const int e = e0 + j;
                acc.step(tile[(e % STEP) * pitch + e / STEP], gtab[2 * j], gtab[2 * j + 1]);
            }
        } else {
#pragma unroll 8
            for (int j = 0; j < kSps; ++j) {
                const int e = e0 + j;
                acc.step(tile[(e % STEP) * pitch + e / STEP], tb[2 * j], tb[2 * j + 1]);
            }
        }
    }
    pw_out[((size_t)item * nlag + m) * kNSymD + i0 + il] = acc.amplitudes();
}

// -----------------------------------------------------------------------------
// Lag scan (mode 0: 33 lags, step 8) of a DRIFTING candidate.
//
// A drifting candidate has one phasor table per symbol, so the table operand differs between the symbols a
// wave holds and cannot come from scalar registers.  With one lag per lane (demod_tile_kernel<8, false>) the
// two 16-byte table reads per lane and step make the kernel LDS-return-bound (20 LDS clocks per wave step,
// four waves per LDS, against 16 packed VALU instructions = 64 clocks).  Here a lane runs THREE lags of one
// symbol (lane = (symbol, g), lags g, g+11, g+22): the table entry is read once per step and serves three
// 16-instruction accumulator updates, so the kernel is VALU-bound again.
//   * workgroup = ONE wave = 5 symbols x 11 lanes (55 of 64 lanes); 33 workgroups per candidate.
//   * the tables are not staged whole (8 KB per symbol): lanes 0..19 carry the 5 x 4 phasor recurrences
//     (wsprd.c:158-188) in registers and emit them 32 steps at a time into a 5 KB LDS ring; the workgroup's
//     LDS is 22 KB, so seven waves share a CU.
//   * samples: one tile for the 5 symbols, transposed by the lag step (lanes of a symbol read consecutive
//     8-byte words) and skewed by 11 words per 256 samples, so that the three symbols of a half wave fall on
//     different banks (symbol stride 32 + 11 = 43 = 11 mod 32 words).
// Same operations per accumulator as demod_kernel => identical bits.
constexpr int kDrSyms = 5;
constexpr int kDrGroups = 11;                       // lanes per symbol; 3 lags each
constexpr int kDrChunk = 32;                        // table steps per ring fill
constexpr int kDrSpan = kSps * kDrSyms + 8 * 32;    // samples the 5 symbols x 33 lags touch: 1536
constexpr int kDrSkew = 11;
constexpr int kDrPitch = kDrSpan / 8 + kDrSkew * (kDrSpan / 256) + 1;      // 8-byte words per tile row
constexpr int kDrTabStride = 2 * kDrChunk + 2;      // float4 words per symbol in the ring (+32 B: bank offset 8 per symbol)

__device__ __forceinline__ int drift_tile_word(int e) { return (e & 7) * kDrPitch + (e >> 3) + kDrSkew * (e >> 8); }

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 3)))
void demod_drift_kernel(const float* __restrict__ dI, const float* __restrict__ dQ, int np,
                        const FineState* __restrict__ items, const int* __restrict__ item_list,
                        float4* __restrict__ pw_out) {
    __shared__ float2 tile[8 * kDrPitch];
    __shared__ float4 tab[kDrSyms * kDrTabStride];
    constexpr int nlag = 3 * kDrGroups;
    const int item = item_list[blockIdx.y];
    const FineState st = items[item];
    const int i0 = blockIdx.x * kDrSyms, lane = threadIdx.x;
    const float* __restrict__ xi = dI + (size_t)st.seg * kIqStride;
    const float* __restrict__ xq = dQ + (size_t)st.seg * kIqStride;
    const int kbase = st.shift_coarse - 128 + kSps * i0;
#pragma unroll
    for (int e0 = 0; e0 < kDrSpan; e0 += 64 * 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 64 * u + lane, k = kbase + e;
            const bool ok = (k > 0) && (k < np);             // wsprd.c:199; zero-fill == skip (x*c = 0 adds exactly)
            v[u] = ok ? make_float2(xi[k], xq[k]) : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) tile[drift_tile_word(e0 + 64 * u + lane)] = v[u];
    }
    // table lanes: (symbol, tone) recurrences, seeds as in phasor_table_kernel
    const int bil = lane >> 2, tone = lane & 3;
    const bool builder = lane < kDrSyms * 4;
    float cd = 1.0f, sd = 0.0f, pc = 1.0f, ps = 0.0f;
    if (builder) {
        const int sym = i0 + bil;
        const float fp = (float)((double)st.freq_coarse + ((double)st.drift / 2.0) * (double)((float)sym - 81.0f) / (double)81.0f);
        const double off = (tone == 0) ? -kDf15 : (tone == 1) ? -kDf05 : (tone == 2) ? kDf05 : kDf15;
        const float dphi = (float)(kTwoPiDt * ((double)fp + off));
        cd = glibc_cosf(dphi);
        sd = glibc_sinf(dphi);
    }
    float* __restrict__ tabw = reinterpret_cast<float*>(tab + bil * kDrTabStride) + tone;

    const int il = lane / kDrGroups, g = lane - il * kDrGroups;
    const bool working = (lane < kDrSyms * kDrGroups) && (i0 + il < kNSymD);
    const float4* __restrict__ tb = tab + (working ? il : 0) * kDrTabStride;
    const int e_lane = 8 * g + kSps * (working ? il : 0);
    ToneAcc acc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) acc[r].clear();
    for (int ch = 0; ch < kSps / kDrChunk; ++ch) {
        __syncthreads();                                     // the previous 32 table steps have been consumed
        if (builder) {
            for (int jj = 0; jj < kDrChunk; ++jj) {
                if (ch + jj > 0) {
                    const float a = pc * cd, b = ps * sd, e = pc * sd, d = ps * cd;
                    pc = a - b;
                    ps = e + d;
                }
                tabw[8 * jj] = pc;
                tabw[8 * jj + 4] = ps;
            }
        }
        __syncthreads();
        if (working) {
#pragma unroll 1
            for (int jb = 0; jb < kDrChunk; jb += 8) {
                const float2* __restrict__ t0[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int e = e_lane + 8 * kDrGroups * r + kDrChunk * ch + jb;      // a multiple of 8
                    t0[r] = tile + (e >> 3) + kDrSkew * (e >> 8);
                }
                // operands of step u+1 are in flight while step u is consumed; the 24 products of a step are
                // formed before the 24 accumulator updates, so dependent instructions sit far apart
                float4 c4 = tb[2 * jb], s4 = tb[2 * jb + 1];
                float2 d[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) d[r] = t0[r][0];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    float4 cn = c4, sn = s4;
                    float2 dn[3] = {d[0], d[1], d[2]};
                    if (u + 1 < 8) {
                        cn = tb[2 * (jb + u + 1)];
                        sn = tb[2 * (jb + u + 1) + 1];
#pragma unroll
                        for (int r = 0; r < 3; ++r) dn[r] = t0[r][(u + 1) * kDrPitch];
                    }
                    const v2f c01 = {c4.x, c4.y}, c23 = {c4.z, c4.w}, s01 = {s4.x, s4.y}, s23 = {s4.z, s4.w};
                    v2f p[3][8];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const v2f xx = {d[r].x, d[r].x}, yy = {d[r].y, d[r].y};
                        p[r][0] = xx * c01; p[r][1] = xx * c23; p[r][2] = xx * s01; p[r][3] = xx * s23;
                        p[r][4] = yy * s01; p[r][5] = yy * s23; p[r][6] = yy * c01; p[r][7] = yy * c23;
                    }
#pragma unroll
                    for (int r = 0; r < 3; ++r) {           // ai = (ai + x*c) + y*s ; aq = (aq - x*s) + y*c (wsprd.c:200-207)
                        acc[r].i01 = acc[r].i01 + p[r][0]; acc[r].i23 = acc[r].i23 + p[r][1];
                        acc[r].q01 = acc[r].q01 - p[r][2]; acc[r].q23 = acc[r].q23 - p[r][3];
                    }
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        acc[r].i01 = acc[r].i01 + p[r][4]; acc[r].i23 = acc[r].i23 + p[r][5];
                        acc[r].q01 = acc[r].q01 + p[r][6]; acc[r].q23 = acc[r].q23 + p[r][7];
                    }
                    c4 = cn; s4 = sn;
#pragma unroll
                    for (int r = 0; r < 3; ++r) d[r] = dn[r];
                }
            }
        }
    }
    if (working) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
            pw_out[((size_t)item * nlag + g + kDrGroups * r) * kNSymD + i0 + il] = acc[r].amplitudes();
    }
}

// -----------------------------------------------------------------------------
// Lag scan (mode 0: 33 lags, step 8) of a DRIFT-FREE candidate as ONE strided correlation, samples in registers.
//
// With one table for the whole frame the accumulator of (symbol s, lag m) is
//     A[u] = sum_j x[k0 + 8 u + j] * tab[j],   u = 32 s + m,   k0 = shift_coarse - 128
// (wsprd.c:190-207: k = lag + 256 s + j with lag = k0 + 8 m): a 256-tap correlation of the sample stream evaluated
// every 8 samples.  Two consequences:
//   * (s, 32) and (s + 1, 0) are the SAME sum over the same samples in the same order, so only u = 0 .. 5184 are
//     distinct (5 185 sums instead of 33 x 162 = 5 346) and the lag-32 row of the output is a copy;
//   * output u + 1 at step j reads the sample output u reads at step j + 8.  A lane that owns ROWS consecutive
//     outputs keeps their current 8-sample vectors in registers; after 8 steps row r takes over row r + 1's
//     vector (a register renaming: the loop is unrolled over ROWS + 1 blocks), and the last row takes the first
//     row of the NEXT lane with one DPP move per register (v_mov_b32_dpp wave_shl:1).  Lane 63's successor lies
//     outside the wave: its vector (8 new samples per 8 steps for the whole wave) is loaded from memory a block
//     ahead by every lane at the same address and enters through the DPP move's "old" operand.
// No LDS and no barrier: a wave is its own workgroup, occupancy is set by registers alone (the tiled kernel's 57 KB
// of samples per 256 lanes held it at two waves per SIMD), and the hot loop has no memory instruction that a lane
// waits for except the table's scalar loads (27 waves of a candidate share the table: scalar-cache hits).
// Per 8 steps and lane: 8 x 16 x ROWS packed multiplies/adds, 16 DPP moves, four uniform 16-byte loads.
// u = 5184 (symbol 161 at lag 32) has no lane: one extra wave per 64 candidates sums it, a candidate per lane; those
// waves are the first workgroups of the 1-D grid.
// Same operations per accumulator as demod_kernel => identical bits.
constexpr int kSysRows = 3;
constexpr int kSysU = 64 * kSysRows;                        // outputs per wave
constexpr int kSysOutputs = 32 * kNSymD;                    // u = 0 .. 5183 (+ u = 5184: the extra)
constexpr int kSysWaves = kSysOutputs / kSysU;              // 27
static_assert(kSysOutputs % kSysU == 0, "waves cover the outputs exactly");

// wsprd.c:199: a sample outside 0 < k < np is skipped; adding x*c with x = 0 leaves the sums as they are
__device__ __forceinline__ float sys_load_checked(const float* __restrict__ x, int k, int np) {
    const float v = x[min(max(k, 0), np - 1)];
    return (k > 0 && k < np) ? v : 0.0f;
}

// lane l takes src of lane l + 1; lane 63 keeps old (v_mov_b32_dpp wave_shl:1; tools/dpp_probe.hip).  The operands are
// float PARAMETERS on purpose: __builtin_bit_cast applied to a vector element (F.y) reads element 0 with this clang.
__device__ __forceinline__ float wave_shl1(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                 0x130, 0xf, 0xf, false));
}

// the whole wave lies inside the record: no bounds tests anywhere
__device__ __forceinline__ void lagsys_wave(const float* __restrict__ xi, const float* __restrict__ xq, int kw,
                                            const float4* __restrict__ gtab, ToneAcc (&acc)[kSysRows]) {
    constexpr int R = kSysRows, NS = R + 1;
    const int lane = threadIdx.x;
    // slot = one 8-sample vector: [0..3] pairs of I, [4..7] pairs of Q
    v2f S[NS][8];
    {
        const float* __restrict__ pi = xi + kw + 8 * R * lane;
        const float* __restrict__ pq = xq + kw + 8 * R * lane;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                S[r][i] = (v2f){pi[8 * r + 2 * i], pi[8 * r + 2 * i + 1]};
                S[r][4 + i] = (v2f){pq[8 * r + 2 * i], pq[8 * r + 2 * i + 1]};
            }
    }
    // lane 63's successor: the same address in every lane, but as VECTOR loads (the value must arrive in a VGPR, and
    // scalar loads would queue behind the table's): the zero the compiler cannot see through keeps them per-lane
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    const float* __restrict__ fi = xi + kw + 8 * R * 64 + zero;
    const float* __restrict__ fq = xq + kw + 8 * R * 64 + zero;
    auto fresh = [&](v2f (&F)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            F[i] = (v2f){fi[2 * i], fi[2 * i + 1]};
            F[4 + i] = (v2f){fq[2 * i], fq[2 * i + 1]};
        }
        fi += 8;
        fq += 8;
    };
    fresh(S[R]);
    struct Stage { float4 c[2], s[2]; };
    auto issue = [&](Stage& g, int j) {                      // j even; clamped past the end (values unused)
        const int jj = j < kSps ? j : 0;
#pragma unroll
        for (int u = 0; u < 2; ++u) { g.c[u] = gtab[2 * (jj + u)]; g.s[u] = gtab[2 * (jj + u) + 1]; }
    };
    Stage A, B;
    issue(A, 0);
#pragma unroll 1
    for (int b0 = 0; b0 < kSps / 8; b0 += NS) {
#pragma unroll
        for (int bb = 0; bb < NS; ++bb) {
            // rows of this block: slots (bb + r) % NS; the slot behind them receives the next vector of lane 63
#pragma unroll
            for (int q = 0; q < 4; ++q) {                    // a stage = steps 2q, 2q + 1 of the block
                Stage& cur = (q & 1) ? B : A;
                Stage& nxt = (q & 1) ? A : B;
                __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this stage's table entries have landed
                __builtin_amdgcn_sched_barrier(0);
                issue(nxt, 8 * (b0 + bb) + 2 * q + 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float4 c4 = cur.c[u], s4 = cur.s[u];
                    const v2f c01 = {c4.x, c4.y}, c23 = {c4.z, c4.w}, s01 = {s4.x, s4.y}, s23 = {s4.z, s4.w};
#pragma unroll
                    for (int r = 0; r < R; ++r) {           // ai = (ai + x*c) + y*s ; aq = (aq - x*s) + y*c (wsprd.c:200-207)
                        const v2f iv = S[(bb + r) % NS][q], qv = S[(bb + r) % NS][4 + q];
                        const float x = u ? iv.y : iv.x, y = u ? qv.y : qv.x;
                        const v2f xx = {x, x}, yy = {y, y};
                        const v2f p0 = xx * c01, p1 = xx * c23, p2 = xx * s01, p3 = xx * s23;
                        const v2f p4 = yy * s01, p5 = yy * s23, p6 = yy * c01, p7 = yy * c23;
                        acc[r].i01 = acc[r].i01 + p0; acc[r].i23 = acc[r].i23 + p1;
                        acc[r].q01 = acc[r].q01 - p2; acc[r].q23 = acc[r].q23 - p3;
                        acc[r].i01 = acc[r].i01 + p4; acc[r].i23 = acc[r].i23 + p5;
                        acc[r].q01 = acc[r].q01 + p6; acc[r].q23 = acc[r].q23 + p7;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // hand-over: the slot that held the first row now belongs to nobody; the last row of the next block is
            // the first row of the next lane (lane 63: the vector loaded a block ago)
            v2f (&F)[8] = S[(bb + R) % NS];
            const v2f (&G)[8] = S[bb % NS];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                F[i].x = wave_shl1(F[i].x, G[i].x);
                F[i].y = wave_shl1(F[i].y, G[i].y);
            }
            fresh(S[bb % NS]);                               // wanted one block from now
        }
    }
}

__device__ __forceinline__ void lagsys_store(const ToneAcc (&acc)[kSysRows], int u0, size_t item, float4* __restrict__ pw_out) {
    constexpr int nlag = 33;
#pragma unroll
    for (int r = 0; r < kSysRows; ++r) {
        const int u = u0 + kSysRows * (int)threadIdx.x + r, sym = u >> 5, m = u & 31;
        const float4 a = acc[r].amplitudes();
        pw_out[(item * nlag + m) * kNSymD + sym] = a;
        if (m == 0 && sym > 0) pw_out[(item * nlag + 32) * kNSymD + sym - 1] = a;     // (s - 1, lag 32) == (s, lag 0)
    }
}

// a wave that hangs over either end of the record (the first wave of an early candidate): every lane walks its
// outputs sample by sample with the reference's bounds test
__device__ __noinline__ void lagsys_edge_wave(const float* __restrict__ xi, const float* __restrict__ xq, int np, int kw,
                                              const float4* __restrict__ gtab, int u0, size_t item,
                                              float4* __restrict__ pw_out) {
    ToneAcc acc[kSysRows];
#pragma unroll
    for (int r = 0; r < kSysRows; ++r) acc[r].clear();
    const int kl = kw + 8 * kSysRows * (int)threadIdx.x;
#pragma unroll 1
    for (int j = 0; j < kSps; ++j) {
        const float4 c4 = gtab[2 * j], s4 = gtab[2 * j + 1];
#pragma unroll
        for (int r = 0; r < kSysRows; ++r) {
            const int k = kl + 8 * r + j;
            acc[r].step(make_float2(sys_load_checked(xi, k, np), sys_load_checked(xq, k, np)), c4, s4);
        }
    }
    lagsys_store(acc, u0, item, pw_out);
}

// 154 VGPRs, three waves per SIMD (a 128-register build spills 28 dwords into the hot loop and runs 12 % slower)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void demod_lagsys_kernel(const float* __restrict__ dI, const float* __restrict__ dQ, int np,
                         const FineState* __restrict__ items, const int* __restrict__ item_list, int nitems,
                         const float* __restrict__ tabs, float4* __restrict__ pw_out) {
    constexpr int nlag = 33;
    const int lane = threadIdx.x;
    // 1-D grid: first the waves that sum u = 5184 (a candidate per lane: 256 dependent steps of per-lane loads -- at the
    // front of the grid they run under the bulk; as the last workgroup of every 64th candidate, which a 2-D grid made
    // them, the final one trailed the launch by ~0.1 ms), then 27 waves per candidate
    const int nextra = (nitems + 63) >> 6;
    if ((int)blockIdx.x < nextra) {
        // u = 5184: (symbol 161, lag 32) of 64 candidates, one per lane
        const int pos = 64 * (int)blockIdx.x + lane;
        if (pos >= nitems) return;
        const int item = item_list[pos];
        const FineState st = items[item];
        const float* __restrict__ xi = dI + (size_t)st.seg * kIqStride;
        const float* __restrict__ xq = dQ + (size_t)st.seg * kIqStride;
        const float4* __restrict__ tb = reinterpret_cast<const float4*>(tabs) + (size_t)st.pad * 512;
        const int k0 = st.shift_coarse - 128 + 8 * kSysOutputs;
        ToneAcc a;
        a.clear();
#pragma unroll 2
        for (int j = 0; j < kSps; ++j)
            a.step(make_float2(sys_load_checked(xi, k0 + j, np), sys_load_checked(xq, k0 + j, np)), tb[2 * j], tb[2 * j + 1]);
        pw_out[((size_t)item * nlag + 32) * kNSymD + kNSymD - 1] = a.amplitudes();
        return;
    }
    // XCD-aware placement: consecutive workgroups go to consecutive XCDs (8 of them, each with its own L2), so
    // workgroup w serves candidate 8 (w / 216) + w % 8, wave (w / 8) % 27: the 27 waves of a candidate -- which share its
    // 8 KB table and read adjacent, overlapping blocks of its samples -- all run behind ONE L2 (otherwise every XCD
    // fetches every table); alone the kernel's time does not change, in the pipeline the step gains about 1 %
    const int w = (int)blockIdx.x - nextra;
    const int pos = 8 * (w / (8 * kSysWaves)) + (w & 7);
    if (pos >= nitems) return;
    const int item = item_list[pos];
    const FineState st = items[item];
    const float* __restrict__ xi = dI + (size_t)st.seg * kIqStride;
    const float* __restrict__ xq = dQ + (size_t)st.seg * kIqStride;
    const int u0 = ((w >> 3) % kSysWaves) * kSysU;
    const int kw = __builtin_amdgcn_readfirstlane(st.shift_coarse - 128 + 8 * u0);
    const float4* __restrict__ gtab = reinterpret_cast<const float4*>(tabs) +
                                      (size_t)__builtin_amdgcn_readfirstlane(st.pad) * 512;
    // samples the wave touches: kw .. kw + 8 * 192 + 8 * 31 + 7 (one more vector is fetched and never used)
    if (kw > 0 && kw + 8 * kSysU + kSps + 8 <= np) {
        ToneAcc acc[kSysRows];
#pragma unroll
        for (int r = 0; r < kSysRows; ++r) acc[r].clear();
        lagsys_wave(xi, xq, kw, gtab, acc);
        lagsys_store(acc, u0, (size_t)item, pw_out);
    } else {
        lagsys_edge_wave(xi, xq, np, kw, gtab, u0, (size_t)item, pw_out);
    }
}

// folds the 162 per-symbol tone amplitudes of one (candidate, lag) in symbol order
__global__ __launch_bounds__(64)
void demod_metric_kernel(const float4* __restrict__ pw, const FineState* __restrict__ items, int nitems,
                         int mode, int nlag, float minsync1, float* __restrict__ sync_out,
                         unsigned char* __restrict__ sym_out, float* __restrict__ rms_out,
                         const unsigned char* __restrict__ pr3) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nitems * nlag) return;
    const int item = idx / nlag;
    if (mode == 2 && !(items[item].sync > minsync1)) return;
    const float4* __restrict__ P = pw + (size_t)idx * kNSymD;
    float ss = 0.0f, totp = 0.0f;
    for (int k = 0; k < kNSymD; ++k) {
        const float4 p = P[k];
        totp = totp + p.x + p.y + p.z + p.w;
        const float cmet = (p.y + p.w) - (p.x + p.z);
        ss = pr3[k] ? ss + cmet : ss - cmet;
    }
    ss = ss / totp;
    if (mode != 2) { sync_out[idx] = ss; return; }
    sync_out[idx] = (ss > -1e30f) ? ss : -1e30f;
    float fsum = 0.0f, f2sum = 0.0f;
    for (int k = 0; k < kNSymD; ++k) {
        const float4 p = P[k];
        const float f = pr3[k] ? p.w - p.y : p.z - p.x;
        fsum += f / 162.0f;
        const float ff = f * f;
        f2sum += ff / 162.0f;
    }
    const float m2 = fsum * fsum;
    const float fac = sqrtf(f2sum - m2);
    float sq = 0.0f;
    unsigned char* __restrict__ so = sym_out + (size_t)idx * kNSymD;
    for (int k = 0; k < kNSymD; ++k) {
        const float4 p = P[k];
        const float f = pr3[k] ? p.w - p.y : p.z - p.x;
        float v = 50.0f * f / fac;
        if (v > 127.0f) v = 127.0f;
        if (v < -128.0f) v = -128.0f;
        const float w = v + 128.0f;
        const unsigned char b = (w == w) ? (unsigned char)(int)w : (unsigned char)0;
        so[k] = b;
        const float y = (float)b - 128.0f;
        sq += y * y;
    }
    rms_out[idx] = sqrtf(sq / 162.0f);
}

// -----------------------------------------------------------------------------
// Frequency scan (mode 1) + first ladder rung (mode 2 at jitter 0) for candidates
// without drift.  The five frequency hypotheses share the samples (same lag) and
// each has ONE phasor table; the winning hypothesis' tone amplitudes are exactly
// what mode 2 recomputes at (best freq, best lag), so the first soft-symbol vector
// comes out of the same pass.  lane = (symbol, frequency).
constexpr int kNFreq = 5;

// The centre hypothesis of the frequency scan (ifreq = 0) is the lag scan's winner over again: mode 0 left
// *freq at freq_coarse and set *shift to the best lag of its grid (wsprd.c:709-719), so mode 1 at f0 = *freq + 0
// and lag = *shift (:721-726) repeats, operation for operation, the sums whose tone amplitudes the lag scan
// already wrote for that lag.  Returns the lag's index in the lag-scan block, or -1 when the state is not a grid
// point of that scan (no lag won: NaN metrics) or no block is given.
__device__ __forceinline__ int centre_from_lag_scan(const FineState& st, int nlag, int lagstep) {
    if (nlag <= 0 || !(st.freq == st.freq_coarse)) return -1;
    const int d = st.shift - (st.shift_coarse - 128);
    if (d < 0 || d % lagstep != 0) return -1;
    const int m = d / lagstep;
    return m < nlag ? m : -1;
}

__global__ __launch_bounds__(64)
void phasor_freq_kernel(const FineState* __restrict__ items, const int* __restrict__ item_list, int ifmin,
                        float fstep, float* __restrict__ tabs) {
    const int slot = blockIdx.x, lane = threadIdx.x;
    if (lane >= 4 * kNFreq) return;
    const FineState st = items[item_list[slot]];
    const int f = lane >> 2, tone = lane & 3;
    const float f0 = st.freq + (float)(ifmin + f) * fstep;
    const float fp = (float)((double)f0 + ((double)st.drift / 2.0) * (double)(0.0f - 81.0f) / (double)81.0f);
    const double off = (tone == 0) ? -kDf15 : (tone == 1) ? -kDf05 : (tone == 2) ? kDf05 : kDf15;
    const float dphi = (float)(kTwoPiDt * ((double)fp + off));
    const float cd = glibc_cosf(dphi), sd = glibc_sinf(dphi);
    float* __restrict__ t = tabs + ((size_t)slot * kNFreq + f) * 2048;
    float c = 1.0f, s = 0.0f;
    for (int j = 0; j < kSps; ++j) {
        if (j > 0) {
            const float a = c * cd, b = s * sd, e = c * sd, d = s * cd;
            c = a - b;
            s = e + d;
        }
        t[8 * j + tone] = c;
        t[8 * j + 4 + tone] = s;
    }
}

constexpr int kFsThreads = 832;                                   // 13 waves: 5 x 162 = 810 working threads (freq_drift_kernel)
constexpr int kFsChunk = 32;
constexpr int kFsPerThread = (kNSymD * kFsChunk + kFsThreads - 1) / kFsThreads;      // 7 samples staged per thread and chunk

// Scalar-table frequency scan: a hypothesis owns three whole waves (lane = symbol, 162 of 192 lanes), so its one
// table is wave-uniform and is read through the scalar cache straight into SGPR operands of the packed multiplies,
// as in the lag scan -- no 40 KB of tables in LDS and no two 16-byte LDS reads per lane and step.  The centre
// hypothesis is not summed: the lag scan has formed exactly these sums for the lag that won (centre_from_lag_scan),
// so twelve waves sum the other four and the first three copy the centre's amplitudes (round 4; until then the
// centre had three waves of its own that only helped staging).  A candidate whose state is not a grid point of the
// lag scan (no lag won) gets its centre from freq_centre_rare_kernel behind this one.
constexpr int kFqThreads = 768;                                   // 4 hypotheses x 3 waves
constexpr int kFqPerThread = (kNSymD * kFsChunk + kFqThreads - 1) / kFqThreads;      // 7 samples staged per thread and chunk

__global__ __launch_bounds__(kFqThreads) __attribute__((amdgpu_waves_per_eu(6, 8), amdgpu_num_vgpr(64)))
void freq_scalar_kernel(const float* __restrict__ dI, const float* __restrict__ dQ, int np,
                        const FineState* __restrict__ items, const int* __restrict__ item_list,
                        const float* __restrict__ tabs, float4* __restrict__ pw_out,
                        const float4* __restrict__ pw_lag, int nlag, int lagstep) {
    __shared__ float2 tile[kNSymD][kFsChunk + 1];
    const int slot = blockIdx.x, tid = threadIdx.x;
    const int item = item_list[slot];
    const FineState st = items[item];
    const float* __restrict__ xi = dI + (size_t)st.seg * kIqStride;
    const float* __restrict__ xq = dQ + (size_t)st.seg * kIqStride;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave / 3, sym = (wave - 3 * h) * 64 + (tid & 63);
    const int f = h < kNFreq / 2 ? h : h + 1;
    const int m_centre = centre_from_lag_scan(st, nlag, lagstep);          // block-uniform
    const bool working = sym < kNSymD;
    const float4* __restrict__ gt = reinterpret_cast<const float4*>(tabs) + ((size_t)slot * kNFreq + f) * (2 * kSps);

    float2 nxt[kFqPerThread];
    auto fetch = [&](int c) {
#pragma unroll
        for (int u = 0; u < kFqPerThread; ++u) {
            const int e = u * kFqThreads + tid, row = e >> 5, col = e & (kFsChunk - 1);
            const int k = st.shift + kSps * row + kFsChunk * c + col;
            const bool ok = (e < kNSymD * kFsChunk) && (k > 0) && (k < np);
            nxt[u] = ok ? make_float2(xi[k], xq[k]) : make_float2(0.0f, 0.0f);
        }
    };
    fetch(0);
    ToneAcc acc;
    acc.clear();
    for (int c = 0; c < kSps / kFsChunk; ++c) {
        __syncthreads();                                             // the previous chunk has been consumed
#pragma unroll
        for (int u = 0; u < kFqPerThread; ++u) {
            const int e = u * kFqThreads + tid;
            if (e < kNSymD * kFsChunk) tile[e >> 5][e & (kFsChunk - 1)] = nxt[u];
        }
        __syncthreads();
        if (c + 1 < kSps / kFsChunk) fetch(c + 1);
        if (working) {
#pragma unroll 8
            for (int jj = 0; jj < kFsChunk; ++jj) {
                const int j = kFsChunk * c + jj;
                acc.step(tile[sym][jj], gt[2 * j], gt[2 * j + 1]);
            }
        }
    }
    if (working) pw_out[((size_t)slot * kNFreq + f) * kNSymD + sym] = acc.amplitudes();
    if (m_centre >= 0 && h == 0 && working)
        pw_out[((size_t)slot * kNFreq + kNFreq / 2) * kNSymD + sym] = pw_lag[((size_t)item * nlag + m_centre) * kNSymD + sym];
}

// Round 6 kept the table in REGISTERS instead: a lane fetched the (cos, sin) of one tone for a pair of steps with vector
// loads four pairs ahead and V_MFMA_F32_4X4X1_16B_F32 (B = 1.0, C = 0: D[i] = A(lane 4b + i) x 1 + 0, exact) spread them
// over the wave -- operands in VGPRs, nothing through the scalar cache.  Bit-identical (trace parity) and NOT faster:
// 375 us against 367 us per 2 048 candidates for this kernel, because the table is ~45 us of it; the bare arithmetic (sixteen
// packed instructions and one tile read per step) takes 242 us, 0.70 of its issue bound like the lag scan and the
// subtraction, staging and barriers ~60 us (profiles/r06_freq_scan_table_in_registers_ab.txt).  The kernel was deleted again;
// the experiment is in the history (commit "freq_bcast_kernel: ablation switches ...").

// Round 5 tried these sums with NO LDS and NO barrier (verdict of round 4: VALU active 11 % of the wave cycles, 69 %
// waiting): a wave on its own, lane = symbol, the lane's 8 or 16 samples of a chunk in registers with the next chunk's
// loads in flight, the four hypotheses walked one after the other over those registers, tables through the scalar
// cache a two- or four-step stage ahead; as three single-wave workgroups per candidate (XCD-aware) and as one workgroup
// of three waves.  Bit-identical outputs (trace parity), 9 % fewer vector instructions -- and SLOWER: 0.51-0.58 ms per
// 2 048 candidates against 0.43-0.44 for the kernel above on the same boxes, with the same counter picture (VALU active
// 0.12-0.14, waiting 0.70-0.72: profiles/r05_freq_scan_barrier_free_ab.txt).  What the waves wait for is therefore not
// the barriers but the TABLE: 32 KB per candidate streamed once through a 16 KB scalar cache, every 64-byte line a trip
// to L2, and scalar loads return out of order, so a wave can be one stage ahead and no more (s_waitcnt lgkmcnt(0));
// the twelve-wave form hides that latency better (24 waves per CU, three waves share each line).  The kernel was
// deleted again; the experiment is in the history (commit "Barrier-free frequency scan kernel").

// The centre hypothesis of the candidates freq_scalar_kernel could not copy it for (rare: no lag won; or nlag = 0,
// the WSPR_K4_FREQ=nocentre switch of the trace tests): one wave per candidate checks, and sums the 162 symbols
// itself, three rounds of 64, samples straight from memory -- the same operations in the same order.
__global__ __launch_bounds__(64)
void freq_centre_rare_kernel(const float* __restrict__ dI, const float* __restrict__ dQ, int np,
                             const FineState* __restrict__ items, const int* __restrict__ item_list,
                             const float* __restrict__ tabs, float4* __restrict__ pw_out, int nlag, int lagstep) {
    const int slot = blockIdx.x;
    const FineState st = items[item_list[slot]];
    if (centre_from_lag_scan(st, nlag, lagstep) >= 0) return;
    const float* __restrict__ xi = dI + (size_t)st.seg * kIqStride;
    const float* __restrict__ xq = dQ + (size_t)st.seg * kIqStride;
    const float4* __restrict__ gt = reinterpret_cast<const float4*>(tabs) + ((size_t)slot * kNFreq + kNFreq / 2) * (2 * kSps);
    for (int sym = threadIdx.x; sym < kNSymD; sym += 64) {
        ToneAcc acc;
        acc.clear();
        for (int j = 0; j < kSps; ++j) {
            const int k = st.shift + kSps * sym + j;
            const bool ok = (k > 0) && (k < np);
            acc.step(ok ? make_float2(xi[k], xq[k]) : make_float2(0.0f, 0.0f), gt[2 * j], gt[2 * j + 1]);
        }
        pw_out[((size_t)slot * kNFreq + kNFreq / 2) * kNSymD + sym] = acc.amplitudes();
    }
}

// The same scan for a DRIFTING candidate: every (hypothesis, symbol) has its own four tone phasors and each
// of them is used by exactly one lane, so there is nothing to share and no table: a lane carries the four
// recurrences (wsprd.c:158-188) in registers next to its accumulators, 12 + 16 packed instructions per sample.
// Staging, thread mapping and output layout are those of freq_tile_kernel, so freq_metric_kernel picks the
// winner and forms the first rung's soft symbols for drifting candidates too (the general kernel ran the five
// hypotheses as five workgroups and the first rung as a sixth pass over the samples).
__global__ __launch_bounds__(kFsThreads) __attribute__((amdgpu_waves_per_eu(7, 8)))     // <= 72 VGPRs: two workgroups per CU
void freq_drift_kernel(const float* __restrict__ dI, const float* __restrict__ dQ, int np,
                       const FineState* __restrict__ items, const int* __restrict__ item_list, int ifmin, float fstep,
                       float4* __restrict__ pw_out, const float4* __restrict__ pw_lag, int nlag, int lagstep) {
    __shared__ float2 tile[kNSymD][kFsChunk + 1];
    const int slot = blockIdx.x, tid = threadIdx.x;
    const int item = item_list[slot];
    const FineState st = items[item];
    const float* __restrict__ xi = dI + (size_t)st.seg * kIqStride;
    const float* __restrict__ xq = dQ + (size_t)st.seg * kIqStride;
    const int f = tid / kNSymD, sym = tid - f * kNSymD;
    // (a thread of the centre hypothesis copies the lag scan's amplitudes when they are its own: see centre_from_lag_scan)
    const int m_centre = (f == kNFreq / 2) ? centre_from_lag_scan(st, pw_lag ? nlag : 0, lagstep) : -1;
    const bool working = f < kNFreq && m_centre < 0;

    float2 nxt[kFsPerThread];
    auto fetch = [&](int c) {
#pragma unroll
        for (int u = 0; u < kFsPerThread; ++u) {
            const int e = u * kFsThreads + tid, row = e >> 5, col = e & (kFsChunk - 1);
            const int k = st.shift + kSps * row + kFsChunk * c + col;
            const bool ok = (e < kNSymD * kFsChunk) && (k > 0) && (k < np);
            nxt[u] = ok ? make_float2(xi[k], xq[k]) : make_float2(0.0f, 0.0f);
        }
    };
    fetch(0);
    v2f cd01 = {1.0f, 1.0f}, cd23 = cd01, sd01 = {0.0f, 0.0f}, sd23 = sd01;
    if (working) {
        const float f0 = st.freq + (float)(ifmin + f) * fstep;          // *freq + ifreq * fstep, wsprd.c:151
        const float fp = (float)((double)f0 + ((double)st.drift / 2.0) * (double)((float)sym - 81.0f) / (double)81.0f);
        const double fpd = (double)fp;
        float sn, cs;
        glibc_sincosf_pair((float)(kTwoPiDt * (fpd - kDf15)), &sn, &cs); cd01.x = cs; sd01.x = sn;
        glibc_sincosf_pair((float)(kTwoPiDt * (fpd - kDf05)), &sn, &cs); cd01.y = cs; sd01.y = sn;
        glibc_sincosf_pair((float)(kTwoPiDt * (fpd + kDf05)), &sn, &cs); cd23.x = cs; sd23.x = sn;
        glibc_sincosf_pair((float)(kTwoPiDt * (fpd + kDf15)), &sn, &cs); cd23.y = cs; sd23.y = sn;
    }
    v2f c01 = {1.0f, 1.0f}, c23 = c01, s01 = {0.0f, 0.0f}, s23 = s01;
    ToneAcc acc;
    acc.clear();
    for (int c = 0; c < kSps / kFsChunk; ++c) {
        __syncthreads();                                             // the previous chunk has been consumed
#pragma unroll
        for (int u = 0; u < kFsPerThread; ++u) {
            const int e = u * kFsThreads + tid;
            if (e < kNSymD * kFsChunk) tile[e >> 5][e & (kFsChunk - 1)] = nxt[u];
        }
        __syncthreads();
        if (c + 1 < kSps / kFsChunk) fetch(c + 1);
        if (working) {
#pragma unroll 8
            for (int jj = 0; jj < kFsChunk; ++jj) {
                if (c + jj > 0) {
                    const v2f a01 = c01 * cd01, b01 = s01 * sd01, e01 = c01 * sd01, d01 = s01 * cd01;
                    const v2f a23 = c23 * cd23, b23 = s23 * sd23, e23 = c23 * sd23, d23 = s23 * cd23;
                    c01 = a01 - b01; s01 = e01 + d01;
                    c23 = a23 - b23; s23 = e23 + d23;
                }
                acc.step(tile[sym][jj], make_float4(c01.x, c01.y, c23.x, c23.y), make_float4(s01.x, s01.y, s23.x, s23.y));
            }
        }
    }
    if (working) pw_out[((size_t)slot * kNFreq + f) * kNSymD + sym] = acc.amplitudes();
    else if (f < kNFreq) pw_out[((size_t)slot * kNFreq + f) * kNSymD + sym] = pw_lag[((size_t)item * nlag + m_centre) * kNSymD + sym];
}

// one wave per candidate: lanes 0..4 fold one frequency hypothesis each (162 symbols in
// order), lane 0 applies the strict '>' pick of wsprd.c:227-232 and updates the state, then --
// if worth a try -- forms the jitter-0 soft symbols from the winner's amplitudes
__global__ __launch_bounds__(64)
void freq_metric_kernel(const float4* __restrict__ pw, FineState* __restrict__ items,
                        const int* __restrict__ item_list, int nshared, int ifmin, float fstep,
                        float minsync1, float* __restrict__ sync_out, unsigned char* __restrict__ sym_out,
                        float* __restrict__ rms_out, const unsigned char* __restrict__ pr3) {
    __shared__ float4 P[kNFreq * kNSymD];
    __shared__ float met[kNFreq];
    __shared__ float fsym[kNSymD];
    __shared__ float fac_s;
    __shared__ int best_s;
    const int slot = blockIdx.x, lane = threadIdx.x;
    const int item = item_list[slot];
    for (int e = lane; e < kNFreq * kNSymD; e += 64) P[e] = pw[(size_t)slot * kNFreq * kNSymD + e];
    __syncthreads();
    if (lane < kNFreq) {
        float ss = 0.0f, totp = 0.0f;
        for (int k = 0; k < kNSymD; ++k) {
            const float4 p = P[lane * kNSymD + k];
            totp = totp + p.x + p.y + p.z + p.w;
            const float cmet = (p.y + p.w) - (p.x + p.z);
            ss = pr3[k] ? ss + cmet : ss - cmet;
        }
        met[lane] = ss / totp;
    }
    __syncthreads();
    if (lane == 0) {
        FineState st = items[item];
        const float fin = st.freq;
        float best = -1e30f, fbest = 0.0f;
        int bshift = 0, bf = -1;
        for (int f = 0; f < kNFreq; ++f)
            if (met[f] > best) { best = met[f]; fbest = fin + (float)(ifmin + f) * fstep; bshift = st.shift; bf = f; }
        st.freq = fbest;
        st.shift = bshift;
        st.sync = best;
        items[item] = st;
        best_s = (best > minsync1) ? bf : -1;
        if (best_s >= 0) sync_out[item] = best;
    }
    __syncthreads();
    const int bf = best_s;
    if (bf < 0) return;
    // mode 2 at (fbest, shift): same accumulators as hypothesis bf (wsprd.c:219-225, 243-256)
    for (int k = lane; k < kNSymD; k += 64) {
        const float4 p = P[bf * kNSymD + k];
        fsym[k] = pr3[k] ? p.w - p.y : p.z - p.x;
    }
    __syncthreads();
    if (lane == 0) {
        float fsum = 0.0f, f2sum = 0.0f;
        for (int k = 0; k < kNSymD; ++k) {
            const float f = fsym[k];
            fsum += f / 162.0f;
            const float ff = f * f;
            f2sum += ff / 162.0f;
        }
        const float m2 = fsum * fsum;
        fac_s = sqrtf(f2sum - m2);
    }
    __syncthreads();
    const float fac = fac_s;
    float sq = 0.0f;                 // sum of squares of small integers: exact in any order
    for (int k = lane; k < kNSymD; k += 64) {
        float v = 50.0f * fsym[k] / fac;
        if (v > 127.0f) v = 127.0f;
        if (v < -128.0f) v = -128.0f;
        const float w = v + 128.0f;
        const unsigned char b = (w == w) ? (unsigned char)(int)w : (unsigned char)0;
        sym_out[(size_t)item * kNSymD + k] = b;
        const float y = (float)b - 128.0f;
        sq += y * y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) rms_out[item] = sqrtf(sq / 162.0f);
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
__global__ void pick_freq_kernel(FineState* __restrict__ items, const int* __restrict__ item_list, int nitems,
                                 const float* __restrict__ sync_in, int nfreq, int ifmin, float fstep) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= nitems) return;
    const int it = item_list ? item_list[pos] : pos;
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
                  const DeviceTables& t, hipStream_t st, int symfac) {
    if (nitems <= 0 || nlag <= 0) return;
    hipLaunchKernelGGL(demod_kernel, dim3(nlag, nitems), dim3(192), 0, st, dI, dQ, samples, items,
                       (const int*)nullptr, mode, nlag, lagstep, ifmin, fstep, jitter, minsync1, sync_out,
                       sym_out, rms_out, t.sync, (float)symfac);
}
// Frequency scan (5 hypotheses at +-0.2 Hz, step 0.1) followed by the first ladder rung.
// Drift-free candidates (list_shared) take the fused tiled path; drifting ones (list_own) the
// general kernel.  Outputs: items[] updated (freq, sync), rung-0 sync/sym/rms at [item].
void launch_freq_scan_and_first_rung(const float* dI, const float* dQ, int samples, FineState* items,
                                     const int* list_shared, int n_shared, const int* list_own, int n_own,
                                     int lagstep, float minsync1, const int* jitter0, float* tabs, float* pw,
                                     float* scratch_sync, float* sync_out, unsigned char* sym_out,
                                     float* rms_out, const DeviceTables& t, hipStream_t st,
                                     const float* pw_lag, int nlag_lag) {
    // pw_lag (optional): the lag scan's amplitude block [item][nlag_lag][162] of the SAME items, still intact --
    // the centre hypothesis is read from it instead of being summed again; pw must then be a different buffer
    const float4* pl = reinterpret_cast<const float4*>(pw_lag);
    // WSPR_REPEAT_FREQ / _LAG / _FANO = 2: the stage's kernels are launched twice (same outputs) -- what a stage costs
    // INSIDE the pipelined step is the difference of two bench lines (docs/HISTORY.md section 4)
    static const int rep_freq = [] { const char* e = lab_env("WSPR_REPEAT_FREQ"); return e ? std::max(1, atoi(e)) : 1; }();
    // WSPR_K4_FREQ=nocentre: no centre hypothesis is taken from the lag scan (every candidate through the rare path)
    static const bool nocentre = [] { const char* e = lab_env("WSPR_K4_FREQ"); return e && e[0] == 'n'; }();
    const int nlag_c = (pl && !nocentre) ? nlag_lag : 0;
    if (n_shared > 0) {
        hipLaunchKernelGGL(phasor_freq_kernel, dim3(n_shared), dim3(64), 0, st, items, list_shared, -2, 0.1f, tabs);
        for (int r = 0; r < rep_freq; ++r)
            hipLaunchKernelGGL(freq_scalar_kernel, dim3(n_shared), dim3(kFqThreads), 0, st, dI, dQ, samples, items,
                               list_shared, tabs, reinterpret_cast<float4*>(pw), pl, nlag_c, lagstep);
        hipLaunchKernelGGL(freq_centre_rare_kernel, dim3(n_shared), dim3(64), 0, st, dI, dQ, samples, items, list_shared,
                           tabs, reinterpret_cast<float4*>(pw), nlag_c, lagstep);
        hipLaunchKernelGGL(freq_metric_kernel, dim3(n_shared), dim3(64), 0, st,
                           reinterpret_cast<const float4*>(pw), items, list_shared, n_shared, -2, 0.1f, minsync1,
                           sync_out, sym_out, rms_out, t.sync);
    }
    if (n_own > 0) {
        static const bool general = [] { const char* e = lab_env("WSPR_K4_DRIFT"); return e && e[0] == 't'; }();
        if (!general) {
            // pw rows of the drifting candidates follow those of the drift-free ones
            float4* pw_own = reinterpret_cast<float4*>(pw) + (size_t)n_shared * kNFreq * kNSymD;
            for (int r = 0; r < rep_freq; ++r)
            hipLaunchKernelGGL(freq_drift_kernel, dim3(n_own), dim3(kFsThreads), 0, st, dI, dQ, samples, items, list_own,
                               -2, 0.1f, pw_own, nocentre ? nullptr : pl, nlag_lag, lagstep);
            hipLaunchKernelGGL(freq_metric_kernel, dim3(n_own), dim3(64), 0, st, pw_own, items, list_own, n_own, -2, 0.1f,
                               minsync1, sync_out, sym_out, rms_out, t.sync);
            return;
        }
        // scratch_sync is indexed [item][5] by the general kernel
        hipLaunchKernelGGL(demod_kernel, dim3(kNFreq, n_own), dim3(192), 0, st, dI, dQ, samples, items, list_own, 1,
                           kNFreq, lagstep, -2, 0.1f, (const int*)nullptr, 0.0f, scratch_sync, (unsigned char*)nullptr,
                           (float*)nullptr, t.sync, 50.0f);
        hipLaunchKernelGGL(pick_freq_kernel, dim3((n_own + 63) / 64), dim3(64), 0, st, items, list_own, n_own,
                           scratch_sync, kNFreq, -2, 0.1f);
        hipLaunchKernelGGL(demod_kernel, dim3(1, n_own), dim3(192), 0, st, dI, dQ, samples, items, list_own, 2, 1,
                           lagstep, 0, 0.0f, jitter0, minsync1, sync_out, sym_out, rms_out, t.sync, 50.0f);
    }
}

void launch_phasor_tables(const FineState* items, int nitems, int mode, float* tabs, hipStream_t st) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(phasor_table_kernel, dim3(1, nitems), dim3(64), 0, st, items, mode, tabs);
}

// item_list_shared / item_list_own: indices into items[] of the candidates without / with drift
void launch_demod_tiled(const float* dI, const float* dQ, int samples, const FineState* items, int nitems,
                        const int* list_shared, int n_shared, const int* list_own, int n_own, int mode,
                        int nlag, int lagstep, float minsync1, const float* tabs, float* pw,
                        float* sync_out, unsigned char* sym_out, float* rms_out,
                        const DeviceTables& t, hipStream_t st) {
    if (nitems <= 0) return;
    auto tile_bytes = [&](int syms) {
        const int span = kSps * syms + lagstep * (nlag - 1);
        const int pitch = (span + lagstep - 1) / lagstep + 1;
        return (size_t)pitch * lagstep * sizeof(float2);
    };
    auto threads = [&](int syms) { return dim3(((syms * nlag + 63) / 64) * 64); };
    float4* pw4 = reinterpret_cast<float4*>(pw);
    // WSPR_K4_LAG=tile: drift-free candidates' full lag scan on demod_tile_kernel<8, true> (one (symbol, lag) per lane,
    // samples in LDS) instead of the register-resident correlation
    static const bool lagsys_kernel = [] { const char* e = lab_env("WSPR_K4_LAG"); return !(e && e[0] == 't'); }();
    // WSPR_K4_DRIFT=tile: drifting candidates' full lag scan on demod_tile_kernel<8, false> (one lag per lane)
    static const bool drift_kernel = [] { const char* e = lab_env("WSPR_K4_DRIFT"); return !(e && e[0] == 't'); }();
    static const int rep_lag = [] { const char* e = lab_env("WSPR_REPEAT_LAG"); return e ? std::max(1, atoi(e)) : 1; }();
#define WSPR_LAUNCH_TILE(STEP)                                                                                   \
    do {                                                                                                         \
        if (n_shared > 0 && STEP == 8 && nlag == 33 && mode == 0 && lagsys_kernel)                               \
            for (int r_ = 0; r_ < rep_lag; ++r_)                                                                 \
            hipLaunchKernelGGL(demod_lagsys_kernel, dim3(kSysWaves * ((n_shared + 7) & ~7) + (n_shared + 63) / 64), dim3(64), 0, st, \
                               dI, dQ, samples, items, list_shared, n_shared, tabs, pw4);                        \
        else if (n_shared > 0)                                                                                   \
            hipLaunchKernelGGL((demod_tile_kernel<STEP, true>), dim3(kNSymD / kTileSymsShared, n_shared),        \
                               threads(kTileSymsShared), tile_bytes(kTileSymsShared), st, dI, dQ, samples,        \
                               items, list_shared, mode, nlag, minsync1, tabs, pw4);                             \
        if (n_own > 0 && STEP == 8 && nlag == 33 && mode == 0 && drift_kernel)                                   \
            for (int r_ = 0; r_ < rep_lag; ++r_)                                                                 \
            hipLaunchKernelGGL(demod_drift_kernel, dim3((kNSymD + kDrSyms - 1) / kDrSyms, n_own), dim3(64), 0, st, \
                               dI, dQ, samples, items, list_own, pw4);                                           \
        else if (n_own > 0)                                                                                      \
            hipLaunchKernelGGL((demod_tile_kernel<STEP, false>), dim3(kNSymD / kTileSymsOwn, n_own),             \
                               threads(kTileSymsOwn), kTileSymsOwn * kOwnTabPitch * 16 + tile_bytes(kTileSymsOwn), st, dI, dQ, \
                               samples, items, list_own, mode, nlag, minsync1, tabs, pw4);                       \
    } while (0)
    if (lagstep == 8) WSPR_LAUNCH_TILE(8);
    else if (lagstep == 16) WSPR_LAUNCH_TILE(16);
    else WSPR_LAUNCH_TILE(3);
#undef WSPR_LAUNCH_TILE
    hipLaunchKernelGGL(demod_metric_kernel, dim3((nitems * nlag + 63) / 64), dim3(64), 0, st, pw4, items, nitems,
                       mode, nlag, minsync1, sync_out, sym_out, rms_out, t.sync);
}

#ifdef WSPR_LAB   // calibration kernel: lab build only
// Calibration for the vector rooflines: register-only chains of separately rounded packed multiplies and adds
// (the instruction mix of the matched-filter sums, nothing else), enough waves to fill every SIMD.  What it
// sustains is the practical ceiling of v_pk_mul_f32 / v_pk_add_f32 at the clock the GPU holds under this load.
namespace {
__global__ __launch_bounds__(256)
void calib_valu_kernel(float* __restrict__ out, int iters) {
    v2f a[8], m = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (v2f){1.0f + threadIdx.x * 1e-6f + k, 1.0f - k * 1e-3f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const v2f p = a[k] * m;            // v_pk_mul_f32
                a[k] = p + c;                      // v_pk_add_f32 (contraction is off in this file)
            }
        }
    }
    v2f s = a[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) s = s + a[k];
    if (s.x == 123.456f) out[0] = s.y;             // keeps the chains alive; practically never true
}
}  // namespace
// flops (one per multiply or add and lane half) of one launch
double launch_calib_valu(float* out, int iters, hipStream_t st) {
    const int wgs = 256 * 8;                       // 8 workgroups of 4 waves per CU
    hipLaunchKernelGGL(calib_valu_kernel, dim3(wgs), dim3(256), 0, st, out, iters);
    return (double)wgs * 256 * iters * 8 * 8 * 4;  // 8 x 8 (mul + add) pairs of 2 lanes-halves
}
#endif  // WSPR_LAB

void launch_pick_lag(FineState* items, int nitems, const float* sync_in, int nlag, int lagstep, hipStream_t st) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(pick_lag_kernel, dim3((nitems + 63) / 64), dim3(64), 0, st, items, nitems, sync_in, nlag, lagstep);
}
void launch_pick_freq(FineState* items, int nitems, const float* sync_in, int nfreq, int ifmin,
                      float fstep, hipStream_t st) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(pick_freq_kernel, dim3((nitems + 63) / 64), dim3(64), 0, st, items, (const int*)nullptr,
                       nitems, sync_in, nfreq, ifmin, fstep);
}

}  // namespace wspr

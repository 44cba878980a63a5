// K1 -- windowed 512-point FFT bank + power spectrogram (gfx950 / wave64).
//
// Replaces reference wsprd/wsprd.c:509-553: 347 sine-windowed FFTs per segment,
// hop 128, squared magnitudes, fft-shifted.  Only the 417 bins any consumer reads
// (fft-shifted 48..464) are written.
//
// Mapping: one wavefront owns one FFT at a time, 8 complex points per lane.
//   * radix-2 decimation-in-frequency, 9 stages = 3 register passes of 3 stages;
//     between passes the 512 points are transposed through a wave-private,
//     bank-conflict-free LDS tile (row stride 72 / pad 9 complex words).
//   * consecutive FFTs overlap by 384 samples: a wave walks a run of consecutive
//     blocks and keeps the raw samples in a sliding register window, so every IQ
//     sample is fetched from HBM/L2 once per wave (2 coalesced 256-B rows of I and
//     of Q per FFT) and no LDS staging of the input is needed.
//   * window and twiddles live in registers for the whole run.
// Roofline: HBM-bound (algorithmic 938 796 B / segment / pass for ~9 MFLOP).
//
// The butterfly arithmetic (u+v, (u-v)*w with separately rounded products) is the
// same as the CPU oracle's orc_fft512(), so results are bit-identical to it.
#include "wspr_device.h"
#include <cstdlib>

#pragma clang fp contract(off)

namespace wspr {

namespace {
constexpr int kWavesPerWg    = 4;
constexpr int kTile          = 576;     // complex words of LDS per wave

__device__ __forceinline__ void bfly(float2& u, float2& v, const float2 w) {
    const float dr = u.x - v.x, di = u.y - v.y;
    u.x = u.x + v.x;
    u.y = u.y + v.y;
    const float t1 = dr * w.x, t2 = di * w.y, t3 = dr * w.y, t4 = di * w.x;
    v.x = t1 - t2;
    v.y = t3 + t4;
}

// three DIF stages on the 8 register-resident points; tw[0..3] stage a (pairs r,r+4),
// tw[4..5] stage b (pairs r,r+2), tw[6] stage c (pairs r,r+1)
__device__ __forceinline__ void pass3(float2 (&x)[8], const float2 (&tw)[7]) {
    bfly(x[0], x[4], tw[0]); bfly(x[1], x[5], tw[1]); bfly(x[2], x[6], tw[2]); bfly(x[3], x[7], tw[3]);
    bfly(x[0], x[2], tw[4]); bfly(x[1], x[3], tw[5]); bfly(x[4], x[6], tw[4]); bfly(x[5], x[7], tw[5]);
    bfly(x[0], x[1], tw[6]); bfly(x[2], x[3], tw[6]); bfly(x[4], x[5], tw[6]); bfly(x[6], x[7], tw[6]);
}

// The last three stages only meet the twiddles 1, -i and (1-i)/sqrt2, (-1-i)/sqrt2.  Written
// out, the general butterfly reduces to the forms below with every surviving operation rounded
// exactly as in the general formula (x*1 = x, x*0 = +-0 adds exactly, -(a*b) = a*(-b)), so the
// bits are those of the general radix-2 butterfly the CPU oracle runs.
__device__ __forceinline__ void bfly_one(float2& u, float2& v) {            // w = 1
    const float dr = u.x - v.x, di = u.y - v.y;
    u.x = u.x + v.x; u.y = u.y + v.y;
    v.x = dr; v.y = di;
}
__device__ __forceinline__ void bfly_mi(float2& u, float2& v) {             // w = -i
    const float dr = u.x - v.x, di = u.y - v.y;
    u.x = u.x + v.x; u.y = u.y + v.y;
    v.x = di; v.y = -dr;
}
__device__ __forceinline__ void bfly_w8(float2& u, float2& v, float c) {    // w = (c, -c)
    const float dr = u.x - v.x, di = u.y - v.y;
    u.x = u.x + v.x; u.y = u.y + v.y;
    const float a = dr * c, b = di * c;
    v.x = a + b; v.y = b - a;
}
__device__ __forceinline__ void bfly_w83(float2& u, float2& v, float c) {   // w = (-c, -c)
    const float dr = u.x - v.x, di = u.y - v.y;
    u.x = u.x + v.x; u.y = u.y + v.y;
    const float a = dr * c, b = di * c;
    v.x = b - a; v.y = -a - b;
}
__device__ __forceinline__ void pass3_last(float2 (&x)[8], float c) {
    bfly_one(x[0], x[4]); bfly_w8(x[1], x[5], c); bfly_mi(x[2], x[6]); bfly_w83(x[3], x[7], c);
    bfly_one(x[0], x[2]); bfly_mi(x[1], x[3]);    bfly_one(x[4], x[6]); bfly_mi(x[5], x[7]);
    bfly_one(x[0], x[1]); bfly_one(x[2], x[3]);   bfly_one(x[4], x[5]); bfly_one(x[6], x[7]);
}

__device__ __forceinline__ unsigned rev6(unsigned v) { return __brev(v) >> 26; }

template <int kBlocksPerWave>
__global__ __launch_bounds__(256)
void fft_bank_kernel(const float* __restrict__ dI, const float* __restrict__ dQ,
                     const int* __restrict__ seg_list, int blocks, float* __restrict__ ps,
                     const float* __restrict__ window, const float2* __restrict__ twiddle) {
    __shared__ float2 tile[kWavesPerWg * kTile];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int seg = seg_list ? seg_list[blockIdx.y] : (int)blockIdx.y;
    const int t_begin = (blockIdx.x * kWavesPerWg + wave) * kBlocksPerWave;
    if (t_begin >= blocks) return;
    const int t_end = min(t_begin + kBlocksPerWave, blocks);

    const float* __restrict__ si = dI + (size_t)seg * kIqStride;
    const float* __restrict__ sq = dQ + (size_t)seg * kIqStride;
    float* __restrict__ out = ps + (size_t)seg * kMaxBlocks * kPsStride;
    float2* X = tile + wave * kTile;

    const int a = lane >> 3, c = lane & 7;
    float win[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) win[r] = window[64 * r + lane];
    // pass A: element n = 64 r + lane ; pass B: n = 64 a + 8 r + c ; pass C: n = 8 lane + r
    const float2 twA[7] = {twiddle[lane], twiddle[64 + lane], twiddle[128 + lane], twiddle[192 + lane],
                           twiddle[2 * lane], twiddle[128 + 2 * lane], twiddle[4 * lane]};
    const float2 twB[7] = {twiddle[8 * c], twiddle[64 + 8 * c], twiddle[128 + 8 * c], twiddle[192 + 8 * c],
                           twiddle[16 * c], twiddle[128 + 16 * c], twiddle[32 * c]};
    const float w8 = twiddle[64].x;                 // cos(pi/4) as float; twiddle[64] = (w8, -w8)

    float ri[8], rq[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = kHop * t_begin + 64 * r + lane;
        ri[r] = si[k];
        rq[r] = sq[k];
    }

    for (int t = t_begin; t < t_end; ++t) {
        float2 x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = make_float2(ri[r] * win[r], rq[r] * win[r]);

        // slide the raw window by one hop (two 64-sample rows) while the FFT runs
        if (t + 1 < t_end) {
#pragma unroll
            for (int r = 0; r < 6; ++r) { ri[r] = ri[r + 2]; rq[r] = rq[r + 2]; }
            const int k = kHop * (t + 1) + 384 + lane;
            ri[6] = si[k];      rq[6] = sq[k];
            ri[7] = si[k + 64]; rq[7] = sq[k + 64];
        }

        pass3(x, twA);
#pragma unroll
        for (int r = 0; r < 8; ++r) X[72 * r + lane] = x[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = X[72 * a + 8 * r + c];
        __builtin_amdgcn_wave_barrier();

        pass3(x, twB);
#pragma unroll
        for (int r = 0; r < 8; ++r) X[72 * a + 9 * r + c] = x[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = X[9 * lane + r];
        __builtin_amdgcn_wave_barrier();

        pass3_last(x, w8);

        // x[r] now holds bin rev9(8*lane + r) = 64*rev3(r) + rev6(lane)
        float* __restrict__ row = out + (size_t)t * kPsStride;
        const int lo = (int)rev6((unsigned)lane);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int rev3 = ((r & 1) << 2) | (r & 2) | ((r >> 2) & 1);
            const int bin = ((64 * rev3 + lo) + kFftSize / 2) & (kFftSize - 1);   // fft-shift
            const int col = bin - kPsBin0;
            if (col >= 0 && col < kPsBins) {
                const float e1 = x[r].x * x[r].x, e2 = x[r].y * x[r].y;
                row[col] = e1 + e2;
            }
        }
    }
}
}  // namespace

// Calibration helper for the HBM PMC counters: a plain 4-byte-per-lane stream copy, the
// same access width as K1's loads/stores, over a known number of bytes.
namespace {
__global__ __launch_bounds__(256) void calib_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
}  // namespace
void launch_calib_copy(const float* src, float* dst, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(calib_copy_kernel, dim3(8192), dim3(256), 0, st, src, dst, n);
}

void launch_fft_bank(const float* dI, const float* dQ, const int* seg_list, int nseg_active,
                     int samples, float* ps, const DeviceTables& t, hipStream_t st) {
    const int blocks = 4 * (samples / kFftSize) - 1;
    if (blocks <= 0 || nseg_active <= 0) return;
    // consecutive FFTs per wave: longer runs re-read less input (run of R blocks loads R+3 hops)
    static const int bpw = [] { const char* e = getenv("WSPR_K1_BLOCKS_PER_WAVE"); return e ? atoi(e) : 8; }();
#define WSPR_K1(R)                                                                                          \
    do {                                                                                                    \
        const int per_wg = R * kWavesPerWg;                                                                 \
        dim3 grid((blocks + per_wg - 1) / per_wg, nseg_active);                                             \
        hipLaunchKernelGGL(fft_bank_kernel<R>, grid, dim3(256), 0, st, dI, dQ, seg_list, blocks, ps,        \
                           t.window, t.twiddle);                                                            \
    } while (0)
    if (bpw == 12) WSPR_K1(12);
    else if (bpw == 16) WSPR_K1(16);
    else if (bpw == 22) WSPR_K1(22);
    else if (bpw == 4) WSPR_K1(4);
    else WSPR_K1(8);
#undef WSPR_K1
}

}  // namespace wspr

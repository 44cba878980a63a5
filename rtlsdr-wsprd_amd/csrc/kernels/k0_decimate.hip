// K0 -- receiver front end: fs/4 mixer + 2-stage CIC (R = 6401) + 33-tap FIR,
// 2.4 Msps unsigned-8-bit IQ -> ~375 sps float32 IQ.
//
// Replaces reference rtlsdr_callback(), rtlsdr_wsprd.c:126-244 (zero initial
// state per segment = the receiver at start-up).  Bound: HBM bandwidth -- one
// pass over 576 000 000 B per 2-minute segment, ~0.02 op/B.
//
// The streaming recurrences are restated as exact integer block sums.  With
// x_r the mixed sample, R = 6401 and block b = samples [bR, (b+1)R):
//     S_b = sum_r x_r            W_b = sum_r (R - off_r) x_r      (off_r = r - bR)
//     I1(b) = I1(b-1) + S_b      I2(b) = I2(b-1) + R*I1(b-1) + W_b    (mod 2^32)
// are the two integrators sampled at the decimation instants, bit-for-bit,
// because int32 wrap-around arithmetic is associative.  The mixer is the
// reference's in-place int8 trick: multiply sample n by (1, j, -1, -j)[n & 3]
// with int8 negation (-(-128) stays -128).
//   pass A (HBM-bound): block sums, one workgroup per PAIR of blocks so that
//                       every 4-byte load is aligned (2R samples = 25 604 B);
//   pass B (tiny)     : serial scan of 45 000 block sums per segment + both combs;
//   pass C (tiny)     : 33-tap compensation FIR, taps summed in reference order.
#include "wspr_device.h"

#pragma clang fp contract(off)

namespace wspr {
namespace {

constexpr int kR = 6401;                 // decimation ratio (DOWNSAMPLING + 1)

__constant__ float kFirTaps[33] = {      // rtlsdr_wsprd.c:142-152
    -0.0027772683f, -0.0005058826f, 0.0049745750f, -0.0034059318f, -0.0077557814f, 0.0139375423f,
    0.0039896935f,  -0.0299394142f, 0.0162250643f, 0.0405130860f,  -0.0580746013f, -0.0272104968f,
    0.1183705475f,  -0.0306029022f, -0.2011241667f, 0.1615898423f, 0.5000000000f,  0.1615898423f,
    -0.2011241667f, -0.0306029022f, 0.1183705475f, -0.0272104968f, -0.0580746013f, 0.0405130860f,
    0.0162250643f,  -0.0299394142f, 0.0039896935f, 0.0139375423f,  -0.0077557814f, -0.0034059318f,
    0.0049745750f,  -0.0005058826f, -0.0027772683f};

__device__ __forceinline__ int s8(unsigned b) { return (int)(b & 0xffu) - 128; }        // (int8)(b ^ 0x80)
__device__ __forceinline__ int neg8(int v) { return (v == -128) ? -128 : -v; }          // int8 negate

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sums[seg][block][4] = {S_I, S_Q, W_I, W_Q}
__global__ __launch_bounds__(256)
void cic_block_sums_kernel(const uint8_t* __restrict__ raw, size_t bytes_per_seg, int nblocks,
                           int32_t* __restrict__ sums) {
    __shared__ unsigned red[4][8];
    const int seg = blockIdx.y, pair = blockIdx.x, tid = threadIdx.x;
    const int blkA = 2 * pair;
    const uint32_t* __restrict__ words =
        reinterpret_cast<const uint32_t*>(raw + (size_t)seg * bytes_per_seg + (size_t)pair * 2 * kR * 2);
    const bool haveB = (blkA + 1) < nblocks;
    const int nsamp = haveB ? 2 * kR : kR;                // samples owned by this workgroup
    const unsigned phase0 = (unsigned)((2 * pair) & 3);   // (2R*pair) mod 4, R odd

    // all sums are modulo 2^32 (the reference's int32 integrators wrap): unsigned math
    unsigned acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};           // A: SI SQ WI WQ, B: SI SQ WI WQ
    const int nwords = (nsamp + 1) / 2;
    const size_t seg_samples = bytes_per_seg / 2;
    const size_t pair_first = (size_t)pair * 2 * kR;
    for (int d = tid; d < nwords; d += 256) {
        // the odd-length tail word may straddle the end of the segment: read 2 bytes there
        const bool whole = pair_first + 2 * (size_t)d + 1 < seg_samples;
        const uint32_t wv = whole ? words[d]
                                  : (uint32_t)reinterpret_cast<const uint16_t*>(words)[2 * d];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int idx = 2 * d + h;
            if (idx < nsamp) {
                const int a = s8(wv >> (16 * h)), b = s8(wv >> (16 * h + 8));
                int xi, xq;
                switch ((phase0 + (unsigned)idx) & 3u) {
                    case 0:  xi = a;        xq = b;        break;
                    case 1:  xi = neg8(b);  xq = a;        break;
                    case 2:  xi = neg8(a);  xq = neg8(b);  break;
                    default: xi = b;        xq = neg8(a);  break;
                }
                const int inB = idx >= kR;
                const int wgt = kR - (idx - (inB ? kR : 0));
                acc[4 * inB + 0] += (unsigned)xi;
                acc[4 * inB + 1] += (unsigned)xq;
                acc[4 * inB + 2] += (unsigned)wgt * (unsigned)xi;
                acc[4 * inB + 3] += (unsigned)wgt * (unsigned)xq;
            }
        }
    }
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const unsigned s = wave_sum(acc[q]);
        if (lane == 0) red[wave][q] = s;
    }
    __syncthreads();
    if (tid < 8) {
        const unsigned s = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        const int blk = blkA + (tid >> 2);
        if (blk < nblocks) sums[((size_t)seg * nblocks + blk) * 4 + (tid & 3)] = (int32_t)s;
    }
}

// integrators at the decimation instants + two combs (delay 2), one thread per
// (segment, rail); output = comb result converted to float, in place of sums
__global__ void cic_scan_kernel(const int32_t* __restrict__ sums, int nblocks, float* __restrict__ comb) {
    const int seg = blockIdx.x, rail = threadIdx.x;       // rail 0 = I, 1 = Q
    if (rail > 1) return;
    const int32_t* __restrict__ s = sums + (size_t)seg * nblocks * 4;
    float* __restrict__ out = comb + ((size_t)seg * 2 + rail) * nblocks;
    uint32_t x1 = 0, x2 = 0, c1a = 0, c1b = 0, c2a = 0, c2b = 0;
    for (int b = 0; b < nblocks; ++b) {
        const uint32_t S = (uint32_t)s[4 * b + rail], W = (uint32_t)s[4 * b + 2 + rail];
        x2 = x2 + (uint32_t)kR * x1 + W;
        x1 = x1 + S;
        const uint32_t y1 = x2 - c1b;  c1b = c1a;  c1a = x2;      // rtlsdr_wsprd.c:204-210
        const uint32_t y2 = y1 - c2b;  c2b = c2a;  c2a = y1;      // :212-218
        out[b] = (float)(int32_t)y2;
    }
}

// 33-tap FIR: 32 previous comb outputs (oldest first) on taps 0..31, then the new
// one on tap 32 (rtlsdr_wsprd.c:220-234)
__global__ __launch_bounds__(256)
void cic_fir_kernel(const float* __restrict__ comb, int nblocks, float* __restrict__ dI,
                    float* __restrict__ dQ, int* __restrict__ n_out) {
    const int seg = blockIdx.y;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m == 0 && n_out) n_out[seg] = nblocks < kMaxSamples ? nblocks : kMaxSamples;
    if (m >= nblocks || m >= kMaxSamples) return;
#pragma unroll
    for (int rail = 0; rail < 2; ++rail) {
        const float* __restrict__ c = comb + ((size_t)seg * 2 + rail) * nblocks;
        float acc = 0.0f;
        for (int j = 0; j < 32; ++j) {
            const int src = m - 32 + j;
            const float v = (src >= 0) ? c[src] : 0.0f;
            const float p = v * kFirTaps[j];
            acc += p;
        }
        const float p = c[m] * kFirTaps[32];
        acc += p;
        (rail == 0 ? dI : dQ)[(size_t)seg * kIqStride + m] = acc;
    }
}
}  // namespace

// scratch: nseg * nblocks * (4 int32 + 2 float)
void launch_decimate(const uint8_t* raw, size_t bytes_per_seg, int nseg, float* dI, float* dQ,
                     int* n_out, int32_t* scratch, hipStream_t st) {
    if (nseg <= 0) return;
    const size_t nsamp = bytes_per_seg / 2;
    const int nblocks = (int)(nsamp / kR);
    if (nblocks <= 0) return;
    int32_t* sums = scratch;
    float* comb = reinterpret_cast<float*>(scratch + (size_t)nseg * nblocks * 4);
    hipLaunchKernelGGL(cic_block_sums_kernel, dim3((nblocks + 1) / 2, nseg), dim3(256), 0, st, raw,
                       bytes_per_seg, nblocks, sums);
    hipLaunchKernelGGL(cic_scan_kernel, dim3(nseg), dim3(64), 0, st, sums, nblocks, comb);
    hipLaunchKernelGGL(cic_fir_kernel, dim3((nblocks + 255) / 256, nseg), dim3(256), 0, st, comb, nblocks,
                       dI, dQ, n_out);
}

}  // namespace wspr

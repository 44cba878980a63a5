// K2 -- candidate peak picker, K3 -- coarse (frequency, lag, drift) sync search.
//
// K2 replaces reference wsprd/wsprd.c:555-631, K3 replaces :646-678.  Both read
// the power spectrogram written by K1 and are bound by that read (HBM/L2):
// 578 796 algorithmic bytes per segment per pass.
//
// Float evaluation order is the reference's: every sum that feeds a threshold or
// an argmax is accumulated by ONE lane in the reference's loop order; lanes
// parallelise over independent sums (bins, or (freq, lag, drift) hypotheses).
#include "wspr_device.h"
#include <cstdlib>

#pragma clang fp contract(off)

namespace wspr {
namespace {

constexpr double kHalfDf = 375.0 / 256.0 / 2.0;      // (DF / 2.0), wsprd.c:615

// ------------------------------------------------------------------ K2 ------
// K2a: time-averaged spectrum.  lane = bin, serial over the 347 time blocks in reference order
// (wsprd.c:556-561).  Pure streaming read of ps (the HBM-bound part of K2): one wave per 64 bins
// so that thousands of independent waves keep the memory system busy.
__global__ __launch_bounds__(64)
void time_average_kernel(const float* __restrict__ ps, const int* __restrict__ seg_list, int blocks,
                         float* __restrict__ psavg) {
    const int seg = seg_list ? seg_list[blockIdx.y] : (int)blockIdx.y;
    const int col = 4 * (blockIdx.x * 64 + threadIdx.x);          // 4 bins (16 B) per lane
    if (col >= kPsBins) return;                                     // columns 417..431 of a row are padding
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* __restrict__ P = reinterpret_cast<const f4*>(ps + (size_t)seg * kMaxBlocks * kPsStride + col);
    constexpr int kRow4 = kPsStride / 4;
    // four independent serial chains per lane; batch the loads ahead of the adds
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    constexpr int kUnroll = 8;
    int t = 0;
    for (; t + kUnroll <= blocks; t += kUnroll) {
        f4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = P[(size_t)(t + u) * kRow4];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; t < blocks; ++t) { const f4 v = P[(size_t)t * kRow4]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    *reinterpret_cast<float4*>(psavg + (size_t)seg * kPsStride + col) = acc;
}

// K2b: one workgroup per segment, everything after the time average (wsprd.c:565-631)
__global__ __launch_bounds__(512)
void pick_peaks_kernel(const float* __restrict__ psavg, const int* __restrict__ seg_list,
                       DevCand* __restrict__ cand, int* __restrict__ npk_out,
                       float* __restrict__ noise_out, float* __restrict__ smspec_out,
                       float min_snr, float floor_snr) {
    __shared__ float avg[kPsBins];
    __shared__ float sm[kSmooth];
    __shared__ float nrm[kSmooth];
    __shared__ float noise_s;
    __shared__ int   pk_bin[kMaxCand];
    __shared__ float pk_snr[kMaxCand];
    __shared__ int   kept_s;

    const int tid = threadIdx.x;
    const int seg = seg_list ? seg_list[blockIdx.x] : (int)blockIdx.x;
    if (tid < kPsBins) avg[tid] = psavg[(size_t)seg * kPsStride + tid];
    if (tid == 0) noise_s = 0.0f;
    __syncthreads();

    // 7-bin boxcar, bins 51+i-3 .. 51+i+3 (wsprd.c:565-573); column = bin - 48
    if (tid < kSmooth) {
        float acc = 0.0f;
#pragma unroll
        for (int d = 0; d < 7; ++d) acc += avg[tid + d];
        sm[tid] = acc;
        if (smspec_out) smspec_out[(size_t)seg * kSmooth + tid] = acc;
    }
    __syncthreads();

    // 30th percentile = element 122 of the ascending sort (wsprd.c:576-583), by rank counting
    if (tid < kSmooth) {
        const float v = sm[tid];
        int less = 0, leq = 0;
        for (int j = 0; j < kSmooth; ++j) {
            const float u = sm[j];
            less += (u < v);
            leq  += (u <= v);
        }
        if (less <= 122 && 122 < leq) noise_s = v;
    }
    __syncthreads();
    const float noise = noise_s;

    // snr-like normalisation with floor (wsprd.c:590-597)
    if (tid < kSmooth) {
        const float q = sm[tid] / noise;
        float v = (float)((double)q - 1.0);
        if (v < min_snr) v = floor_snr;
        nrm[tid] = v;
    }
    __syncthreads();

    // strict local maxima, first 200 only, then the +-110 Hz window (wsprd.c:608-629): ordered
    // compaction with wave ballots (the reference walks j = 1..409 once)
    __shared__ unsigned long long bal_peak[8], bal_keep[8];
    const int wv = tid >> 6, ln = tid & 63;
    const unsigned long long below = (ln == 0) ? 0ull : (~0ull >> (64 - ln));
    bool peak = false;
    if (tid >= 1 && tid < kSmooth - 1) {
        const float v = nrm[tid];
        peak = (v > nrm[tid - 1]) && (v > nrm[tid + 1]);
    }
    const unsigned long long bp = __ballot(peak);
    if (ln == 0) bal_peak[wv] = bp;
    __syncthreads();
    int found_before = __popcll(bp & below);
    for (int q = 0; q < wv; ++q) found_before += __popcll(bal_peak[q]);
    const float fj = (float)((double)(tid - 205) * kHalfDf);
    const bool keep = peak && (found_before < kMaxCand) && (fj >= -110.0f) && (fj <= 110.0f);
    const unsigned long long bk = __ballot(keep);
    if (ln == 0) bal_keep[wv] = bk;
    __syncthreads();
    if (keep) {
        int idx = __popcll(bk & below);
        for (int q = 0; q < wv; ++q) idx += __popcll(bal_keep[q]);
        pk_bin[idx] = tid;
        pk_snr[idx] = (float)(10.0 * (double)log10f(nrm[tid]) - (double)26.3f);
    }
    if (tid == 0) {
        int kept = 0;
        for (int q = 0; q < 8; ++q) kept += __popcll(bal_keep[q]);
        kept_s = kept;
        npk_out[seg] = kept;
        if (noise_out) noise_out[seg] = noise;
    }
    __syncthreads();

    // stable sort by snr, strongest first (wsprd.c:631; glibc qsort is a merge sort)
    const int kept = kept_s;
    if (tid < kept) {
        const float mine = pk_snr[tid];
        int rank = 0;
        for (int j = 0; j < kept; ++j) {
            const float o = pk_snr[j];
            rank += (o > mine) || (o == mine && j < tid);
        }
        const int j = pk_bin[tid];
        DevCand cd;
        cd.freq  = (float)((double)(j - 205) * kHalfDf);
        cd.snr   = mine;
        cd.peak  = nrm[j];
        cd.shift = 0;
        cd.drift = 0.0f;
        cd.sync  = 0.0f;
        cand[(size_t)seg * kMaxCand + rank] = cd;
    }
}

// ------------------------------------------------------------------ K3 ------
// One workgroup per (segment, candidate); lane = one (frequency bin, lag, drift
// pattern) hypothesis, accumulating its 162-term sums in symbol order.
//
// Reference quirks reproduced:
//  * "/ DF" expands to "/375.0/256.0", so the per-symbol drift offset only ever
//    lowers the bin by one for (k>81, drift<0) or (k<81, drift>0): three distinct
//    patterns; with a strict '>' the first drift of each sign wins, i.e. the label
//    is -maxdrift, 0 or +1.
//  * a negative time index reads the previous bin's row, 347+index.
constexpr int kCoarseRows = 11;
constexpr int kCoarsePitch = 352;

__global__ __launch_bounds__(320)
void coarse_sync_kernel(const float* __restrict__ ps, const int* __restrict__ seg_list, int blocks,
                        DevCand* __restrict__ cand, const int* __restrict__ npk, int maxdrift,
                        const unsigned char* __restrict__ pr3) {
    __shared__ float amp[kCoarseRows * kCoarsePitch];
    __shared__ float res[288];
    __shared__ float best_s;
    __shared__ int arg_s;
    const int tid = threadIdx.x;
    const int seg = seg_list ? seg_list[blockIdx.x] : (int)blockIdx.x;
    const float* __restrict__ P = ps + (size_t)seg * kMaxBlocks * kPsStride;
    const int ncand = npk[seg];
    const int npat = (maxdrift > 0) ? 3 : 1;
    const int nhyp = 3 * 32 * npat;

    for (int c = blockIdx.y; c < ncand; c += gridDim.y) {
        DevCand cd = cand[(size_t)seg * kMaxCand + c];
        const int if0 = (int)((double)cd.freq / kHalfDf + 256.0);
        const int col0 = if0 - 6 - kPsBin0;            // first staged column
        __syncthreads();
        // stage sqrt(ps) for the 11 columns this candidate can touch; loads are issued in batches
        // of 4 so that their latencies overlap
        const int total = kCoarseRows * blocks;
        for (int e0 = tid; e0 < total; e0 += 4 * (int)blockDim.x) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * (int)blockDim.x;
                const int t = e / kCoarseRows, r = e - t * kCoarseRows;
                v[u] = (e < total) ? P[(size_t)t * kPsStride + col0 + r] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * (int)blockDim.x;
                const int t = e / kCoarseRows, r = e - t * kCoarseRows;
                if (e < total) amp[r * kCoarsePitch + t] = sqrtf(v[u]);
            }
        }
        __syncthreads();

        if (tid < nhyp) {
            const int fi = tid / (32 * npat);
            const int rem = tid - fi * 32 * npat;
            const int k0 = rem / npat - 10;
            const int pat = (npat == 3) ? rem % 3 : 1;     // 0: drift<0, 1: drift 0, 2: drift>0
            const int ifr = if0 - 1 + fi;
            float ss = 0.0f, pw = 0.0f;
            bool any = false;
#pragma unroll 6
            for (int k = 0; k < kNSymD; ++k) {
                const int low = (pat == 0 && k > 81) || (pat == 2 && k < 81);
                const int ifd = ifr - low;
                int kidx = k0 + 2 * k;
                if (kidx < blocks) {
                    int row = ifd - 3 - (if0 - 6);
                    if (kidx < 0) { row -= 1; kidx += blocks; }   // previous row of the flat array
                    const float* a = amp + row * kCoarsePitch + kidx;
                    const float p0 = a[0], p1 = a[2 * kCoarsePitch], p2 = a[4 * kCoarsePitch], p3 = a[6 * kCoarsePitch];
                    const float m = (p1 + p3) - (p0 + p2);
                    ss = pr3[k] ? ss + m : ss - m;
                    pw = pw + p0 + p1 + p2 + p3;
                    any = true;
                }
            }
            res[tid] = any ? ss / pw : __int_as_float(0x7fc00000);
        }
        __syncthreads();
        // first hypothesis (in the reference's loop order) with the strictly largest metric:
        // max by value, ties to the lowest index -- an order-independent reduction
        if (tid < 64) {
            float best = -1e30f;
            int arg = -1;
            for (int h = tid; h < nhyp; h += 64)
                if (res[h] > best) { best = res[h]; arg = h; }       // ascending h: keeps the first
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o);
                const int oa = __shfl_xor(arg, o);
                const bool take = (oa >= 0) && (arg < 0 || ob > best || (ob == best && oa < arg));
                if (take) { best = ob; arg = oa; }
            }
            if (tid == 0) { best_s = best; arg_s = arg; }
        }
        __syncthreads();
        if (tid == 0) {
            const float best = best_s;
            const int arg = arg_s;
            if (arg >= 0) {
                const int fi = arg / (32 * npat);
                const int rem = arg - fi * 32 * npat;
                const int k0 = rem / npat - 10;
                const int pat = (npat == 3) ? rem % 3 : 1;
                cd.shift = 128 * (k0 + 1);
                cd.drift = (pat == 0) ? (float)(-maxdrift) : (pat == 2 ? 1.0f : 0.0f);
                cd.freq  = (float)((double)(if0 - 1 + fi - 256) * kHalfDf);
                cd.sync  = best;
                cand[(size_t)seg * kMaxCand + c] = cd;
            }
        }
    }
}
}  // namespace

void launch_time_average(const float* ps, const int* seg_list, int nseg_active, int blocks, float* psavg,
                         hipStream_t st) {
    if (nseg_active <= 0) return;
    hipLaunchKernelGGL(time_average_kernel, dim3((kPsBins + 255) / 256, nseg_active), dim3(64), 0, st, ps, seg_list,
                       blocks, psavg);
}

void launch_pick_peaks(const float* ps, const int* seg_list, int nseg_active, int blocks, float* psavg,
                       DevCand* cand, int* npk, float* noise_out, float* smspec_out,
                       const DeviceTables& t, hipStream_t st, bool have_avg) {
    if (nseg_active <= 0) return;
    if (!have_avg) launch_time_average(ps, seg_list, nseg_active, blocks, psavg, st);
    hipLaunchKernelGGL(pick_peaks_kernel, dim3(nseg_active), dim3(512), 0, st, psavg, seg_list,
                       cand, npk, noise_out, smspec_out, t.min_snr, t.floor_snr);
}

void launch_coarse_sync(const float* ps, const int* seg_list, int nseg_active, int blocks,
                        DevCand* cand, const int* npk, int maxdrift,
                        const DeviceTables& t, hipStream_t st) {
    if (nseg_active <= 0) return;
    hipLaunchKernelGGL(coarse_sync_kernel, dim3(nseg_active, 8), dim3(320), 0, st, ps, seg_list, blocks,
                       cand, npk, maxdrift, t.sync);
}

}  // namespace wspr

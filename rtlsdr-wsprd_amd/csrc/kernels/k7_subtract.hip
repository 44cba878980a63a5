// K7 -- coherent subtraction of a decoded signal from the segment's IQ.
//
// Replaces reference subtract_signal2(), wsprd/wsprd.c:316-413:
//   r(t) = exp(j*phi(t))  regenerated from the 162 channel symbols,
//   c(t) = LPF[s(t) * conj(r(t))]   360-tap normalised sine window,
//   s'(t) = s(t) - c(t) r(t) / edge_norm.
//
// Parity rules kept: phi is the reference's *float* running sum (41 472 serial
// adds, dphi changes every 256 samples) -- one lane per job walks it 64 steps at
// a time and hands the 64 phases to the wave through LDS, the other lanes then
// evaluate glibc-exact cosf/sinf and the s*conj(r) products in parallel; each
// low-pass output is one lane's serial 360-term sum in tap order.
// Bound: fp32 VALU + LDS (64 MFLOP per job), not HBM.
#include "wspr_device.h"
#include "glibc_sincosf.h"

#pragma clang fp contract(off)

namespace wspr {
namespace {

constexpr double kTwoPiDt = 2.0 * 3.14159265358979323846 * 1.0 / 375.0;

// scratch layout per job: refi | refq | ci | cq   (each kSigLen floats)
__global__ __launch_bounds__(64)
void sub_reference_kernel(const float* __restrict__ dI, const float* __restrict__ dQ, int np,
                          const SubJob* __restrict__ jobs, float* __restrict__ scratch) {
    __shared__ float dphi_s[kNSymD];
    __shared__ float phi_s[64];
    const int lane = threadIdx.x;
    const SubJob* job = jobs + blockIdx.x;
    const int seg = job->seg, shift = job->shift;
    const float f0 = job->f0, drift = job->drift;
    float* __restrict__ refi = scratch + (size_t)blockIdx.x * 4 * kSigLen;
    float* __restrict__ refq = refi + kSigLen;
    float* __restrict__ ci = refq + kSigLen;
    float* __restrict__ cq = ci + kSigLen;
    const float* __restrict__ xi = dI + (size_t)seg * kIqStride;
    const float* __restrict__ xq = dQ + (size_t)seg * kIqStride;

    for (int i = lane; i < kNSymD; i += 64) {
        const float cs = (float)job->sym[i];
        // wsprd.c:343, all double: TWOPIDT*(f0 + (drift/2.0)*(i - 81.0)/81.0 + (cs - 1.5)*375.0/256.0)
        const double arg = (double)f0 + ((double)drift / 2.0) * ((double)(float)i - 81.0) / 81.0
                           + ((double)cs - 1.5) * 375.0 / 256.0;
        dphi_s[i] = (float)(kTwoPiDt * arg);
    }
    __syncthreads();

    float phi = 0.0f;                       // meaningful in lane 0 only
    for (int chunk = 0; chunk < kSigLen / 64; ++chunk) {
        if (lane == 0) {
            const float d = dphi_s[chunk >> 2];          // 256 samples = 4 chunks per symbol
#pragma unroll 16
            for (int u = 0; u < 64; ++u) { phi_s[u] = phi; phi = phi + d; }
        }
        __syncthreads();
        const int n = chunk * 64 + lane;
        const float ph = phi_s[lane];
        const float cr = glibc_cosf(ph), sr = glibc_sinf(ph);
        refi[n] = cr;
        refq[n] = sr;
        const int k = shift + n;
        float a = 0.0f, b = 0.0f;
        if (k > 0 && k < np) {
            const float x = xi[k], y = xq[k];
            const float p1 = x * cr, p2 = y * sr, p3 = y * cr, p4 = x * sr;
            a = p1 + p2;                  // Re{s conj(r)}
            b = p3 - p4;                  // Im{s conj(r)}
        }
        ci[n] = a;
        cq[n] = b;
        __syncthreads();
    }
}

constexpr int kFirBlock = 256;
constexpr int kFirSpan = kFirBlock + kLpfTaps - 1;      // 615

__global__ __launch_bounds__(kFirBlock)
void sub_filter_kernel(float* __restrict__ dI, float* __restrict__ dQ, int np,
                       const SubJob* __restrict__ jobs, const float* __restrict__ scratch,
                       const float* __restrict__ lpf, const float* __restrict__ lpf_part) {
    __shared__ float ti[kFirSpan], tq[kFirSpan], w[kLpfTaps];
    const int tid = threadIdx.x;
    const SubJob* job = jobs + blockIdx.y;
    const float* __restrict__ refi = scratch + (size_t)blockIdx.y * 4 * kSigLen;
    const float* __restrict__ refq = refi + kSigLen;
    const float* __restrict__ ci = refq + kSigLen;
    const float* __restrict__ cq = ci + kSigLen;
    const int n0 = blockIdx.x * kFirBlock;

    // the reference filters a zero-padded copy (360 leading zeros); outside the
    // signal the products are zero and adding them is exact
    for (int e = tid; e < kFirSpan; e += kFirBlock) {
        const int n = n0 - kLpfTaps / 2 + e;
        const bool in = (n >= 0) && (n < kSigLen);
        ti[e] = in ? ci[n] : 0.0f;
        tq[e] = in ? cq[n] : 0.0f;
    }
    for (int e = tid; e < kLpfTaps; e += kFirBlock) w[e] = lpf[e];
    __syncthreads();

    const int n = n0 + tid;
    if (n >= kSigLen) return;
    float si = 0.0f, sq = 0.0f;
#pragma unroll 8
    for (int j = 0; j < kLpfTaps; ++j) {
        const float a = w[j] * ti[tid + j], b = w[j] * tq[tid + j];
        si = si + a;
        sq = sq + b;
    }
    float norm = 1.0f;                                   // wsprd.c:397-404
    if (n < kLpfTaps / 2)                    norm = lpf_part[kLpfTaps / 2 + n];
    else if (n > kSigLen - 1 - kLpfTaps / 2) norm = lpf_part[kLpfTaps / 2 + kSigLen - 1 - n];
    const int k = job->shift + n;
    if (k > 0 && k < np) {
        float* __restrict__ xi = dI + (size_t)job->seg * kIqStride;
        float* __restrict__ xq = dQ + (size_t)job->seg * kIqStride;
        const float cr = refi[n], sr = refq[n];
        const float a = si * cr, b = sq * sr, c = si * sr, d = sq * cr;
        const float ri = a - b, rq = c + d;
        xi[k] = xi[k] - ri / norm;
        xq[k] = xq[k] - rq / norm;
    }
}

// ---- receiver normalisation, rtlsdr_wsprd.c:284-305 -------------------------
__global__ __launch_bounds__(1024)
void normalise_kernel(float* __restrict__ dI, float* __restrict__ dQ, const int* __restrict__ n_valid,
                      int n_total) {
    __shared__ float red[1024];
    const int seg = blockIdx.x, tid = threadIdx.x;
    float* __restrict__ xi = dI + (size_t)seg * kIqStride;
    float* __restrict__ xq = dQ + (size_t)seg * kIqStride;
    const int nv = n_valid ? n_valid[seg] : n_total;
    float peak = 1e-24f;
    for (int i = tid; i < n_total; i += 1024) {
        float a = xi[i], b = xq[i];
        if (i >= nv) { a = 0.0f; b = 0.0f; xi[i] = 0.0f; xq[i] = 0.0f; }
        peak = fmaxf(peak, fmaxf(fabsf(a), fabsf(b)));       // max is order-independent
    }
    red[tid] = peak;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    const float scale = (float)(0.5 / (double)red[0]);
    for (int i = tid; i < n_total; i += 1024) { xi[i] = xi[i] * scale; xq[i] = xq[i] * scale; }
}
}  // namespace

void launch_subtract(float* dI, float* dQ, int samples, const SubJob* jobs, int njobs,
                     float* scratch, const DeviceTables& t, hipStream_t st) {
    if (njobs <= 0) return;
    hipLaunchKernelGGL(sub_reference_kernel, dim3(njobs), dim3(64), 0, st, dI, dQ, samples, jobs, scratch);
    hipLaunchKernelGGL(sub_filter_kernel, dim3((kSigLen + kFirBlock - 1) / kFirBlock, njobs), dim3(kFirBlock), 0, st,
                       dI, dQ, samples, jobs, scratch, t.lpf, t.lpf_part);
}

void launch_normalise(float* dI, float* dQ, const int* n_valid, int nseg, int n_total, hipStream_t st) {
    if (nseg <= 0) return;
    hipLaunchKernelGGL(normalise_kernel, dim3(nseg), dim3(1024), 0, st, dI, dQ, n_valid, n_total);
}

}  // namespace wspr

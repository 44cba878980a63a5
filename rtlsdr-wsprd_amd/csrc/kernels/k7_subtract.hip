// K7 -- coherent subtraction of a decoded signal from the segment's IQ.
//
// Replaces reference subtract_signal2(), wsprd/wsprd.c:316-413:
//   r(t) = exp(j*phi(t))  regenerated from the 162 channel symbols,
//   c(t) = LPF[s(t) * conj(r(t))]   360-tap normalised sine window,
//   s'(t) = s(t) - c(t) r(t) / edge_norm.
//
// Three kernels, each keeping the reference's evaluation order where it matters:
//   sub_phase_kernel   phi is the reference's *float* running sum (41 472 serial adds,
//                      dphi changes every 256 samples).  One lane per job walks it; a wave
//                      covers 64 jobs and transposes 64x64 blocks through LDS so that the
//                      stores to HBM are coalesced rows.
//   sub_ref_kernel     fully parallel: glibc-exact sincos of every phase (one shared
//                      argument reduction), r and the products s*conj(r).
//   sub_filter_kernel  each low-pass output is one serial 360-term sum in tap order; a lane
//                      owns 4 consecutive outputs and slides a 4-sample register window, the
//                      tile is stored transposed-by-4 in LDS so the per-step read is
//                      conflict-free; the epilogue subtracts c*r/norm in place.
// Bound: fp32 VALU (64 MFLOP of separately rounded mul/add per job), not HBM.
#include "wspr_device.h"
#include "glibc_sincosf.h"

#pragma clang fp contract(off)

namespace wspr {
namespace {

constexpr double kTwoPiDt = 2.0 * 3.14159265358979323846 * 1.0 / 375.0;

// scratch (floats): phiT[kSigLen][njobs_pad] (job fastest) | per job: ref[kSigLen] float2 | cc[kSigLen] float2
constexpr size_t kSubPerJob = 4 * (size_t)kSigLen;
__host__ __device__ inline size_t jobs_padded(int njobs) { return ((size_t)njobs + 63) / 64 * 64; }

// lane = job: the serial float phase walk; one coalesced 256-B store per step per wave
__global__ __launch_bounds__(64)
void sub_phase_kernel(const SubJob* __restrict__ jobs, int njobs, float* __restrict__ phiT) {
    const int job = blockIdx.x * 64 + threadIdx.x;
    const SubJob* jb = jobs + (job < njobs ? job : 0);
    const float f0 = jb->f0, drift = jb->drift;
    const size_t pitch = jobs_padded(njobs);
    float* __restrict__ out = phiT + job;
    float phi = 0.0f;
    for (int i = 0; i < kNSymD; ++i) {
        const float cs = (float)jb->sym[i];
        // wsprd.c:343, all double: TWOPIDT*(f0 + (drift/2.0)*(i - 81.0)/81.0 + (cs - 1.5)*375.0/256.0)
        const double arg = (double)f0 + ((double)drift / 2.0) * ((double)(float)i - 81.0) / 81.0
                           + ((double)cs - 1.5) * 375.0 / 256.0;
        const float dphi = (float)(kTwoPiDt * arg);
#pragma unroll 8
        for (int j = 0; j < kSps; ++j) {
            out[(size_t)(i * kSps + j) * pitch] = phi;
            phi = phi + dphi;
        }
    }
}

// 64 jobs x 64 samples per workgroup: phases arrive job-fastest, are transposed through LDS,
// then each wave owns one job row at a time (sample-fastest, coalesced)
__global__ __launch_bounds__(256)
void sub_ref_kernel(const float* __restrict__ dI, const float* __restrict__ dQ, int np,
                    const SubJob* __restrict__ jobs, int njobs, const float* __restrict__ phiT,
                    float* __restrict__ perjob) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const size_t pitch = jobs_padded(njobs);
    for (int r = wave; r < 64; r += 4) tile[r][lane] = phiT[(size_t)(n0 + r) * pitch + j0 + lane];
    __syncthreads();
    for (int q = wave; q < 64; q += 4) {
        const int jobi = j0 + q;
        if (jobi >= njobs) break;
        const SubJob* job = jobs + jobi;
        const int n = n0 + lane;
        float2* __restrict__ ref = reinterpret_cast<float2*>(perjob + (size_t)jobi * kSubPerJob);
        float2* __restrict__ cc = ref + kSigLen;
        float sr, cr;
        glibc_sincosf_pair(tile[lane][q], &sr, &cr);
        ref[n] = make_float2(cr, sr);
        const int k = job->shift + n;
        float a = 0.0f, b = 0.0f;
        if (k > 0 && k < np) {
            const float x = dI[(size_t)job->seg * kIqStride + k], y = dQ[(size_t)job->seg * kIqStride + k];
            const float p1 = x * cr, p2 = y * sr, p3 = y * cr, p4 = x * sr;
            a = p1 + p2;                  // Re{s conj(r)}
            b = p3 - p4;                  // Im{s conj(r)}
        }
        cc[n] = make_float2(a, b);
    }
}

constexpr int kFirThreads = 256;
constexpr int kFirPerLane = 4;
constexpr int kFirOut = kFirThreads * kFirPerLane;             // 1024 outputs per workgroup
constexpr int kFirSpan = kFirOut + kLpfTaps - 1;               // 1383 inputs
constexpr int kFirPitch = (kFirSpan + 3) / 4 + 1;              // transposed-by-4 row pitch

__global__ __launch_bounds__(kFirThreads)
void sub_filter_kernel(float* __restrict__ dI, float* __restrict__ dQ, int np,
                       const SubJob* __restrict__ jobs, const float* __restrict__ perjob,
                       const float* __restrict__ lpf, const float* __restrict__ lpf_part) {
    __shared__ float2 tile[4 * kFirPitch];
    __shared__ float w[kLpfTaps];
    const int tid = threadIdx.x;
    const SubJob* job = jobs + blockIdx.y;
    const float2* __restrict__ ref = reinterpret_cast<const float2*>(perjob + (size_t)blockIdx.y * kSubPerJob);
    const float2* __restrict__ cc = ref + kSigLen;
    const int n0 = blockIdx.x * kFirOut;

    // the reference filters a zero-padded copy (360 leading zeros); outside the signal the
    // products are zero and adding them is exact
    for (int e = tid; e < kFirSpan; e += kFirThreads) {
        const int n = n0 - kLpfTaps / 2 + e;
        const bool in = (n >= 0) && (n < kSigLen);
        tile[(e & 3) * kFirPitch + (e >> 2)] = in ? cc[n] : make_float2(0.0f, 0.0f);
    }
    for (int e = tid; e < kLpfTaps; e += kFirThreads) w[e] = lpf[e];
    __syncthreads();

    // outputs n0 + 4*tid + r, r = 0..3; input index of tap j for output r: 4*tid + r + j
    float si[4] = {0.0f, 0.0f, 0.0f, 0.0f}, sq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float2 x[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = tile[r * kFirPitch + tid];            // e = 4*tid + r
#pragma unroll 4
    for (int j = 0; j < kLpfTaps; ++j) {
        const float wj = w[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = wj * x[r].x, b = wj * x[r].y;
            si[r] = si[r] + a;
            sq[r] = sq[r] + b;
        }
        x[0] = x[1]; x[1] = x[2]; x[2] = x[3];
        const int e = 4 * tid + j + 4;                                       // next input of output 3
        x[3] = tile[(e & 3) * kFirPitch + (e >> 2)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + 4 * tid + r;
        if (n >= kSigLen) break;
        float norm = 1.0f;                                   // wsprd.c:397-404
        if (n < kLpfTaps / 2)                    norm = lpf_part[kLpfTaps / 2 + n];
        else if (n > kSigLen - 1 - kLpfTaps / 2) norm = lpf_part[kLpfTaps / 2 + kSigLen - 1 - n];
        const int k = job->shift + n;
        if (k > 0 && k < np) {
            float* __restrict__ xi = dI + (size_t)job->seg * kIqStride;
            float* __restrict__ xq = dQ + (size_t)job->seg * kIqStride;
            const float2 rr = ref[n];
            const float a = si[r] * rr.x, b = sq[r] * rr.y, c = si[r] * rr.y, d = sq[r] * rr.x;
            const float ri = a - b, rq = c + d;
            xi[k] = xi[k] - ri / norm;
            xq[k] = xq[k] - rq / norm;
        }
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

// scratch floats needed for njobs jobs (transposed phase table + per-job r and s*conj(r))
size_t subtract_scratch_floats(int njobs) { return (size_t)kSigLen * jobs_padded(njobs) + (size_t)njobs * kSubPerJob; }

void launch_subtract(float* dI, float* dQ, int samples, const SubJob* jobs, int njobs,
                     float* scratch, const DeviceTables& t, hipStream_t st) {
    if (njobs <= 0) return;
    float* phiT = scratch;
    float* perjob = scratch + (size_t)kSigLen * jobs_padded(njobs);
    hipLaunchKernelGGL(sub_phase_kernel, dim3((njobs + 63) / 64), dim3(64), 0, st, jobs, njobs, phiT);
    hipLaunchKernelGGL(sub_ref_kernel, dim3(kSigLen / 64, (njobs + 63) / 64), dim3(256), 0, st, dI, dQ, samples, jobs,
                       njobs, phiT, perjob);
    hipLaunchKernelGGL(sub_filter_kernel, dim3((kSigLen + kFirOut - 1) / kFirOut, njobs), dim3(kFirThreads), 0, st,
                       dI, dQ, samples, jobs, perjob, t.lpf, t.lpf_part);
}

void launch_normalise(float* dI, float* dQ, const int* n_valid, int nseg, int n_total, hipStream_t st) {
    if (nseg <= 0) return;
    hipLaunchKernelGGL(normalise_kernel, dim3(nseg), dim3(1024), 0, st, dI, dQ, n_valid, n_total);
}

}  // namespace wspr

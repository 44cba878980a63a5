// Can the fp32 matrix pipe form the decoder's ROUNDED PRODUCTS (D = 0 + a*b: one rounding, exactly v_mul_f32's) while the
// vector pipe does the separately rounded ADDS?  Per loop trip: one V_MFMA_F32_32X32X1_2B_F32 with C = 0 (2 048 products,
// 32 registers per lane) and sixteen v_pk_add_f32 that add the PREVIOUS trip's products to 32 accumulators.  Timed:
// the adds alone, the MFMAs alone, both -- with 1, 2 and 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/mfma_valu_overlap_probe.hip -o tools/mfma_valu_overlap_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v32f __attribute__((ext_vector_type(32)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int kMode>   // 1: adds only, 2: MFMA only, 3: both
__global__ __launch_bounds__(256) void probe(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a = in[tid & 1023], b = in[(tid + 7) & 1023];
    v32f acc, p0, p1;
    for (int r = 0; r < 32; ++r) { acc[r] = 0.0f; p0[r] = in[(tid + r) & 1023]; p1[r] = in[(tid + 2 * r) & 1023]; }
    const v32f zero = {};
    for (int it = 0; it < iters; it += 2) {
        if (kMode & 2) p1 = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, zero, 0, 0, 0);
        if (kMode & 1) {
#pragma unroll
            for (int r = 0; r < 32; r += 2) {
                v2f x = {acc[r], acc[r + 1]}, y = {p0[r], p0[r + 1]};
                x = x + y;
                acc[r] = x.x; acc[r + 1] = x.y;
            }
        }
        a += 1.0f;
        if (kMode & 2) p0 = __builtin_amdgcn_mfma_f32_32x32x1f32(b, a, zero, 0, 0, 0);
        if (kMode & 1) {
#pragma unroll
            for (int r = 0; r < 32; r += 2) {
                v2f x = {acc[r], acc[r + 1]}, y = {p1[r], p1[r + 1]};
                x = x + y;
                acc[r] = x.x; acc[r + 1] = x.y;
            }
        }
        b += 1.0f;
    }
    float s = 0.0f;
    for (int r = 0; r < 32; ++r) s += acc[r] + p0[r] + p1[r];
    out[tid] = s;
}

// wave specialisation: waves 0-3 of a 512-thread workgroup only multiply on the matrix pipe, waves 4-7 only add -- every
// SIMD holds one wave of each kind, no instruction of one depends on the other
__global__ __launch_bounds__(512) void probe_split(const float* __restrict__ in, float* __restrict__ out, int iters, int which) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a = in[tid & 1023], b = in[(tid + 7) & 1023];
    v32f acc, p0;
    for (int r = 0; r < 32; ++r) { acc[r] = 0.0f; p0[r] = in[(tid + r) & 1023]; }
    const v32f zero = {};
    const bool mfma_wave = threadIdx.x < 256;
    if (mfma_wave && (which & 2)) {
        for (int it = 0; it < iters; ++it) {
            p0 = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, zero, 0, 0, 0);
            a += p0[0];
        }
    } else if (!mfma_wave && (which & 1)) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 32; r += 2) {
                v2f x = {acc[r], acc[r + 1]}, y = {p0[r], p0[r + 1]};
                x = x + y;
                acc[r] = x.x; acc[r + 1] = x.y;
            }
        }
    }
    float s = a;
    for (int r = 0; r < 32; ++r) s += acc[r] + p0[r];
    out[tid] = s;
}

float run_split(int which, int iters, const float* in, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe_split, dim3(256), dim3(512), 0, 0, in, out, iters, which);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe_split, dim3(256), dim3(512), 0, 0, in, out, iters, which);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int kMode> float run(int wgs, int iters, const float* in, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<kMode>, dim3(wgs), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<kMode>, dim3(wgs), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 256 * 8);
    hipMemset(in, 0, 4096);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int wgs = 256 * wps;                          // a workgroup = 4 waves = one per SIMD of a CU
        const float t1 = run<1>(wgs, iters, in, out), t2 = run<2>(wgs, iters, in, out), t3 = run<3>(wgs, iters, in, out);
        // per trip and wave: 16 packed adds = 64 cycles of vector issue; one 32x32x1_2B = 64 cycles of the matrix pipe
        const double trips = (double)iters * wps;           // per SIMD
        printf("%d wave(s) per SIMD: adds alone %.3f ms (%.1f ns per trip and SIMD), MFMA alone %.3f ms (%.1f), both %.3f ms (%.1f); "
               "both / (adds + MFMA) = %.2f, both / max = %.2f\n", wps, t1, t1 * 1e6 / trips, t2, t2 * 1e6 / trips, t3, t3 * 1e6 / trips,
               t3 / (t1 + t2), t3 / (t1 > t2 ? t1 : t2));
    }
    const float s1 = run_split(1, iters, in, out), s2 = run_split(2, iters, in, out), s3 = run_split(3, iters, in, out);
    printf("wave-specialised (one adding wave and one multiplying wave per SIMD): adding waves alone %.3f ms, multiplying waves alone %.3f ms, "
           "both kinds at once %.3f ms; both / max = %.2f\n", s1, s2, s3, s3 / (s1 > s2 ? s1 : s2));
    return 0;
}

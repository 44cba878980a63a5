// Does the fp32 matrix pipe round the product before it adds (the reference's arithmetic: separately rounded multiply
// and add), or does it accumulate the exact product (a fused multiply-add)?  D = C + a*b through V_MFMA_F32_4X4X1_16B_F32
// (K = 1: one product per output) on operand sets where the two differ, next to v_mul_f32 + v_add_f32 and v_fma_f32.
// hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma_f32_rounding_probe.hip -o tools/mfma_f32_rounding_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe(const float* a, const float* b, const float* c, float* mfma, float* sep, float* fused, int n) {
    for (int t = 0; t < n; ++t) {
        v4f acc = {c[t], c[t], c[t], c[t]};
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[t], b[t], acc, 0, 0, 0);
        if (threadIdx.x == 0) {
            mfma[t] = acc[0];
            const float p = a[t] * b[t];
            sep[t] = p + c[t];
            fused[t] = __builtin_fmaf(a[t], b[t], c[t]);
        }
    }
}

int main() {
    const int n = 6;
    float a[n], b[n], c[n];
    const float e12 = ldexpf(1.0f, -12);
    a[0] = 1.0f + e12; b[0] = 1.0f + e12; c[0] = -(1.0f + ldexpf(1.0f, -11));        // fused: 2^-24, separate: 0
    a[1] = ldexpf(1.0f, -12); b[1] = ldexpf(1.0f, -12); c[1] = 1.0f;                   // tie: round to even -> 1
    a[2] = ldexpf(1.0f, -12); b[2] = ldexpf(1.0f, -12); c[2] = 1.0f + ldexpf(1.0f, -23);   // tie -> 1 + 2^-22
    a[3] = ldexpf(1.0f, -100); b[3] = ldexpf(1.0f, -40); c[3] = 0.0f;                  // denormal product
    a[4] = 3.0f; b[4] = ldexpf(1.0f, -149); c[4] = ldexpf(1.0f, -149);                 // denormal operands
    a[5] = 1.0f + ldexpf(1.0f, -23); b[5] = 1.0f - ldexpf(1.0f, -24); c[5] = -1.0f;    // product needs 48 bits
    float *da, *db, *dc, *dm, *ds, *df;
    hipMalloc(&da, sizeof a); hipMalloc(&db, sizeof a); hipMalloc(&dc, sizeof a);
    hipMalloc(&dm, sizeof a); hipMalloc(&ds, sizeof a); hipMalloc(&df, sizeof a);
    hipMemcpy(da, a, sizeof a, hipMemcpyHostToDevice); hipMemcpy(db, b, sizeof a, hipMemcpyHostToDevice);
    hipMemcpy(dc, c, sizeof a, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, dm, ds, df, n);
    float m[n], s[n], f[n];
    hipMemcpy(m, dm, sizeof a, hipMemcpyDeviceToHost); hipMemcpy(s, ds, sizeof a, hipMemcpyDeviceToHost);
    hipMemcpy(f, df, sizeof a, hipMemcpyDeviceToHost);
    int like_sep = 0, like_fused = 0;
    for (int t = 0; t < n; ++t) {
        unsigned um, us, uf;
        memcpy(&um, &m[t], 4); memcpy(&us, &s[t], 4); memcpy(&uf, &f[t], 4);
        printf("case %d: mfma %.9g (%08x)   mul+add %.9g (%08x)   fma %.9g (%08x)\n", t, m[t], um, s[t], us, f[t], uf);
        like_sep += um == us; like_fused += um == uf;
    }
    printf("V_MFMA_F32_4X4X1 equals mul+add in %d of %d cases, fma in %d of %d\n", like_sep, n, like_fused, n);
    return 0;
}

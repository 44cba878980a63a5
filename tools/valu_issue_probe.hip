// Micro-benchmark: what a wave's packed-fp32 instruction stream sustains at LOW occupancy (two waves per SIMD), by
// operand kind.  The decoder's matched-filter updates are  acc = (acc + x*c) + y*s  with x, y per lane and c, s
// wave-uniform: c, s can sit in VGPRs (one copy per lane) or in SGPRs (scalar operands of v_pk_mul_f32).
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_issue_probe.hip -o tools/valu_issue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: c,s in VGPRs (broadcast x via op_sel); 1: c,s wave-uniform in SGPRs; 2: plain a*m+c chains
__global__ __launch_bounds__(64) void probe(float* out, int iters, const float* seed) {
    extern __shared__ char pad[];              // occupancy limiter
    const int lane = threadIdx.x;
    v2f acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = (v2f){1.0f + lane * 1e-6f + k, 1.0f - k * 1e-3f};
    v2f c01 = {seed[0], seed[1]}, c23 = {seed[2], seed[3]}, s01 = {seed[4], seed[5]}, s23 = {seed[6], seed[7]};
    if (MODE == 0) { c01.x += lane * 1e-9f; }   // defeat uniformity analysis: keeps c,s in VGPRs
    float x = 1.0f + lane * 1e-7f, y = 1.0f - lane * 1e-7f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const v2f xx = {x, x}, yy = {y, y};
            if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { acc[4 * r + k] = acc[4 * r + k] * c01 + s01; acc[4 * r + k] = acc[4 * r + k] * c23 + s23; }
            } else {
                const v2f p0 = xx * c01, p1 = xx * c23, p2 = xx * s01, p3 = xx * s23;
                const v2f p4 = yy * s01, p5 = yy * s23, p6 = yy * c01, p7 = yy * c23;
                acc[4 * r + 0] = acc[4 * r + 0] + p0; acc[4 * r + 1] = acc[4 * r + 1] + p1;
                acc[4 * r + 2] = acc[4 * r + 2] - p2; acc[4 * r + 3] = acc[4 * r + 3] - p3;
                acc[4 * r + 0] = acc[4 * r + 0] + p4; acc[4 * r + 1] = acc[4 * r + 1] + p5;
                acc[4 * r + 2] = acc[4 * r + 2] + p6; acc[4 * r + 3] = acc[4 * r + 3] + p7;
            }
            x += 1e-7f; y -= 1e-7f;
        }
    }
    v2f t = acc[0];
#pragma unroll
    for (int k = 1; k < 12; ++k) t = t + acc[k];
    if (t.x == 123.456f) out[0] = t.y + pad[0];
}

template <int MODE> void run(const char* name, int waves_per_cu, float* out, const float* seed) {
    const int iters = 20000;
    const size_t lds = 160 * 1024 / waves_per_cu - 512;      // waves_per_cu single-wave workgroups fit a CU
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * waves_per_cu;
    probe<MODE><<<grid, 64, lds>>>(out, 100, seed);
    hipEventRecord(a);
    probe<MODE><<<grid, 64, lds>>>(out, iters, seed);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)grid * iters * 3 * 16;       // packed instructions per wave
    const double per_simd_cycle = instr / (1024.0 * ms * 1e-3 * 2.4e9);
    printf("%-34s %2d waves/CU: %.3f ms, %.2f packed instr per SIMD per 4 cycles (1.00 = full rate), %.1f TF/s\n", name,
           waves_per_cu, ms, per_simd_cycle * 4, instr * 64 * 2 / (ms * 1e-3) / 1e12);
}

int main() {
    float *out, *seed; hipMalloc(&out, 64); hipMalloc(&seed, 64);
    float h[8] = {1.0000001f, 0.9999999f, 1.0000002f, 0.9999998f, 1e-9f, -1e-9f, 2e-9f, -2e-9f};
    hipMemcpy(seed, h, 32, hipMemcpyHostToDevice);
    for (int w : {4, 8, 16, 32}) {
        run<0>("x*c, c in VGPRs (op_sel broadcast)", w, out, seed);
        run<1>("x*c, c in SGPRs", w, out, seed);
        run<2>("a*m+c chains (calibration)", w, out, seed);
    }
    return 0;
}

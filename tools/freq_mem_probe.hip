// Is the frequency scan limited by where its samples come from?  freq_scalar_kernel / freq_scalar_kernel on 2 048 candidates
// that (a) all live in 4 segments (samples stay in cache) and (b) live in 2 048 different segments (680 MB streamed).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/freq_mem_probe.hip -o tools/freq_mem_probe.bin
#include "../rtlsdr-wsprd_amd/csrc/kernels/k4_demod.hip"
#include <cstdio>
#include <vector>
using namespace wspr;
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    const int n = 2048, np = 45000;
    float *dI, *dQ, *tabs; FineState* dit; int* dl; float4 *pw;
    OK(hipMalloc(&dI, (size_t)n * kIqStride * 4)); OK(hipMalloc(&dQ, (size_t)n * kIqStride * 4));
    OK(hipMemset(dI, 0x3c, (size_t)n * kIqStride * 4)); OK(hipMemset(dQ, 0x3b, (size_t)n * kIqStride * 4));
    OK(hipMalloc(&tabs, (size_t)n * 5 * 2048 * 4)); OK(hipMalloc(&dit, n * sizeof(FineState))); OK(hipMalloc(&dl, n * 4));
    OK(hipMalloc(&pw, (size_t)n * 5 * kNSymD * 16));
    std::vector<int> list(n);
    for (int i = 0; i < n; ++i) list[i] = i;
    OK(hipMemcpy(dl, list.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    for (int spread = 0; spread < 2; ++spread) {
        std::vector<FineState> items(n);
        for (int i = 0; i < n; ++i) {
            FineState f{};
            f.seg = spread ? i : i % 4; f.freq = f.freq_coarse = -30.0f + 0.01f * i; f.shift = f.shift_coarse = 300 + i % 700; f.pad = i;
            items[i] = f;
        }
        OK(hipMemcpy(dit, items.data(), n * sizeof(FineState), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(phasor_freq_kernel, dim3(n), dim3(64), 0, 0, dit, dl, -2, 0.1f, tabs);
        for (int which = 0; which < 2; ++which)
            for (int rep = 0; rep < 3; ++rep) {
                OK(hipEventRecord(e0, 0));
                for (int k = 0; k < 10; ++k) {
                    if (which == 0)
                        hipLaunchKernelGGL(freq_scalar_kernel, dim3(n), dim3(kFqThreads), 0, 0, dI, dQ, np, dit, dl, tabs, pw, (const float4*)nullptr, 0, 8);
                    else
                        hipLaunchKernelGGL(freq_scalar_kernel, dim3(n), dim3(kFqThreads), 0, 0, dI, dQ, np, dit, dl, tabs, pw, (const float4*)nullptr, 0, 8);
                }
                OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
                float ms; OK(hipEventElapsedTime(&ms, e0, e1));
                if (rep == 2) printf("%s, candidates in %s: %.3f ms per 2048 (five hypotheses summed)\n", which ? "freq_scalar_kernel" : "freq_scalar_kernel  ",
                                     spread ? "2048 segments" : "4 segments   ", ms / 10);
            }
    }
    return 0;
}

// Unit check of demod_lagsys_kernel against demod_tile_kernel<8, true> (one (symbol, lag) per lane, samples in LDS: the
// kernel quick mode and WSPR_K4_LAG=tile use) -- same amplitudes, bit for bit -- on random data, incl. candidates
// that hang over either end of the record.  Includes the kernel file itself (its kernels live in an anonymous namespace).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I rtlsdr-wsprd_amd/csrc/kernels tools/lagsys_check.hip -o tools/lagsys_check.bin
#include "../rtlsdr-wsprd_amd/csrc/kernels/k4_demod.hip"
#include <cstdio>
#include <vector>
#include <cstring>
using namespace wspr;
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int nseg = 4, np = argc > 1 ? atoi(argv[1]) : 45000;
    const int shifts[] = {349, 5, -700, 128, 129, 3300, 2816, 1000};
    const int n = 8;
    std::vector<float> I((size_t)nseg * kIqStride), Q((size_t)nseg * kIqStride);
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) % 20001 - 10000) * 1e-4f; };
    for (auto& v : I) v = rnd();
    for (auto& v : Q) v = rnd();
    std::vector<FineState> items(n);
    std::vector<int> list(n);
    for (int i = 0; i < n; ++i) {
        FineState f{};
        f.seg = i % nseg; f.freq = f.freq_coarse = -30.0f + 7.3f * i; f.drift = 0; f.shift = f.shift_coarse = shifts[i]; f.pad = i;
        items[i] = f; list[i] = i;
    }
    float *dI, *dQ, *tabs; FineState* dit; int* dl; float4 *pa, *pb;
    const size_t npw = (size_t)n * 33 * kNSymD;
    OK(hipMalloc(&dI, I.size() * 4)); OK(hipMalloc(&dQ, Q.size() * 4)); OK(hipMalloc(&tabs, (size_t)n * 2048 * 4));
    OK(hipMalloc(&dit, n * sizeof(FineState))); OK(hipMalloc(&dl, n * 4)); OK(hipMalloc(&pa, npw * 16)); OK(hipMalloc(&pb, npw * 16));
    OK(hipMemcpy(dI, I.data(), I.size() * 4, hipMemcpyHostToDevice)); OK(hipMemcpy(dQ, Q.data(), Q.size() * 4, hipMemcpyHostToDevice));
    OK(hipMemcpy(dit, items.data(), n * sizeof(FineState), hipMemcpyHostToDevice)); OK(hipMemcpy(dl, list.data(), n * 4, hipMemcpyHostToDevice));
    OK(hipMemset(pa, 0xff, npw * 16)); OK(hipMemset(pb, 0xee, npw * 16));
    hipLaunchKernelGGL(phasor_table_kernel, dim3(1, n), dim3(64), 0, 0, dit, 0, tabs);
    const int span = kSps * kTileSymsShared + 8 * 32, pitch = (span + 7) / 8 + 1;
    const size_t tile_bytes = (size_t)pitch * 8 * sizeof(float2);
    const dim3 tile_threads(((kTileSymsShared * 33 + 63) / 64) * 64);
    hipLaunchKernelGGL((demod_tile_kernel<8, true>), dim3(kNSymD / kTileSymsShared, n), tile_threads, tile_bytes, 0, dI, dQ, np, dit,
                       dl, 0, 33, 0.0f, tabs, pa);
    hipLaunchKernelGGL(demod_lagsys_kernel, dim3(kSysWaves * ((n + 7) & ~7) + (n + 63) / 64), dim3(64), 0, 0, dI, dQ, np, dit, dl, n, tabs, pb);
    OK(hipDeviceSynchronize());
    std::vector<float> a(npw * 4), b(npw * 4);
    OK(hipMemcpy(a.data(), pa, npw * 16, hipMemcpyDeviceToHost)); OK(hipMemcpy(b.data(), pb, npw * 16, hipMemcpyDeviceToHost));
    long bad = 0;
    for (int it = 0; it < n; ++it) {
        long badi = 0; int shown = 0;
        for (int m = 0; m < 33; ++m) for (int sy = 0; sy < kNSymD; ++sy) {
            const size_t o = (((size_t)it * 33 + m) * kNSymD + sy) * 4;
            if (memcmp(&a[o], &b[o], 16)) {
                ++badi;
                if (shown++ < 6) printf("  item %d lag %d sym %d (u %d): tile %g %g %g %g  lagsys %g %g %g %g\n", it, m, sy, 32 * sy + m,
                                        a[o], a[o+1], a[o+2], a[o+3], b[o], b[o+1], b[o+2], b[o+3]);
            }
        }
        printf("item %d shift %d: %ld of %d differ\n", it, shifts[it], badi, 33 * kNSymD);
        bad += badi;
    }
    printf(bad ? "MISMATCH\n" : "lagsys == tile kernel bit for bit\n");
    {   // timing on many candidates (the same 8 items repeated)
        const int nb = 2048;
        std::vector<int> big(nb);
        for (int i = 0; i < nb; ++i) big[i] = (i % 2) ? 7 : 0;          // in-range candidates only
        int* dbl; OK(hipMalloc(&dbl, nb * 4)); OK(hipMemcpy(dbl, big.data(), nb * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            OK(hipEventRecord(e0, 0));
            for (int k = 0; k < 5; ++k)
                hipLaunchKernelGGL(demod_lagsys_kernel, dim3(kSysWaves * ((nb + 7) & ~7) + (nb + 63) / 64), dim3(64), 0, 0, dI, dQ, np, dit, dbl, nb, tabs, pb);
            OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
            float ms; OK(hipEventElapsedTime(&ms, e0, e1));
            printf("lagsys: %.3f ms per 2048 candidates (two cached segments)\n", ms / 5);
            OK(hipEventRecord(e0, 0));
            for (int k = 0; k < 5; ++k)
                hipLaunchKernelGGL((demod_tile_kernel<8, true>), dim3(kNSymD / kTileSymsShared, nb), tile_threads, tile_bytes, 0, dI, dQ,
                                   np, dit, dbl, 0, 33, 0.0f, tabs, pa);
            OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
            OK(hipEventElapsedTime(&ms, e0, e1));
            printf("tile kernel: %.3f ms per 2048 candidates\n", ms / 5);
        }
    }
    return bad != 0;
}

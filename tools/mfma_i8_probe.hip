// Register layout of v_mfma_i32_16x16x64_i8 (gfx950), found by experiment: one non-zero byte in A and in B at a time.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_i8_probe.hip -o tools/mfma_i8_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const int* a, const int* b, int* c) {
    const int l = threadIdx.x;
    v4i A = {a[4 * l], a[4 * l + 1], a[4 * l + 2], a[4 * l + 3]}, B = {b[4 * l], b[4 * l + 1], b[4 * l + 2], b[4 * l + 3]};
    v4i C = {0, 0, 0, 0};
    C = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, C, 0, 0, 0);
    for (int i = 0; i < 4; ++i) c[4 * l + i] = C[i];
}
int main() {
    int *da, *db, *dc;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dc, 1024);
    signed char ha[1024], hb[1024]; int hc[256];
    // 1. which (lane, byte) of A pairs with which (lane, byte) of B, and where the product lands
    int shown = 0;
    for (int la = 0; la < 64; la += 21) for (int ba = 0; ba < 16; ba += 5) {
        memset(ha, 0, 1024); ha[16 * la + ba] = 3;
        for (int i = 0; i < 1024; ++i) hb[i] = (signed char)(1 + (i % 16) + 16 * ((i / 16) / 16));   // value encodes (byte, lane/16)
        hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
        k<<<1, 64>>>(da, db, dc); hipMemcpy(hc, dc, 1024, hipMemcpyDeviceToHost);
        printf("A lane %2d byte %2d ->", la, ba);
        int cnt = 0;
        for (int i = 0; i < 256; ++i) if (hc[i]) { if (cnt++ < 4) printf(" C[lane %d reg %d]=%d", i / 4, i % 4, hc[i] / 3); }
        printf(" (%d outputs)\n", cnt);
        ++shown;
    }
    // 2. wrap-around of the accumulator
    memset(ha, 0, 1024); memset(hb, 0, 1024);
    for (int i = 0; i < 1024; ++i) { ha[i] = 127; hb[i] = 127; }
    hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dc); hipMemcpy(hc, dc, 1024, hipMemcpyDeviceToHost);
    printf("all 127 x 127 over K = 64: C = %d (expected %d)\n", hc[0], 127 * 127 * 64);
    return 0;
}

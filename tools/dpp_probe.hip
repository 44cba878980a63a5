// What v_mov_b32_dpp wave_shl:1 / wave_shr:1 / wave_rol:1 / row_shl:1 do on this GPU (lane i prints where its value came from).
// hipcc --offload-arch=gfx950 -O3 tools/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__global__ void k(int* o) {
    const int lane = threadIdx.x;
    o[lane] = __builtin_amdgcn_update_dpp(1000 + lane, lane, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
void run(const char* name) {
    int* d; int h[64];
    (void)hipMalloc(&d, 256);
    hipLaunchKernelGGL(k<CTRL>, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%s:", name);
    for (int i = 0; i < 64; ++i) printf(" %d", h[i]);
    printf("\n");
    (void)hipFree(d);
}
int main() {
    run<0x130>("wave_shl1");
    run<0x138>("wave_shr1");
    run<0x134>("wave_rol1");
    run<0x13c>("wave_ror1");
    run<0x101>("row_shl1");
    run<0x111>("row_shr1");
    return 0;
}

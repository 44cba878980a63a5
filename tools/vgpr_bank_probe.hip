// VGPR bank hypothesis: does v_pk_add_f32 A, A, P run faster when A and P sit in different VGPR banks (index mod 4)?
#include <hip/hip_runtime.h>
#include <cstdio>
#define STEP(A0,A1,A2,A3,A4,A5,A6,A7,P0,P1,P2,P3,P4,P5,P6,P7) \
    "v_pk_mul_f32 " P0 ", v[4:5], s[8:9] op_sel_hi:[0,1]\n\t" \
    "v_pk_mul_f32 " P1 ", v[4:5], s[10:11] op_sel_hi:[0,1]\n\t" \
    "v_pk_mul_f32 " P2 ", v[4:5], s[12:13] op_sel_hi:[0,1]\n\t" \
    "v_pk_mul_f32 " P3 ", v[4:5], s[14:15] op_sel_hi:[0,1]\n\t" \
    "v_pk_mul_f32 " P4 ", v[4:5], s[12:13] op_sel:[1,0] op_sel_hi:[1,1]\n\t" \
    "v_pk_mul_f32 " P5 ", v[4:5], s[14:15] op_sel:[1,0] op_sel_hi:[1,1]\n\t" \
    "v_pk_mul_f32 " P6 ", v[4:5], s[8:9] op_sel:[1,0] op_sel_hi:[1,1]\n\t" \
    "v_pk_mul_f32 " P7 ", v[4:5], s[10:11] op_sel:[1,0] op_sel_hi:[1,1]\n\t" \
    "v_pk_add_f32 " A0 ", " A0 ", " P0 "\n\t" \
    "v_pk_add_f32 " A1 ", " A1 ", " P1 "\n\t" \
    "v_pk_add_f32 " A2 ", " A2 ", " P2 "\n\t" \
    "v_pk_add_f32 " A3 ", " A3 ", " P3 "\n\t" \
    "v_pk_add_f32 " A0 ", " A0 ", " P4 "\n\t" \
    "v_pk_add_f32 " A1 ", " A1 ", " P5 "\n\t" \
    "v_pk_add_f32 " A2 ", " A2 ", " P6 "\n\t" \
    "v_pk_add_f32 " A3 ", " A3 ", " P7 "\n\t"
#define CLOB "v4","v5","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23", \
             "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71", \
             "s8","s9","s10","s11","s12","s13","s14","s15"
template <int MODE>
__global__ __launch_bounds__(64) void probe(float* out, int iters) {
    extern __shared__ char pad[];
    asm volatile("v_mov_b32 v4, 1.0\n\tv_mov_b32 v5, 1.0\n\t"
                 "v_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\t"
                 "v_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\tv_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\t"
                 "s_mov_b32 s8, 0x3a000000\n\ts_mov_b32 s9, 0x3a000000\n\ts_mov_b32 s10, 0x3a000000\n\ts_mov_b32 s11, 0x3a000000\n\t"
                 "s_mov_b32 s12, 0x3a000000\n\ts_mov_b32 s13, 0x3a000000\n\ts_mov_b32 s14, 0x3a000000\n\ts_mov_b32 s15, 0x3a000000\n\t" ::: CLOB);
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)       // accumulators AND products in pairs (4k, 4k+1): same banks
            asm volatile(STEP("v[8:9]","v[12:13]","v[16:17]","v[20:21]","","","","", "v[40:41]","v[44:45]","v[48:49]","v[52:53]","v[56:57]","v[60:61]","v[64:65]","v[68:69]")
                         STEP("v[8:9]","v[12:13]","v[16:17]","v[20:21]","","","","", "v[40:41]","v[44:45]","v[48:49]","v[52:53]","v[56:57]","v[60:61]","v[64:65]","v[68:69]")
                         STEP("v[8:9]","v[12:13]","v[16:17]","v[20:21]","","","","", "v[40:41]","v[44:45]","v[48:49]","v[52:53]","v[56:57]","v[60:61]","v[64:65]","v[68:69]") ::: CLOB);
        else if (MODE == 1)  // accumulators in (4k, 4k+1), products in (4k+2, 4k+3): different banks
            asm volatile(STEP("v[8:9]","v[12:13]","v[16:17]","v[20:21]","","","","", "v[42:43]","v[46:47]","v[50:51]","v[54:55]","v[58:59]","v[62:63]","v[66:67]","v[70:71]")
                         STEP("v[8:9]","v[12:13]","v[16:17]","v[20:21]","","","","", "v[42:43]","v[46:47]","v[50:51]","v[54:55]","v[58:59]","v[62:63]","v[66:67]","v[70:71]")
                         STEP("v[8:9]","v[12:13]","v[16:17]","v[20:21]","","","","", "v[42:43]","v[46:47]","v[50:51]","v[54:55]","v[58:59]","v[62:63]","v[66:67]","v[70:71]") ::: CLOB);
        else                 // mixed as a compiler might: accumulators alternate banks, products alternate banks
            asm volatile(STEP("v[8:9]","v[10:11]","v[12:13]","v[14:15]","","","","", "v[40:41]","v[42:43]","v[44:45]","v[46:47]","v[48:49]","v[50:51]","v[52:53]","v[54:55]")
                         STEP("v[8:9]","v[10:11]","v[12:13]","v[14:15]","","","","", "v[40:41]","v[42:43]","v[44:45]","v[46:47]","v[48:49]","v[50:51]","v[52:53]","v[54:55]")
                         STEP("v[8:9]","v[10:11]","v[12:13]","v[14:15]","","","","", "v[40:41]","v[42:43]","v[44:45]","v[46:47]","v[48:49]","v[50:51]","v[52:53]","v[54:55]") ::: CLOB);
    }
    float r;
    asm volatile("v_add_f32 %0, v8, v12" : "=v"(r) :: CLOB);
    if (r == 123.456f) out[0] = r + pad[0];
}
template <int MODE> void run(const char* name, int waves_per_cu, float* out) {
    const int iters = 20000;
    const size_t lds = 160 * 1024 / waves_per_cu - 512;
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * waves_per_cu;
    probe<MODE><<<grid, 64, lds>>>(out, 100);
    hipEventRecord(a);
    probe<MODE><<<grid, 64, lds>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)grid * iters * 3 * 16;
    printf("%-44s %2d waves/CU: %.3f ms, %.2f packed instr per SIMD per 4 cycles at 2.4 GHz\n", name, waves_per_cu, ms,
           instr / (1024.0 * ms * 1e-3 * 2.4e9) * 4);
}
int main() {
    float* out; hipMalloc(&out, 64);
    for (int w : {4, 8, 16}) {
        run<0>("acc and products on the same two banks", w, out);
        run<1>("acc on banks 0,1, products on banks 2,3", w, out);
        run<2>("alternating", w, out);
    }
    return 0;
}

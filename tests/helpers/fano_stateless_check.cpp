// Test helper: the storage-free Fano search (csrc/kernels/fano_stateless.h, the routine the GPU
// runs) compiled for the host, callable from ctypes with the reference fano() argument meaning.
#include <cstring>
#include "../../rtlsdr-wsprd_amd/csrc/kernels/fano_stateless.h"

extern "C" int fano_stateless_host(unsigned* metric, unsigned* cycles, unsigned* maxnp, unsigned char* data,
                                   const unsigned char* symbols, unsigned nbits, const int mettab[2][256], int delta,
                                   unsigned maxcycles) {
    wspr::Metric4 bm[128];
    for (unsigned k = 0; k < nbits; ++k) {
        const int a0 = mettab[0][symbols[2 * k]], a1 = mettab[1][symbols[2 * k]];
        const int b0 = mettab[0][symbols[2 * k + 1]], b1 = mettab[1][symbols[2 * k + 1]];
        bm[k].m[0] = (short)(a0 + b0); bm[k].m[1] = (short)(a0 + b1);
        bm[k].m[2] = (short)(a1 + b0); bm[k].m[3] = (short)(a1 + b1);
    }
    wspr::FanoResult r;
    wspr::fano_stateless([&](int pos) -> const wspr::Metric4& { return bm[pos]; }, nbits, delta, maxcycles, r);
    *metric = r.metric; *cycles = r.cycles; *maxnp = r.maxnp;
    std::memcpy(data, r.data, 10);
    return r.ret;
}

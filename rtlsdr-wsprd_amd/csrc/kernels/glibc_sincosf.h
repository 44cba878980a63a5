// Bit-exact single-precision sine/cosine for the device.
//
// The reference decoder takes every phasor seed and every sample of the
// subtraction reference signal from glibc's sinf()/cosf()
// (wsprd/wsprd.c:158-172, :347-348, :360, :512).  Decisions downstream are
// argmax/threshold tests on float32 sums, so the device must return the *same
// float* the host libm returns, not merely a value within an ulp.  ocml's
// sinf/cosf differ from glibc's in the last bit for a few percent of inputs.
//
// glibc >= 2.28 implements sinf/cosf (sysdeps/ieee754/flt-32/s_sincosf.h,
// s_sinf.c, s_cosf.c; algorithm published as ARM "optimized-routines" sincosf,
// Szabolcs Nagy / Wilco Dijkstra, MIT licence) as: range reduction to
// [-pi/4, pi/4] in double precision (one multiply-subtract for |x| < 120, a
// 192-bit 4/pi table for larger |x|), then a degree-7/8 double-precision
// polynomial, rounded once to float.  This header restates that published
// algorithm with explicitly written IEEE double operations, so the result is a pure
// function of the input bits on any IEEE machine.
//
// x86-64 glibc dispatches (ifunc) to a build of the same source compiled with
// -mfma on every CPU that has FMA3 -- i.e. on any host an MI355X sits in -- and
// in that build each `a + b*c` of the polynomial / reduction is one fused
// operation.  WSPR_SINCOS_FMA=1 (default) writes those fusions out with fma(),
// which reproduces that libm on ALL 2^32 inputs; =0 reproduces the non-FMA
// (SSE2) variant, which differs on 34 inputs out of 2^32 (tests/test_sincosf.py).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define WSPR_HD __host__ __device__ __forceinline__
#else
#define WSPR_HD static inline
#endif

#ifndef WSPR_SINCOS_FMA
#define WSPR_SINCOS_FMA 1
#endif

namespace wspr {

// a + b*c, fused or not according to the libm variant being reproduced
WSPR_HD double mad(double b, double c, double a) {
#if WSPR_SINCOS_FMA
    return __builtin_fma(b, c, a);
#else
    return a + b * c;
#endif
}

WSPR_HD uint32_t f32_bits(float f) {
    union { float f; uint32_t u; } v;
    v.f = f;
    return v.u;
}
WSPR_HD uint32_t abstop12(float f) { return (f32_bits(f) >> 20) & 0x7ff; }

// 4/pi to 192 bits, 8 new bits per entry (so any 32-bit window is addressable)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__
#endif
static const uint32_t kInvPio4[24] = {
    0xa2u,       0xa2f9u,     0xa2f983u,   0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u,
    0x6e4e4415u, 0x4e441529u, 0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u,
    0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u, 0x34ddc0dbu, 0xddc0db62u,
    0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};

// Evaluate sin (quadrant even) or cos (quadrant odd) of the reduced argument.
// neg selects the negated coefficient set used for quadrants 2 and 3.
WSPR_HD float sincos_poly(double x, double x2, bool neg, int n) {
    const double sg = neg ? -1.0 : 1.0;
    if ((n & 1) == 0) {
        const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
        double x3 = x * x2;
        double t  = mad(x2, s3, s2);
        double x7 = x3 * x2;
        double s  = mad(x3, s1, x);
        return (float)mad(x7, t, s);
    } else {
        const double c0 = sg * 0x1p0,                   c1 = sg * -0x1.ffffffd0c621cp-2,
                     c2 = sg * 0x1.55553e1068f19p-5,    c3 = sg * -0x1.6c087e89a359dp-10,
                     c4 = sg * 0x1.99343027bf8c3p-16;
        double x4 = x2 * x2;
        double t2 = mad(x2, c4, c3);
        double t1 = mad(x2, c1, c0);
        double x6 = x4 * x2;
        double c  = mad(x4, c2, t1);
        return (float)mad(x6, t2, c);
    }
}

// |x| < 120: n = round(x * 2/pi) via a 2^24-scaled product, r = x - n*pi/2
WSPR_HD double reduce_small(double x, int* np) {
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    double r = x * hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return mad(-(double)n, hpi, x);
}

// |x| >= 120: exact fixed-point product of the 24-bit mantissa with 96 bits of 4/pi
WSPR_HD double reduce_big(uint32_t xi, int* np) {
    const uint32_t* arr = &kInvPio4[(xi >> 26) & 15];
    int shift = (xi >> 23) & 7;
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    uint64_t res0 = (uint64_t)(uint32_t)(xi * arr[0]);
    uint64_t res1 = (uint64_t)xi * arr[4];
    uint64_t res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    uint64_t n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    double x = (double)(int64_t)res0;
    *np = (int)n;
    return x * 0x1.921FB54442D18p-62;
}

// which = 0: sine, 1: cosine
WSPR_HD float glibc_sincosf(float y, int which) {
    double x = y;
    int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {             // |y| < pi/4
        double x2 = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) return which ? 1.0f : y;
        return sincos_poly(x, x2, false, which);
    } else if (abstop12(y) < abstop12(120.0f)) {
        x = reduce_small(x, &n);
        double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;   // sign table {1,-1,-1,1}
        return sincos_poly(x * s, x * x, (n & 2) != 0, n ^ which);
    } else if (abstop12(y) < 0x7f8) {                          // finite
        uint32_t xi = f32_bits(y);
        int sign = (int)(xi >> 31);
        x = reduce_big(xi, &n);
        int q = (n + sign) & 3;
        double s = (q == 1 || q == 2) ? -1.0 : 1.0;
        return sincos_poly(x * s, x * x, ((n + sign) & 2) != 0, n ^ which);
    }
    return y - y;                                              // inf/nan -> nan
}

// sinf(y) and cosf(y) with ONE shared argument reduction (the two libm calls reduce the
// same argument identically, so sharing it changes nothing but the cost)
// Both polynomials of one reduced argument, signs left out: sincos_poly() is odd in x on its sine branch (x enters every
// term once) and its cosine branch only multiplies every coefficient by sg = +-1, and round-to-nearest is symmetric, so
//   sincos_poly(x * s, x2, neg, even n) == s * S   and   sincos_poly(x * s, x2, neg, odd n) == (neg ? -C : C)
// bit for bit, with S and C evaluated once for +x and sg = +1.  (A lane-dependent n made a wavefront walk both branches
// for the sine AND for the cosine: four polynomials per sample where two are enough.)
WSPR_HD void sincos_both(double x, double x2, float* S, float* Cc) {
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10,
                 c4 = 0x1.99343027bf8c3p-16;
    const double x3 = x * x2;
    const double t  = mad(x2, s3, s2);
    const double x7 = x3 * x2;
    const double s  = mad(x3, s1, x);
    *S = (float)mad(x7, t, s);
    const double x4 = x2 * x2;
    const double t2 = mad(x2, c4, c3);
    const double t1 = mad(x2, c1, c0);
    const double x6 = x4 * x2;
    const double c  = mad(x4, c2, t1);
    *Cc = (float)mad(x6, t2, c);
}

// quadrant n (and the reference's sign bookkeeping) applied to the two polynomial values
WSPR_HD void sincos_place(float S, float Cc, int n, int nsign, float* sn, float* cs) {
    const int q = nsign & 3;
    const float sp = (q == 1 || q == 2) ? -S : S;              // sign table {1,-1,-1,1} on the sine branch
    const float cp = (nsign & 2) ? -Cc : Cc;                   // sg on the cosine branch
    const bool odd = (n & 1) != 0;
    *sn = odd ? cp : sp;
    *cs = odd ? sp : cp;
}

WSPR_HD void glibc_sincosf_pair(float y, float* sn, float* cs) {
    double x = y;
    int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) { *sn = y; *cs = 1.0f; return; }
        sincos_both(x, x * x, sn, cs);
    } else if (abstop12(y) < abstop12(120.0f)) {
        x = reduce_small(x, &n);
        float S, Cc;
        sincos_both(x, x * x, &S, &Cc);
        sincos_place(S, Cc, n, n, sn, cs);
    } else if (abstop12(y) < 0x7f8) {
        const uint32_t xi = f32_bits(y);
        const int sign = (int)(xi >> 31);
        x = reduce_big(xi, &n);
        float S, Cc;
        sincos_both(x, x * x, &S, &Cc);
        sincos_place(S, Cc, n, n + sign, sn, cs);
    } else {
        *sn = *cs = y - y;
    }
}

WSPR_HD float glibc_sinf(float y) { return glibc_sincosf(y, 0); }
WSPR_HD float glibc_cosf(float y) { return glibc_sincosf(y, 1); }

}  // namespace wspr

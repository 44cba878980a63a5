// K1 -- windowed 512-point FFT bank + power spectrogram (gfx950 / wave64).
//
// Replaces reference wsprd/wsprd.c:509-553: 347 sine-windowed FFTs per segment,
// hop 128, squared magnitudes, fft-shifted.  Only the 417 bins any consumer reads
// (fft-shifted 48..464) are written.
//
// Mapping: one wavefront owns one FFT at a time, 8 complex points per lane.
//   * radix-2 decimation-in-frequency, 9 stages = 3 register passes of 3 stages;
//     between passes the 512 points are transposed through a wave-private,
//     bank-conflict-free LDS tile (row stride 72 / pad 9 complex words).
//   * consecutive FFTs overlap by 384 samples: a wave walks a run of consecutive
//     blocks and keeps the raw samples in a sliding register window (2 coalesced
//     256-B rows of I and of Q fetched per FFT); the four waves of a workgroup walk
//     adjacent runs, so the overlap between runs is served by the caches.
//   * output layout is BIN-MAJOR like the reference's ps[512][blocks] (wsprd.c:517):
//     ps[segment][bin 48..464][time, pitch 352].  The coarse sync reads whole bin rows
//     (11 per candidate) and the time average walks a bin's row in time order, so both
//     consumers stream contiguous memory.  A workgroup covers 4 x kRun consecutive
//     time blocks: the powers are collected in an LDS tile [417 bins][4 kRun] and
//     written as full 16-byte words, 64/128 contiguous bytes per bin row.
//   * window and twiddles live in registers for the whole run.
//   * a complex point is one VGPR pair and every butterfly is written for the packed fp32 pipe
//     (v_pk_add_f32 / v_pk_mul_f32 with op_sel broadcasts): five instructions per general
//     butterfly, two to four for the trivial twiddles of the last pass; the run loop is unrolled
//     by four so the sliding sample window rotates by renaming instead of register moves.
// Roofline: HBM-bound (algorithmic 938 796 B / segment / pass for ~9 MFLOP); at ~210 packed VALU
// instructions per FFT the vector pipe is the next limit, not far behind.
//
// The butterfly arithmetic (u+v, (u-v)*w with separately rounded products) is the
// same as the CPU oracle's orc_fft512(), so results are bit-identical to it.
#include "wspr_device.h"
#include <cstdlib>

#pragma clang fp contract(off)

namespace wspr {

namespace {
constexpr int kWavesPerWg    = 4;
constexpr int kTile          = 576;     // complex words of LDS per wave
// output tile: [417 bins][4 kRun + 1] floats (odd pitch: the 64 bins a wave writes per instruction
// fall on 32 banks, two lanes each -- the minimum for a 64-wide 4-byte access)

typedef float v2 __attribute__((ext_vector_type(2)));   // (re, im): one VGPR pair, v_pk_*_f32 operands

// General radix-2 DIF butterfly  u' = u + v,  v' = (u - v) * w  with the four products rounded
// separately (t1 = dr*wx, t2 = di*wy, t3 = dr*wy, t4 = di*wx; v' = (t1 - t2, t3 + t4)), exactly the
// oracle's orc_fft512().  Five packed instructions: wn = (-wy, wx) turns the mixed subtract/add
// into one packed add, since di*(-wy) = -(di*wy) and t1 + (-t2) = t1 - t2 bit for bit.
__device__ __forceinline__ void bfly(v2& u, v2& v, const v2 w, const v2 wn) {
    const v2 d = u - v;
    u = u + v;
    const v2 p = d.xx * w;
    const v2 q = d.yy * wn;
    v = p + q;
}

struct Tw {
    v2 w[7];
    // (-wy, wx): a swizzle and a sign of w, folded into the operand modifiers of the packed multiply
    __device__ __forceinline__ v2 wn(int k) const { return v2{-w[k].y, w[k].x}; }
};
__device__ __forceinline__ void set_tw(Tw& t, int k, const float2 f) { t.w[k] = v2{f.x, f.y}; }

// Four independent butterflies of one stage, issued operation by operation (all differences, all sums, all
// products, ...): a packed result is never read by the very next instruction, which would cost a wait state each.
__device__ __forceinline__ void bfly4(v2& u0, v2& v0, const v2 w0, const v2 n0, v2& u1, v2& v1, const v2 w1, const v2 n1,
                                      v2& u2, v2& v2_, const v2 w2, const v2 n2, v2& u3, v2& v3, const v2 w3, const v2 n3) {
    const v2 d0 = u0 - v0, d1 = u1 - v1, d2 = u2 - v2_, d3 = u3 - v3;
    u0 = u0 + v0; u1 = u1 + v1; u2 = u2 + v2_; u3 = u3 + v3;
    const v2 p0 = d0.xx * w0, p1 = d1.xx * w1, p2 = d2.xx * w2, p3 = d3.xx * w3;
    const v2 q0 = d0.yy * n0, q1 = d1.yy * n1, q2 = d2.yy * n2, q3 = d3.yy * n3;
    v0 = p0 + q0; v1 = p1 + q1; v2_ = p2 + q2; v3 = p3 + q3;
}

// three DIF stages on the 8 register-resident points; tw 0..3 stage a (pairs r,r+4),
// 4..5 stage b (pairs r,r+2), 6 stage c (pairs r,r+1)
__device__ __forceinline__ void pass3(v2 (&x)[8], const Tw& t) {
    bfly4(x[0], x[4], t.w[0], t.wn(0), x[1], x[5], t.w[1], t.wn(1), x[2], x[6], t.w[2], t.wn(2), x[3], x[7], t.w[3], t.wn(3));
    bfly4(x[0], x[2], t.w[4], t.wn(4), x[1], x[3], t.w[5], t.wn(5), x[4], x[6], t.w[4], t.wn(4), x[5], x[7], t.w[5], t.wn(5));
    bfly4(x[0], x[1], t.w[6], t.wn(6), x[2], x[3], t.w[6], t.wn(6), x[4], x[5], t.w[6], t.wn(6), x[6], x[7], t.w[6], t.wn(6));
}

// The last three stages only meet the twiddles 1, -i and (1-i)/sqrt2, (-1-i)/sqrt2.  Written
// out, the general butterfly reduces to the forms below with every surviving operation rounded
// exactly as in the general formula (x*1 = x, x*0 = +-0 adds exactly, -(a*b) = a*(-b),
// x - y = x + (-y)), so the bits are those of the general radix-2 butterfly the CPU oracle runs.
__device__ __forceinline__ void bfly_one(v2& u, v2& v) {            // w = 1
    const v2 d = u - v;
    u = u + v;
    v = d;
}
__device__ __forceinline__ void bfly_mi(v2& u, v2& v) {             // w = -i : v' = (di, -dr)
    const v2 d = u - v;
    u = u + v;
    v = v2{d.y, -d.x};
}
__device__ __forceinline__ void bfly_w8(v2& u, v2& v, float c) {    // w = (c, -c) : v' = (a + b, b - a)
    const v2 d = u - v;
    u = u + v;
    const v2 m = d * c;                                              // (a, b)
    v = m + v2{m.y, -m.x};
}
__device__ __forceinline__ void bfly_w83(v2& u, v2& v, float c) {   // w = (-c, -c) : v' = (b - a, -a - b)
    const v2 d = u - v;
    u = u + v;
    const v2 m = d * c;
    v = v2{m.y, -m.x} + (-m);
}
__device__ __forceinline__ void pass3_last(v2 (&x)[8], float c) {
    // the same butterflies, stage by stage with the independent operations of a stage side by side (see bfly4)
    {   // stage a: (0,4) w = 1, (1,5) w8, (2,6) -i, (3,7) w83
        const v2 d0 = x[0] - x[4], d1 = x[1] - x[5], d2 = x[2] - x[6], d3 = x[3] - x[7];
        x[0] = x[0] + x[4]; x[1] = x[1] + x[5]; x[2] = x[2] + x[6]; x[3] = x[3] + x[7];
        const v2 m1 = d1 * c, m3 = d3 * c;
        x[4] = d0;
        x[6] = v2{d2.y, -d2.x};
        x[5] = m1 + v2{m1.y, -m1.x};
        x[7] = v2{m3.y, -m3.x} + (-m3);
    }
    {   // stage b: (0,2) 1, (1,3) -i, (4,6) 1, (5,7) -i
        const v2 d0 = x[0] - x[2], d1 = x[1] - x[3], d2 = x[4] - x[6], d3 = x[5] - x[7];
        x[0] = x[0] + x[2]; x[1] = x[1] + x[3]; x[4] = x[4] + x[6]; x[5] = x[5] + x[7];
        x[2] = d0; x[6] = d2;
        x[3] = v2{d1.y, -d1.x};
        x[7] = v2{d3.y, -d3.x};
    }
    {   // stage c: all w = 1
        const v2 d0 = x[0] - x[1], d1 = x[2] - x[3], d2 = x[4] - x[5], d3 = x[6] - x[7];
        x[0] = x[0] + x[1]; x[2] = x[2] + x[3]; x[4] = x[4] + x[5]; x[6] = x[6] + x[7];
        x[1] = d0; x[3] = d1; x[5] = d2; x[7] = d3;
    }
}

__device__ __forceinline__ unsigned rev6(unsigned v) { return __brev(v) >> 26; }

// Rows of the output tile whose bin index has bit 5 set start one word late.  A ds_write_b32 is served in two groups
// of 32 lanes on 32 banks, and the first 32 lanes hold the EVEN lo (bit-reversed lane numbers): with the odd pitch
// alone their rows fall on 16 banks, two lanes each; rows 32 apart now differ by one bank and the 32 rows of a group
// cover all 32.  (A row has 16 values in 17 words, so the skew stays inside it.)
__device__ __forceinline__ int out_skew(int row) { return (row >> 5) & 1; }
struct OutCols { int base, at1, at6; };                      // see out_columns()
constexpr int kOutSpill = 64 + 32;                           // floats behind the output tile (lane + time offset)
__device__ constexpr int kOutRow[8] = {208, -48, 336, 80, 272, 16, 400, 144};
constexpr bool out_rows_share_a_skew() {
    for (int r = 0; r < 8; ++r) if (((kOutRow[r] % 64) + 64) % 64 != 16) return false;
    return true;
}
static_assert(out_rows_share_a_skew(), "every kOutRow[r] is 16 mod 64: one skew serves the eight rows of a lane");

// One FFT of the run: window, 3 passes, power into the workgroup's output tile.  `raw` is the sliding
// window of raw samples, raw[(base + r) & 7] = row r of this block; rotating `base` by 2 per block
// instead of moving registers needs the run loop unrolled by 4 (kBase is a compile-time constant).
template <int kBase, int kPitch, bool kBarrierBeforeWrite = false, bool kRefill = false>
__device__ __forceinline__ void one_fft(v2 (&raw)[8], const float (&win)[8], const Tw& twA, const Tw& twB, float w8,
                                        v2* __restrict__ X, int lane, int a, int c,
                                        const float* __restrict__ si, const float* __restrict__ sq, int t, bool more,
                                        float* __restrict__ otile, const OutCols ocol, int tl) {
    v2 x[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = raw[(kBase + r) & 7] * win[r];

    // slide the raw window by one hop (two 64-sample rows) while the FFT runs; the last FFT of a group (kRefill)
    // fetches the whole first window of the wave's next group instead: block t, rows 0..7 into raw[0..7]
    // (no branch around the loads -- it would make the compiler copy the whole window at every FFT: where nothing
    // follows, `more` is false and block 0 is fetched and never used)
    if (kRefill) {
        const int k0 = more ? kHop * t + lane : lane;
#pragma unroll
        for (int r = 0; r < 8; ++r) raw[r] = v2{si[k0 + 64 * r], sq[k0 + 64 * r]};
    } else {
        const int k = more ? kHop * (t + 1) + 384 + lane : lane;
        raw[(kBase + 0) & 7] = v2{si[k], sq[k]};
        raw[(kBase + 1) & 7] = v2{si[k + 64], sq[k + 64]};
    }

    pass3(x, twA);
#pragma unroll
    for (int r = 0; r < 8; ++r) X[72 * r + lane] = x[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = X[72 * a + 8 * r + c];
    __builtin_amdgcn_wave_barrier();

    pass3(x, twB);
#pragma unroll
    for (int r = 0; r < 8; ++r) X[72 * a + 9 * r + c] = x[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = X[9 * lane + r];
    __builtin_amdgcn_wave_barrier();

    pass3_last(x, w8);

    // x[r] now holds bin rev9(8*lane + r) = 64*rev3(r) + rev6(lane); see OutCols for its row in the output tile
    if (kBarrierBeforeWrite) __syncthreads();       // fused kernel: the previous group's tile has been consumed
    float* __restrict__ ob = otile + tl;                            // 64 lanes -> 64 different bins: 2 lanes per bank
    v2 e[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) e[r] = x[r] * x[r];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float pw = e[r].x + e[r].y;
        if (r == 1)      ob[ocol.at1] = pw;
        else if (r == 6) ob[ocol.at6] = pw;
        else ob[ocol.base + kOutRow[r] * kPitch] = pw;
    }
}

// Where register r's bin goes.  With lo = rev6(lane), x[r] is natural bin 64 rev3(r) + lo; fft-shifted (+256 mod 512)
// and counted from bin 48 (the first one kept) that is tile row lo + kOutRow[r].  Six of the eight rows exist for
// every lane; r = 1 (row lo - 48) only for lo >= 48, r = 6 (row lo + 400) only below row 417, i.e. lo < 17.  A lane
// without such a row writes its value to a word of its own behind the tile instead (kOutSpill floats): no branch.
template <int kPitch>
__device__ __forceinline__ OutCols out_columns(int lane) {
    static_assert(kPsBin0 == 48 && kPsBins == 417 && kFftSize == 512, "kOutRow is written for bins 48..464 of 512");
    const int lo = (int)rev6((unsigned)lane);
    const int spill = kPsBins * kPitch + lane;
    const int sk = out_skew(lo + 16);                // every kOutRow[r] is 16 mod 64: one skew for the lane's eight rows
    return OutCols{lo * kPitch + sk, lo >= 48 ? (lo + kOutRow[1]) * kPitch + sk : spill,
                   lo < 17 ? (lo + kOutRow[6]) * kPitch + sk : spill};
}

template <int kRun>
__global__ __launch_bounds__(256)
void fft_bank_kernel(const float* __restrict__ dI, const float* __restrict__ dQ,
                     const int* __restrict__ seg_list, int blocks, float* __restrict__ ps,
                     const float* __restrict__ window, const float2* __restrict__ twiddle) {
    extern __shared__ __attribute__((aligned(16))) char k1_smem[];
    v2* tile = reinterpret_cast<v2*>(k1_smem);                                     // 4 x 576 complex words
    float* otile = reinterpret_cast<float*>(k1_smem + kWavesPerWg * kTile * sizeof(v2));   // [417][kOutPitch]
    constexpr int kWgTimes = kWavesPerWg * kRun;
    constexpr int kOutPitch = kWgTimes + 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int seg = seg_list ? seg_list[blockIdx.y] : (int)blockIdx.y;
    const int t0 = blockIdx.x * kWgTimes;
    const int t_begin = t0 + wave * kRun;
    const int t_end = min(t_begin + kRun, blocks);

    const float* __restrict__ si = dI + (size_t)seg * kIqStride;
    const float* __restrict__ sq = dQ + (size_t)seg * kIqStride;
    v2* X = tile + wave * kTile;

    if (t_begin < t_end) {
        const int a = lane >> 3, c = lane & 7;
        float win[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) win[r] = window[64 * r + lane];
        // pass A: element n = 64 r + lane ; pass B: n = 64 a + 8 r + c ; pass C: n = 8 lane + r
        Tw twA, twB;
        set_tw(twA, 0, twiddle[lane]);     set_tw(twA, 1, twiddle[64 + lane]);
        set_tw(twA, 2, twiddle[128 + lane]); set_tw(twA, 3, twiddle[192 + lane]);
        set_tw(twA, 4, twiddle[2 * lane]); set_tw(twA, 5, twiddle[128 + 2 * lane]);
        set_tw(twA, 6, twiddle[4 * lane]);
        set_tw(twB, 0, twiddle[8 * c]);    set_tw(twB, 1, twiddle[64 + 8 * c]);
        set_tw(twB, 2, twiddle[128 + 8 * c]); set_tw(twB, 3, twiddle[192 + 8 * c]);
        set_tw(twB, 4, twiddle[16 * c]);   set_tw(twB, 5, twiddle[128 + 16 * c]);
        set_tw(twB, 6, twiddle[32 * c]);
        const float w8 = twiddle[64].x;                 // cos(pi/4) as float; twiddle[64] = (w8, -w8)
        const OutCols ocol = out_columns<kOutPitch>(lane);

        v2 raw[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int k = kHop * t_begin + 64 * r + lane;
            raw[r] = v2{si[k], sq[k]};
        }
        for (int t = t_begin; t < t_end; t += 4) {
            one_fft<0, kOutPitch>(raw, win, twA, twB, w8, X, lane, a, c, si, sq, t, t + 1 < t_end, otile, ocol, t - t0);
            if (t + 1 >= t_end) break;
            one_fft<2, kOutPitch>(raw, win, twA, twB, w8, X, lane, a, c, si, sq, t + 1, t + 2 < t_end, otile, ocol, t + 1 - t0);
            if (t + 2 >= t_end) break;
            one_fft<4, kOutPitch>(raw, win, twA, twB, w8, X, lane, a, c, si, sq, t + 2, t + 3 < t_end, otile, ocol, t + 2 - t0);
            if (t + 3 >= t_end) break;
            one_fft<6, kOutPitch>(raw, win, twA, twB, w8, X, lane, a, c, si, sq, t + 3, t + 4 < t_end, otile, ocol, t + 3 - t0);
        }
    }
    __syncthreads();

    // tile -> HBM: bin row b, times t0 .. t0 + kWgTimes - 1 as 16-byte words (8 or 4 per row: 128 or 64
    // contiguous bytes); streaming stores -- the spectrogram is far larger than the L2s and is next read
    // by other kernels.  Times beyond `blocks` (padding of the 352-float row) are written as zeros.
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int kParts = kWgTimes / 4;
    float* __restrict__ out = ps + (size_t)seg * kPsBins * kPsTPitch + t0;
    for (int e = threadIdx.x; e < kPsBins * kParts; e += 256) {
        const int b = e / kParts, part = e - b * kParts;
        const int tl = 4 * part;
        if (t0 + tl >= blocks) continue;
        const float* __restrict__ src = otile + b * kOutPitch + out_skew(b) + tl;
        f4 v;
        v.x = src[0];
        v.y = (t0 + tl + 1 < blocks) ? src[1] : 0.0f;
        v.z = (t0 + tl + 2 < blocks) ? src[2] : 0.0f;
        v.w = (t0 + tl + 3 < blocks) ? src[3] : 0.0f;
        __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out + (size_t)b * kPsTPitch + tl));
    }
}

// Two consecutive FFTs of a wave in lockstep (fused kernel).  A lone FFT stalls twice on its own transposes: eight LDS
// writes, eight reads, and nothing to do until they return.  Here the two take turns: while one's transposed points
// are on their way back from the LDS the other runs a register pass, so each read has ~75 packed instructions to
// land under.  One tile serves both: the LDS executes a wave's accesses in issue order, so B's writes cannot pass
// A's reads.  `S` is the wave's window of raw rows for one group of four FFTs: slot i holds row i (64 samples) of
// the group's first block, rows 10..13 replace rows 0..3 once the first pair has been windowed (kFirst = 0), and
// the next group's rows 0..9 are fetched once the second pair has (kFirst = 4).  obA/obB: the tile columns of the two
// powers; where the second FFT does not exist (the segment's last blocks) it runs on clamped rows and writes nothing.
template <int kFirst, int kPitch, bool kBarrierBeforeWrite>
__device__ __forceinline__ void fft_pair(v2 (&S)[10], const float (&win)[8], const Tw& twA, const Tw& twB, float w8,
                                         v2* __restrict__ X, int lane, int a, int c,
                                         const float* __restrict__ si, const float* __restrict__ sq, int k_next,
                                         float* __restrict__ obA, float* __restrict__ obB, bool has_b, const OutCols ocol) {
    v2 xa[8], xb[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        xa[r] = S[(kFirst + r) % 10] * win[r];
        xb[r] = S[(kFirst + 2 + r) % 10] * win[r];
    }
    // rows that follow: no branch, one clamp of the (wave-uniform) start.  Rows a real FFT needs never reach the
    // clamp (they end inside the segment's samples, and a row of the buffers is kIqStride floats: the clamp only
    // moves fetches that lie wholly behind the last block, whose values are never used)
    constexpr int kRows = kFirst == 0 ? 4 : 10;
    const float* __restrict__ pi = si + (min(k_next, kIqStride - 64 * kRows) + lane);
    const float* __restrict__ pq = sq + (min(k_next, kIqStride - 64 * kRows) + lane);
#pragma unroll
    for (int i = 0; i < kRows; ++i) S[i] = v2{pi[64 * i], pq[64 * i]};
    pass3(xa, twA);
#pragma unroll
    for (int r = 0; r < 8; ++r) X[72 * r + lane] = xa[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 8; ++r) xa[r] = X[72 * a + 8 * r + c];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);

    pass3(xb, twA);                                    // ... while A's points come back
#pragma unroll
    for (int r = 0; r < 8; ++r) X[72 * r + lane] = xb[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 8; ++r) xb[r] = X[72 * a + 8 * r + c];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);

    pass3(xa, twB);
#pragma unroll
    for (int r = 0; r < 8; ++r) X[72 * a + 9 * r + c] = xa[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 8; ++r) xa[r] = X[9 * lane + r];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);

    pass3(xb, twB);
#pragma unroll
    for (int r = 0; r < 8; ++r) X[72 * a + 9 * r + c] = xb[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 8; ++r) xb[r] = X[9 * lane + r];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);

    pass3_last(xa, w8);
    if (kBarrierBeforeWrite) __syncthreads();       // the previous group's tile has been consumed
    {
        v2 e[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) e[r] = xa[r] * xa[r];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float pw = e[r].x + e[r].y;
            if (r == 1)      obA[ocol.at1] = pw;
            else if (r == 6) obA[ocol.at6] = pw;
            else obA[ocol.base + kOutRow[r] * kPitch] = pw;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    pass3_last(xb, w8);
    if (has_b) {                                     // wave-uniform
        v2 e[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) e[r] = xb[r] * xb[r];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float pw = e[r].x + e[r].y;
            if (r == 1)      obB[ocol.at1] = pw;
            else if (r == 6) obB[ocol.at6] = pw;
            else obB[ocol.base + kOutRow[r] * kPitch] = pw;
        }
    }
}

// Fused form for large batches: ONE workgroup walks a whole segment, 16 time blocks (4 per wave) at a
// time, and folds every finished tile into the segment's time-averaged spectrum (wsprd.c:556-561:
// psavg[bin] = sum over the time blocks IN ORDER) before the next group starts -- thread = bin, so each
// running sum is the reference's serial one.  The spectrogram is then never read back for the average:
// the stage's HBM traffic drops from IQ + 2 x ps to IQ + ps.  A wave's four blocks are consecutive (sliding
// sample window inside the group); between groups the window is reloaded (the rows come from the caches).
template <int kRun>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void fft_bank_avg_kernel(const float* __restrict__ dI, const float* __restrict__ dQ,
                         const int* __restrict__ seg_list, int blocks, float* __restrict__ ps,
                         float* __restrict__ psavg, const float* __restrict__ window,
                         const float2* __restrict__ twiddle) {
    extern __shared__ __attribute__((aligned(16))) char k1_smem[];
    v2* tile = reinterpret_cast<v2*>(k1_smem);
    float* otile = reinterpret_cast<float*>(k1_smem + kWavesPerWg * kTile * sizeof(v2));
    constexpr int kWgTimes = kWavesPerWg * kRun;
    constexpr int kOutPitch = kWgTimes + 1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int seg = seg_list ? seg_list[blockIdx.x] : (int)blockIdx.x;
    const float* __restrict__ si = dI + (size_t)seg * kIqStride;
    const float* __restrict__ sq = dQ + (size_t)seg * kIqStride;
    v2* X = tile + wave * kTile;

    const int a = lane >> 3, c = lane & 7;
    float win[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) win[r] = window[64 * r + lane];
    Tw twA, twB;
    set_tw(twA, 0, twiddle[lane]);     set_tw(twA, 1, twiddle[64 + lane]);
    set_tw(twA, 2, twiddle[128 + lane]); set_tw(twA, 3, twiddle[192 + lane]);
    set_tw(twA, 4, twiddle[2 * lane]); set_tw(twA, 5, twiddle[128 + 2 * lane]);
    set_tw(twA, 6, twiddle[4 * lane]);
    set_tw(twB, 0, twiddle[8 * c]);    set_tw(twB, 1, twiddle[64 + 8 * c]);
    set_tw(twB, 2, twiddle[128 + 8 * c]); set_tw(twB, 3, twiddle[192 + 8 * c]);
    set_tw(twB, 4, twiddle[16 * c]);   set_tw(twB, 5, twiddle[128 + 16 * c]);
    set_tw(twB, 6, twiddle[32 * c]);
    const float w8 = twiddle[64].x;
    const OutCols ocol = out_columns<kOutPitch>(lane);

    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int kParts = kWgTimes / 4;
    float* __restrict__ out_seg = ps + (size_t)seg * kPsBins * kPsTPitch;
    const int b_lo = threadIdx.x, b_hi = threadIdx.x + 256;              // the bins this thread averages
    const int b_hi2 = min(b_hi, kPsBins - 1);
    v2 acc = {0.0f, 0.0f};

    // a wave's window of raw rows lives across groups (see fft_pair)
    v2 S[10];
    {
        const int k0 = min(kHop * (wave * kRun), kIqStride - 640) + lane;
#pragma unroll
        for (int i = 0; i < 10; ++i) S[i] = v2{si[k0 + 64 * i], sq[k0 + 64 * i]};
    }
    for (int t0 = 0; t0 < blocks; t0 += kWgTimes) {
        const int t_begin = t0 + wave * kRun;
        const int n_here = min(kRun, blocks - t_begin);                 // FFTs this wave has in the group (<= 0: none)
        if (n_here > 0) {
            static_assert(kRun == 4, "a group is two pairs of FFTs per wave");
            float* const ob = otile + (t_begin - t0);
            // the barrier that frees the tile sits inside the first pair, just before its first powers are written:
            // a wave that is done with the previous tile starts computing at once
            fft_pair<0, kOutPitch, true>(S, win, twA, twB, w8, X, lane, a, c, si, sq, kHop * t_begin + 640,
                                         ob, ob + 1, n_here >= 2, ocol);
            if (n_here >= 3)        // (a group with fewer blocks for this wave is the segment's last: nothing to refill)
                fft_pair<4, kOutPitch, false>(S, win, twA, twB, w8, X, lane, a, c, si, sq, kHop * (t_begin + kWgTimes),
                                              ob + 2, ob + 3, n_here >= 4, ocol);
        } else {
            __syncthreads();
        }
        __syncthreads();                                               // the group's tile is complete
        const int nt = min(kWgTimes, blocks - t0);
        // Only the rows the coarse sync can reach go to HBM: a candidate lies within +-110 Hz (wsprd.c:600-606), i.e.
        // if0 in [106, 406], and reads bins if0 - 6 .. if0 + 4 = 100 .. 410; the 106 rows outside were needed
        // for the time average alone, which this kernel forms itself.
        constexpr int kRowLo = 100 - kPsBin0, kRowHi = 410 - kPsBin0;
        static_assert(kParts == 4, "the store loop splits a row piece into four 16-byte words");
        if (nt == kWgTimes) {                                          // every group but the segment's last
#pragma unroll 1
            for (int e = threadIdx.x + kRowLo * kParts; e < (kRowHi + 1) * kParts; e += 256) {
                const int b = e >> 2, tl = 4 * (e & 3);
                const float* __restrict__ src = otile + b * kOutPitch + out_skew(b) + tl;
                const f4 v = {src[0], src[1], src[2], src[3]};
                __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out_seg + (size_t)b * kPsTPitch + t0 + tl));
            }
        } else {
            for (int e = threadIdx.x + kRowLo * kParts; e < (kRowHi + 1) * kParts; e += 256) {
                const int b = e >> 2, tl = 4 * (e & 3);
                if (tl >= nt) continue;
                const float* __restrict__ src = otile + b * kOutPitch + out_skew(b) + tl;
                f4 v;
                v.x = src[0];
                v.y = (tl + 1 < nt) ? src[1] : 0.0f;
                v.z = (tl + 2 < nt) ? src[2] : 0.0f;
                v.w = (tl + 3 < nt) ? src[3] : 0.0f;
                __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out_seg + (size_t)b * kPsTPitch + t0 + tl));
            }
        }
        {
            // the two running sums of a thread advance together, one packed add per time block (each half is the
            // reference's serial sum; threads without a second bin carry a copy of bin 416 along and drop it)
            const float* __restrict__ m0 = otile + b_lo * kOutPitch + out_skew(b_lo);
            const float* __restrict__ m1 = otile + b_hi2 * kOutPitch + out_skew(b_hi2);
            if (nt == kWgTimes) {
#pragma unroll
                for (int j = 0; j < kWgTimes; ++j) acc = acc + v2{m0[j], m1[j]};
            } else {
                for (int j = 0; j < nt; ++j) acc = acc + v2{m0[j], m1[j]};
            }
        }
    }
    psavg[(size_t)seg * kPsStride + b_lo] = acc.x;
    if (b_hi < kPsBins) psavg[(size_t)seg * kPsStride + b_hi] = acc.y;
}

}  // namespace

#ifdef WSPR_LAB   // calibration kernels: lab build only
// Calibration helper for the HBM PMC counters: a plain 4-byte-per-lane stream copy, the
// same access width as K1's loads/stores, over a known number of bytes.
namespace {
__global__ __launch_bounds__(256) void calib_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
}  // namespace
void launch_calib_copy(const float* src, float* dst, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(calib_copy_kernel, dim3(8192), dim3(256), 0, st, src, dst, n);
}
// Tuned stream copy, the ceiling of mixed read + write traffic: 16 bytes per lane, kCopyInFlight independent loads
// per lane issued before the first store, non-temporal in both directions (nothing is reused), a grid-stride loop
// over a resident grid (8 workgroups of 256 per CU: every wave slot filled once, no tail).  variant selects the
// cache policy so that the best one can be found on the box at hand: 0 = non-temporal loads and stores,
// 1 = default loads + non-temporal stores, 2 = default both (round 2's kernel, but with loads in flight).
namespace {
constexpr int kCopyInFlight = 4;
typedef float f4c __attribute__((ext_vector_type(4)));
template <int kVariant>
__global__ __launch_bounds__(256) void calib_copy16_kernel(const f4c* __restrict__ src, f4c* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (kCopyInFlight - 1) * stride < n4; i += kCopyInFlight * stride) {
        f4c v[kCopyInFlight];
#pragma unroll
        for (int u = 0; u < kCopyInFlight; ++u)
            v[u] = kVariant == 0 ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < kCopyInFlight; ++u) {
            if (kVariant <= 1) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}
}  // namespace
// write-only stream (non-temporal 16-byte stores of a constant): what the memory system takes in one direction
namespace {
__global__ __launch_bounds__(256) void calib_fill16_kernel(f4c* __restrict__ dst, size_t n4, float value) {
    const size_t stride = (size_t)gridDim.x * 256;
    const f4c v = {value, value, value, value};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) __builtin_nontemporal_store(v, dst + i);
}
}  // namespace
// write-only stream in the SPECTROGRAM's pattern: one workgroup per segment walks the 22 groups of 16 time blocks and
// stores, per group, a 64-byte piece of each of the 311 bin rows the fused K1 sends to HBM (rows 1408 bytes apart) --
// K1's stores without its arithmetic
namespace {
__global__ __launch_bounds__(256) void calib_fill_ps_kernel(float* __restrict__ ps, float value) {
    float* __restrict__ out = ps + (size_t)blockIdx.x * kPsBins * kPsTPitch;
    const f4c v = {value, value, value, value};
    for (int g = 0; g < 22; ++g)
        for (int e = threadIdx.x + 52 * 4; e < 363 * 4; e += 256) {
            const int b = e >> 2, part = e & 3;
            __builtin_nontemporal_store(v, reinterpret_cast<f4c*>(out + (size_t)b * kPsTPitch + 16 * g + 4 * part));
        }
}
}  // namespace
void launch_calib_copy16(const float* src, float* dst, size_t n, hipStream_t st, int variant) {
    if (variant == 4) {                                      // n floats of room: floor(n / (417 * 352)) segments
        const int nseg = (int)(n / ((size_t)kPsBins * kPsTPitch));
        if (nseg > 0) hipLaunchKernelGGL(calib_fill_ps_kernel, dim3(nseg), dim3(256), 0, st, dst, 1.0f);
        return;
    }
    if (variant == 3) {
        hipLaunchKernelGGL(calib_fill16_kernel, dim3(256 * 8), dim3(256), 0, st, reinterpret_cast<f4c*>(dst), n / 4, 1.0f);
        return;
    }
    const f4c* s4 = reinterpret_cast<const f4c*>(src);
    f4c* d4 = reinterpret_cast<f4c*>(dst);
    const dim3 grid(256 * 8), block(256);
    if (variant == 0) hipLaunchKernelGGL(calib_copy16_kernel<0>, grid, block, 0, st, s4, d4, n / 4);
    else if (variant == 1) hipLaunchKernelGGL(calib_copy16_kernel<1>, grid, block, 0, st, s4, d4, n / 4);
    else hipLaunchKernelGGL(calib_copy16_kernel<2>, grid, block, 0, st, s4, d4, n / 4);
}
#endif  // WSPR_LAB

void launch_fft_bank(const float* dI, const float* dQ, const int* seg_list, int nseg_active,
                     int samples, float* ps, const DeviceTables& t, hipStream_t st) {
    const int blocks = 4 * (samples / kFftSize) - 1;
    if (blocks <= 0 || nseg_active <= 0) return;
    // four consecutive FFTs per wave: a workgroup covers 16 time blocks (64-byte row segments, 47 KB of LDS, three
    // workgroups per CU).  Eight per wave (128-byte segments, 73 KB, two per CU) measured slower: 282 vs 259 us per
    // 1024 segments.
    constexpr int R = 4;
    constexpr int per_wg = R * kWavesPerWg;
    const size_t lds = kWavesPerWg * kTile * sizeof(v2) + ((size_t)kPsBins * (per_wg + 1) + kOutSpill) * sizeof(float);
    static std::atomic<unsigned> opted{0};
    lds_opt_in(reinterpret_cast<const void*>(&fft_bank_kernel<R>), lds, opted);
    dim3 grid((blocks + per_wg - 1) / per_wg, nseg_active);
    hipLaunchKernelGGL(fft_bank_kernel<R>, grid, dim3(256), lds, st, dI, dQ, seg_list, blocks, ps, t.window, t.twiddle);
}

// K1 + K2a in one kernel (see fft_bank_avg_kernel); one workgroup per segment, so only for batches that
// fill the GPU on their own.
void launch_fft_bank_avg(const float* dI, const float* dQ, const int* seg_list, int nseg_active,
                         int samples, float* ps, float* psavg, const DeviceTables& t, hipStream_t st) {
    const int blocks = 4 * (samples / kFftSize) - 1;
    if (blocks <= 0 || nseg_active <= 0) return;
    constexpr int R = 4;
    const size_t lds = kWavesPerWg * kTile * sizeof(v2) + ((size_t)kPsBins * (R * kWavesPerWg + 1) + kOutSpill) * sizeof(float);
    hipLaunchKernelGGL(fft_bank_avg_kernel<R>, dim3(nseg_active), dim3(256), lds, st, dI, dQ, seg_list, blocks, ps,
                       psavg, t.window, t.twiddle);
}

}  // namespace wspr

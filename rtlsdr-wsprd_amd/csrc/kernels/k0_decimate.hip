// K0 -- receiver front end: fs/4 mixer + 2-stage CIC (R = 6401) + 33-tap FIR,
// 2.4 Msps unsigned-8-bit IQ -> ~375 sps float32 IQ.
//
// Replaces reference rtlsdr_callback(), rtlsdr_wsprd.c:126-244.  The reference keeps the
// decimator state in `static` variables, i.e. it streams across callbacks and segments; here the
// state is an explicit DecimState per receiver (null = zero state = the receiver at start-up), so
// the same kernels serve a whole 2-minute buffer or a stream cut into arbitrary chunks.
// Bound: HBM bandwidth -- one pass over 576 000 000 B per 2-minute segment, ~0.02 op/B.
//
// The streaming recurrences are restated as exact integer block sums.  With
// x_r the mixed sample, R = 6401 and block b = samples [bR, (b+1)R):
//     S_b = sum_r x_r            W_b = sum_r (R - off_r) x_r      (off_r = r - bR)
//     I1(b) = I1(b-1) + S_b      I2(b) = I2(b-1) + R*I1(b-1) + W_b    (mod 2^32)
// are the two integrators sampled at the decimation instants, bit-for-bit,
// because int32 wrap-around arithmetic is associative.  The mixer is the
// reference's in-place int8 trick: multiply sample n by (1, j, -1, -j)[n & 3]
// with int8 negation (-(-128) stays -128).
//   pass A (HBM-bound): block sums, one workgroup per pair of blocks, aligned 16-byte loads;
//   pass B (tiny)     : the two integrators as parallel prefix sums over the 45 000 block sums;
//   pass C (tiny)     : both combs + 33-tap compensation FIR, taps summed in reference order.
// With a carried state, block 0 is the remainder of the block that was open when the previous
// chunk ended (phase p samples already integrated: its weights simply start at R - p), the
// integrators start from the carried values, the combs/FIR reach back into the last 36 carried
// integrator samples, and a trailing partial block yields the state to carry on:
//     I1_end = I1(last) + S_tail,   I2_end = I2(tail as if complete) - (R - n_tail) * I1_end.
// Requirement: every segment row starts 16-byte aligned (bytes_per_seg % 16 == 0) and the buffer
// is readable up to the next multiple of 16 bytes.
#include "wspr_device.h"

#pragma clang fp contract(off)

namespace wspr {
namespace {

constexpr int kR = 6401;                 // decimation ratio (DOWNSAMPLING + 1)

#define WSPR_FIR_TAPS {      /* rtlsdr_wsprd.c:142-152 */                                                     \
    -0.0027772683f, -0.0005058826f, 0.0049745750f, -0.0034059318f, -0.0077557814f, 0.0139375423f,             \
    0.0039896935f,  -0.0299394142f, 0.0162250643f, 0.0405130860f,  -0.0580746013f, -0.0272104968f,            \
    0.1183705475f,  -0.0306029022f, -0.2011241667f, 0.1615898423f, 0.5000000000f,  0.1615898423f,             \
    -0.2011241667f, -0.0306029022f, 0.1183705475f, -0.0272104968f, -0.0580746013f, 0.0405130860f,             \
    0.0162250643f,  -0.0299394142f, 0.0039896935f, 0.0139375423f,  -0.0077557814f, -0.0034059318f,            \
    0.0049745750f,  -0.0005058826f, -0.0027772683f}
__constant__ float kFirTaps[33] = WSPR_FIR_TAPS;
const float kFirTapsHost[33] = WSPR_FIR_TAPS;

__device__ __forceinline__ int s8(unsigned b) { return (int)(b & 0xffu) - 128; }        // (int8)(b ^ 0x80)
__device__ __forceinline__ int neg8(int v) { return (v == -128) ? -128 : -v; }          // int8 negate

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sums[seg][block][4] = {S_I, S_Q, W_I, W_Q}
// One workgroup per pair of blocks.  The stream is read as aligned 16-byte vectors (8 complex
// samples); a vector starts at a sample index that is a multiple of 8, so the mixer phase of its
// k-th sample is the compile-time constant k & 3.  Samples of a vector that belong to a
// neighbouring pair (the pair boundaries are not 16-byte aligned) are masked out.
__global__ __launch_bounds__(256)
void cic_block_sums_kernel(const uint8_t* __restrict__ raw, size_t bytes_per_seg, int nblocks,
                           int32_t* __restrict__ sums, const DecimState* __restrict__ states) {
    __shared__ unsigned red[4][8];
    const int seg = blockIdx.y, pair = blockIdx.x, tid = threadIdx.x;
    const long seg_samples = (long)(bytes_per_seg / 2);
    const int phase = states ? (int)states[seg].phase : 0;
    // blocks of this segment: with a state, every block the chunk touches (the last may be partial)
    const int nb = states ? (int)((phase + seg_samples + kR - 1) / kR) : nblocks;
    const int blkA = 2 * pair;
    if (blkA >= nb) return;
    const bool haveB = (blkA + 1) < nb;
    const long first = (long)pair * 2 * kR - phase;               // first sample of this pair (< 0: carried over)
    const long last = min(first + (haveB ? 2 * kR : kR), seg_samples);   // one past its last sample
    const uint4* __restrict__ vec = reinterpret_cast<const uint4*>(raw + (size_t)seg * bytes_per_seg);
    const long v_lo = (first < 0 ? 0 : first) >> 3, v_hi = (last + 7) >> 3;   // vectors touching [first, last)
    const long v_max = (seg_samples + 7) >> 3;

    // all sums are modulo 2^32 (the reference's int32 integrators wrap): unsigned math
    unsigned acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};                   // A: SI SQ WI WQ, B: SI SQ WI WQ
    auto process = [&](long v, const uint4 q) {
        const long n0 = 8 * v;
        const int idx0 = (int)(n0 - first);
        const bool fullA = (idx0 >= 0) && (idx0 + 7 < kR) && (n0 + 7 < last);
        const bool fullB = haveB && (idx0 >= kR) && (n0 + 7 < last);
        // a raw byte 0x00 is -128 after the sign flip and its int8 negation wraps: such
        // vectors (clipping only) and the few that straddle a block edge take the per-sample path
        const unsigned z = ((q.x - 0x01010101u) & ~q.x) | ((q.y - 0x01010101u) & ~q.y) |
                           ((q.z - 0x01010101u) & ~q.z) | ((q.w - 0x01010101u) & ~q.w);
        if ((fullA || fullB) && !(z & 0x80808080u)) {
            // Fast path, 8 samples of one block.  Sample k of the vector has mixer phase k & 3, so a
            // dword [I_k Q_k I_k+1 Q_k+1] (k even) contributes  xi = (+I_k, -Q_k+1), xq = (+Q_k, +I_k+1)
            // for k = 0, 4 and  xi = (-I_k, +Q_k+1), xq = (-Q_k, -I_k+1)  for k = 2, 6:
            // signed 4 x int8 dot products with constant weights (bytes little-endian in the constant).
            const int w0 = (int)(q.x ^ 0x80808080u), w1 = (int)(q.y ^ 0x80808080u);
            const int w2 = (int)(q.z ^ 0x80808080u), w3 = (int)(q.w ^ 0x80808080u);
            constexpr int kIe = (int)0xff000001u;   // (+1, 0, 0, -1): xi of phases (0,1)
            constexpr int kIo = (int)0x010000ffu;   // (-1, 0, 0, +1): xi of phases (2,3)
            constexpr int kQe = (int)0x00010100u;   // ( 0,+1,+1, 0): xq of phases (0,1)
            constexpr int kQo = (int)0x00ffff00u;   // ( 0,-1,-1, 0): xq of phases (2,3)
            int tI = __builtin_amdgcn_sdot4(w0, kIe, 0, false);
            tI = __builtin_amdgcn_sdot4(w1, kIo, tI, false);
            tI = __builtin_amdgcn_sdot4(w2, kIe, tI, false);
            tI = __builtin_amdgcn_sdot4(w3, kIo, tI, false);
            int tQ = __builtin_amdgcn_sdot4(w0, kQe, 0, false);
            tQ = __builtin_amdgcn_sdot4(w1, kQo, tQ, false);
            tQ = __builtin_amdgcn_sdot4(w2, kQe, tQ, false);
            tQ = __builtin_amdgcn_sdot4(w3, kQo, tQ, false);
            // sum_k k * x_k: weights (0,-1 | -2,+3 | +4,-5 | -6,+7) for xi, (0,+1 | -2,-3 | +4,+5 | -6,-7) for xq
            int uI = __builtin_amdgcn_sdot4(w0, (int)0xff000000u, 0, false);
            uI = __builtin_amdgcn_sdot4(w1, (int)0x030000feu, uI, false);
            uI = __builtin_amdgcn_sdot4(w2, (int)0xfb000004u, uI, false);
            uI = __builtin_amdgcn_sdot4(w3, (int)0x070000fau, uI, false);
            int uQ = __builtin_amdgcn_sdot4(w0, (int)0x00010000u, 0, false);
            uQ = __builtin_amdgcn_sdot4(w1, (int)0x00fdfe00u, uQ, false);
            uQ = __builtin_amdgcn_sdot4(w2, (int)0x00050400u, uQ, false);
            uQ = __builtin_amdgcn_sdot4(w3, (int)0x00f9fa00u, uQ, false);
            // W += sum_k (R - off0 - k) x_k = (R - off0) * t - u
            const int wbase = kR - (fullB ? idx0 - kR : idx0);
            const unsigned wI = (unsigned)(__mul24(wbase, tI) - uI), wQ = (unsigned)(__mul24(wbase, tQ) - uQ);
            const unsigned mA = fullA ? 1u : 0u, mB = fullA ? 0u : 1u;
            acc[0] += mA * (unsigned)tI;  acc[1] += mA * (unsigned)tQ;  acc[2] += mA * wI;  acc[3] += mA * wQ;
            acc[4] += mB * (unsigned)tI;  acc[5] += mB * (unsigned)tQ;  acc[6] += mB * wI;  acc[7] += mB * wQ;
            return;
        }
        const unsigned wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long n = n0 + k;                                // absolute sample index in the segment
            const unsigned wv = wds[k >> 1] >> (16 * (k & 1));
            const int a = s8(wv), b = s8(wv >> 8);
            int xi, xq;
            switch (k & 3) {                                      // (1, j, -1, -j)[n & 3], n = 8v + k
                case 0:  xi = a;        xq = b;        break;
                case 1:  xi = neg8(b);  xq = a;        break;
                case 2:  xi = neg8(a);  xq = neg8(b);  break;
                default: xi = b;        xq = neg8(a);  break;
            }
            const int idx = (int)(n - first);
            const bool inA = (n >= first) && (idx < kR) && (n < last);
            const bool inB = (idx >= kR) && (n < last);
            const unsigned wgt = (unsigned)(kR - (inB ? idx - kR : idx));
            const unsigned mA = inA ? 1u : 0u, mB = inB ? 1u : 0u;
            const unsigned ui = (unsigned)xi, uq = (unsigned)xq;
            acc[0] += mA * ui;        acc[1] += mA * uq;
            acc[2] += mA * wgt * ui;  acc[3] += mA * wgt * uq;
            acc[4] += mB * ui;        acc[5] += mB * uq;
            acc[6] += mB * wgt * ui;  acc[7] += mB * wgt * uq;
        }
    };
    // a pair of blocks is ~1600 vectors, 6-7 per thread: four loads are issued before the first is used, so
    // that a workgroup keeps 16 KB in flight instead of 4 (the stream needs ~10 MB outstanding chip-wide)
    const long v_end = min(v_hi, v_max);
    for (long v0 = v_lo + tid; v0 < v_end; v0 += 4 * 256) {
        uint4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long v = v0 + 256 * u;
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const u4 t = __builtin_nontemporal_load(reinterpret_cast<const u4*>(vec) + ((v < v_end) ? v : v_lo));   // rows are allocated in whole vectors
            q[u] = make_uint4(t.x, t.y, t.z, t.w);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long v = v0 + 256 * u;
            if (v < v_end) process(v, q[u]);
        }
    }
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const unsigned s = wave_sum(acc[q]);
        if (lane == 0) red[wave][q] = s;
    }
    __syncthreads();
    if (tid < 8) {
        const unsigned s = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        const int blk = blkA + (tid >> 2);
        if (blk < nb) sums[((size_t)seg * nblocks + blk) * 4 + (tid & 3)] = (int32_t)s;
    }
}

// ---- whole segments (no carried state): one WAVE per block ------------------------------------------------
// The general kernel above routes every vector through block-edge tests (64-bit), A/B accumulator masks and
// an inline per-sample path that a whole wave executes whenever one of its lanes holds an edge vector: ~55
// instructions per 16 bytes plus ~250 per edge, 0.53 of HBM peak where a kernel that only reads the same rows
// in the same pattern reaches 0.88 (wspr_calib_read).  Without a carried state the geometry is fixed, so:
//   * a wave owns one block: 799 or 800 interior vectors + at most two edge vectors, four accumulators, no
//     routing; the reduction is one wave sum, no LDS, no barrier;
//   * three rounds of 4 x 64 interior vectors run with no bounds test at all (768 <= 799);
//   * ONE masked round takes the remaining <= 32 interior vectors and the two edge vectors (lanes 62, 63):
//     bytes outside the block are cleared after the sign flip (x = 0 contributes nothing), the weight base
//     R + lo - 8v covers a vector that starts before the block;
//   * raw bytes 0x00 (-128, whose int8 negation wraps) are only detected in the hot loop
//     (two instructions per dword); a wave that saw one redoes its block sample by sample afterwards.
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <bool kMasked>
__device__ __forceinline__ void block_sum_vector(const u4 q, const int wbase, const int k_lo, const int k_hi,
                                                 unsigned (&acc)[4], unsigned (&zacc)[4]) {
    // has-zero-byte test, one subtract and one three-input bit operation per dword
    zacc[0] |= (q.x - 0x01010101u) & ~q.x;
    zacc[1] |= (q.y - 0x01010101u) & ~q.y;
    zacc[2] |= (q.z - 0x01010101u) & ~q.z;
    zacc[3] |= (q.w - 0x01010101u) & ~q.w;
    int w[4] = {(int)(q.x ^ 0x80808080u), (int)(q.y ^ 0x80808080u), (int)(q.z ^ 0x80808080u), (int)(q.w ^ 0x80808080u)};
    if (kMasked) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const unsigned m0 = (2 * d >= k_lo && 2 * d < k_hi) ? 0x0000ffffu : 0u;
            const unsigned m1 = (2 * d + 1 >= k_lo && 2 * d + 1 < k_hi) ? 0xffff0000u : 0u;
            w[d] &= (int)(m0 | m1);
        }
    }
    // sample k of the vector has mixer phase k & 3 (a vector starts at a multiple of 8): xi, xq as signed
    // 4 x int8 dot products with constant weights, see cic_block_sums_kernel; nu = -sum_k k x_k
    int tI = __builtin_amdgcn_sdot4(w[0], (int)0xff000001u, 0, false);
    tI = __builtin_amdgcn_sdot4(w[1], (int)0x010000ffu, tI, false);
    tI = __builtin_amdgcn_sdot4(w[2], (int)0xff000001u, tI, false);
    tI = __builtin_amdgcn_sdot4(w[3], (int)0x010000ffu, tI, false);
    int tQ = __builtin_amdgcn_sdot4(w[0], (int)0x00010100u, 0, false);
    tQ = __builtin_amdgcn_sdot4(w[1], (int)0x00ffff00u, tQ, false);
    tQ = __builtin_amdgcn_sdot4(w[2], (int)0x00010100u, tQ, false);
    tQ = __builtin_amdgcn_sdot4(w[3], (int)0x00ffff00u, tQ, false);
    // W += sum_k (R - (8v + k - lo)) x_k = wbase * t - sum_k k x_k; the second term is accumulated by the dot
    // products themselves (wrap-around int32 sums are associative)
    int nuI = __builtin_amdgcn_sdot4(w[0], (int)0x01000000u, (int)acc[2], false);
    nuI = __builtin_amdgcn_sdot4(w[1], (int)0xfd000002u, nuI, false);
    nuI = __builtin_amdgcn_sdot4(w[2], (int)0x050000fcu, nuI, false);
    nuI = __builtin_amdgcn_sdot4(w[3], (int)0xf9000006u, nuI, false);
    int nuQ = __builtin_amdgcn_sdot4(w[0], (int)0x00ff0000u, (int)acc[3], false);
    nuQ = __builtin_amdgcn_sdot4(w[1], (int)0x00030200u, nuQ, false);
    nuQ = __builtin_amdgcn_sdot4(w[2], (int)0x00fbfc00u, nuQ, false);
    nuQ = __builtin_amdgcn_sdot4(w[3], (int)0x00070600u, nuQ, false);
    acc[0] += (unsigned)tI;
    acc[1] += (unsigned)tQ;
    acc[2] = (unsigned)nuI + (unsigned)__mul24(wbase, tI);
    acc[3] = (unsigned)nuQ + (unsigned)__mul24(wbase, tQ);
}

// per-sample path of one vector, samples n in [lo, hi) only (the reference's arithmetic, sample by sample)
__device__ __noinline__ void block_sum_vector_exact(const u4 q, const int n0, const int lo, const int hi,
                                                    unsigned (&acc)[4]) {
    const unsigned wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int n = n0 + k;
        const unsigned wv = wds[k >> 1] >> (16 * (k & 1));
        const int a = s8(wv), b = s8(wv >> 8);
        int xi, xq;
        switch (k & 3) {                                          // (1, j, -1, -j)[n & 3], n = 8v + k
            case 0:  xi = a;        xq = b;        break;
            case 1:  xi = neg8(b);  xq = a;        break;
            case 2:  xi = neg8(a);  xq = neg8(b);  break;
            default: xi = b;        xq = neg8(a);  break;
        }
        const unsigned m = (n >= lo && n < hi) ? 1u : 0u;
        const unsigned wgt = (unsigned)(kR - (n - lo));
        acc[0] += m * (unsigned)xi;        acc[1] += m * (unsigned)xq;
        acc[2] += m * wgt * (unsigned)xi;  acc[3] += m * wgt * (unsigned)xq;
    }
}

// ---- whole segments, block sums on the MATRIX pipe (round 4) -------------------------------------------------------
// The block sums are int8 x int8 -> int32 contractions: S = sum_r (+-1 / 0) x_r,  W = sum_r (R - off_r)(+-1 / 0) x_r.
// Round 3's kernel (removed in round 5: it survived only behind a switch) formed them with sixteen v_dot4_i32_i8 per
// 16-byte vector: 37 vector instructions per vector ("K0 and the decoder").  V_MFMA_I32_16X16X64_I8 takes the same contraction off the vector pipes:
//   B (data)    lane l of a group of 64 consecutive vectors holds vector v0 + 64 g + l: column n = l & 15, k-block l >> 4;
//   A (weights) row i = lane & 15, k-block lane >> 4, one weight per byte of a vector (tools/mfma_i8_probe.hip):
//               row 0 / 1   the mixer's +-1 / 0 pattern for x_i / x_q              -> t   (sum over the column's vectors)
//               row 2 / 3   the same times -k, k = 0..7 the sample's place in the vector  -> nu  (-sum k x_k)
//               row 4 / 5   the same times the k-block 0..3                        -> sum blk * t
//   C (int32)   accumulates over the 13 groups of a block; magnitudes stay below 2^20, far from the accumulator's range
//               (the large factors -- R - off up to 6 401 -- are applied afterwards in wrapping 32-bit arithmetic).
// With v - v0 = 64 g + 16 blk + n the weight of vector v is wb0 - 8 (v - v0), so per block
//   W = sum_n [(wb0 - 8 n) T_n + N_n] - 128 sum_n B_n - 512 sum_g g T_g,   sum_g g T_g = G T - sum_{j=1..G} P_j
// (T_n, N_n, B_n: rows 0/1, 2/3, 4/5 of column n; P_j: row 0/1 after j groups, kept as one running add per group).
// What is left on the vector pipes per 16 bytes: the sign flip (4), the zero-byte test (8), two adds.  The two edge
// vectors of a block, and a block that holds a raw 0x00 byte, take the paths of the kernel above.
typedef int v4i __attribute__((ext_vector_type(4)));
struct alignas(16) MfmaWeights { signed char w[4][64][16]; };     // [group % 4][lane][byte of a vector]
constexpr MfmaWeights make_k0_weights() {
    MfmaWeights m{};
    for (int j = 0; j < 4; ++j)
        for (int lane = 0; lane < 64; ++lane) {
            const int row = lane & 15, blk = lane >> 4;
            for (int k = 0; k < 8; ++k) {
                // x_i = (+a, -b, -a, +b)[k & 3], x_q = (+b, +a, -b, -a)[k & 3]   (a = byte 2k, b = byte 2k + 1)
                const int ia[4] = {1, 0, -1, 0}, ib[4] = {0, -1, 0, 1}, qa[4] = {0, 1, 0, -1}, qb[4] = {1, 0, -1, 0};
                const int wa = (row & 1) ? qa[k & 3] : ia[k & 3], wb = (row & 1) ? qb[k & 3] : ib[k & 3];
                const int f = row < 2 ? 1 : row < 4 ? -k : row < 6 ? blk : row < 8 ? j : 0;
                m.w[j][lane][2 * k] = (signed char)(f * wa);
                m.w[j][lane][2 * k + 1] = (signed char)(f * wb);
            }
        }
    return m;
}
__constant__ MfmaWeights kK0Weights = make_k0_weights();

__device__ __forceinline__ void zero_byte_test(const u4 q, unsigned (&zacc)[4]) {
    zacc[0] |= (q.x - 0x01010101u) & ~q.x;
    zacc[1] |= (q.y - 0x01010101u) & ~q.y;
    zacc[2] |= (q.z - 0x01010101u) & ~q.z;
    zacc[3] |= (q.w - 0x01010101u) & ~q.w;
}

__global__ __launch_bounds__(256)
void cic_block_sums_mfma_kernel(const uint8_t* __restrict__ raw, size_t bytes_per_seg, int nblocks, int nseg,
                                int32_t* __restrict__ sums) {
    // grid: (quads of blocks, segments), or -- gridDim.y == 1 with fewer workgroups than there are quads -- a RESIDENT
    // grid whose workgroups take the (segment, quad) items in turn (WSPR_K0_RESIDENT: the front end then holds a fixed
    // number of wave slots per CU and leaves the others, and most of the vector pipes, to the decoder's kernels)
    const int lane = threadIdx.x & 63;
    const int quads = (nblocks + 3) / 4;
    const bool resident = gridDim.y == 1 && nseg > 1;
    const int items = resident ? quads * nseg : 1;
  for (int item = resident ? (int)blockIdx.x : 0; item < items; item += resident ? (int)gridDim.x : 1) {
    const int seg = resident ? item / quads : (int)blockIdx.y;
    const int b = (resident ? item - seg * quads : (int)blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (b >= nblocks) continue;                                   // wave-uniform
    const int lo = b * kR, hi = lo + kR;                          // the block's samples
    const int vi_lo = (lo + 7) >> 3, vi_hi = hi >> 3;             // its interior vectors [vi_lo, vi_hi): 799 or 800
    const int v_max = (int)((bytes_per_seg / 2 + 7) >> 3);        // vectors in a row
    const u4* __restrict__ vec = reinterpret_cast<const u4*>(raw + (size_t)seg * bytes_per_seg);
    v4i A[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) A[j] = *reinterpret_cast<const v4i*>(kK0Weights.w[j][lane]);
    v4i C = {0, 0, 0, 0};
    unsigned zacc[4] = {0u, 0u, 0u, 0u};
    auto group = [&](const u4 q, const v4i& Aj) {
        zero_byte_test(q, zacc);
        const v4i B = {(int)(q.x ^ 0x80808080u), (int)(q.y ^ 0x80808080u), (int)(q.z ^ 0x80808080u), (int)(q.w ^ 0x80808080u)};
        C = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aj, B, C, 0, 0, 0);
    };
    // the thirteenth group: the 31 or 32 interior vectors behind the twelve full groups (lanes beyond them carry zeros)
    const int vt = vi_lo + 768 + lane;
    const bool tail_lane = vt < vi_hi;
    // the two edge vectors (lanes 62, 63 of the vector path): one before the first interior vector, one behind the last
    int ve = 0, k_lo = 0, k_hi = 0;
    if (lane == 62) { ve = vi_lo - 1; k_lo = lo - 8 * ve; k_hi = (k_lo < 8) ? 8 : 0; }       // k_lo == 8: the block starts on a vector
    if (lane == 63) { ve = vi_hi; k_hi = hi - 8 * ve; }                                      // 0: it ends on one
    const u4* __restrict__ p0 = vec + vi_lo + lane;
    // groups g = 4 s + j: rows 6 / 7 weigh a group by j; the super-group index s is applied to what rows 0 / 1 hold after
    // every fourth group (P1, P2, P3: the sums before super-groups 1, 2, 3)
    unsigned pI = 0u, pQ = 0u;
    u4 qa[4], qb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) qa[u] = __builtin_nontemporal_load(p0 + 64 * u);
#pragma unroll
    for (int u = 0; u < 4; ++u) qb[u] = __builtin_nontemporal_load(p0 + 256 + 64 * u);
#pragma unroll
    for (int u = 0; u < 4; ++u) group(qa[u], A[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) qa[u] = __builtin_nontemporal_load(p0 + 512 + 64 * u);
    pI += (unsigned)C[0]; pQ += (unsigned)C[1];
#pragma unroll
    for (int u = 0; u < 4; ++u) group(qb[u], A[u]);
    u4 qt = __builtin_nontemporal_load(vec + min(vt, v_max - 1));
    const u4 qe = __builtin_nontemporal_load(vec + min(max(ve, 0), v_max - 1));
    pI += (unsigned)C[0]; pQ += (unsigned)C[1];
#pragma unroll
    for (int u = 0; u < 4; ++u) group(qa[u], A[u]);
    pI += (unsigned)C[0]; pQ += (unsigned)C[1];
    if (!tail_lane) qt = (u4){0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};           // x = 0 after the sign flip; no zero byte
    group(qt, A[0]);                                                                         // group 12 = super-group 3, j = 0
    // sum_s s T_s over the four super-groups = 3 T - (P1 + P2 + P3)
    unsigned acc[4] = {0u, 0u, 0u, 0u}, zedge[4] = {0u, 0u, 0u, 0u};
    block_sum_vector<true>(qe, kR + lo - 8 * ve, k_lo, k_hi, acc, zedge);                     // non-zero in lanes 62, 63 only
    if (lane >= 62) { zacc[0] |= zedge[0]; zacc[1] |= zedge[1]; zacc[2] |= zedge[2]; zacc[3] |= zedge[3]; }
    if (__any(((zacc[0] | zacc[1] | zacc[2] | zacc[3]) & 0x80808080u) != 0u)) {                      // clipping at the negative rail: exact path
        unsigned ex[4] = {0u, 0u, 0u, 0u};                         // (its own array: the call takes its address)
        for (int v = vi_lo - 1 + lane; v <= vi_hi; v += 64) {
            if (v < 0 || v >= v_max) continue;
            const u4 q = __builtin_nontemporal_load(vec + v);
            block_sum_vector_exact(q, 8 * v, lo, hi, ex);
        }
        acc[0] = ex[0]; acc[1] = ex[1]; acc[2] = ex[2]; acc[3] = ex[3];
    } else if (lane < 16) {
        // weight of vector v0 + 256 s + 64 j + 16 blk + n:  wb0 - 8 (256 s + 64 j + 16 blk + n)
        const unsigned wb = (unsigned)(kR + lo - 8 * vi_lo) - 2048u * 3u - 8u * (unsigned)lane;
        acc[0] += (unsigned)C[0];
        acc[1] += (unsigned)C[1];
        acc[2] += wb * (unsigned)C[0] + 2048u * pI + (unsigned)C[2];
        acc[3] += wb * (unsigned)C[1] + 2048u * pQ + (unsigned)C[3];
    } else if (lane < 32) {
        acc[2] -= 128u * (unsigned)C[0] + 512u * (unsigned)C[2];   // rows 4 / 5: k-block times t; rows 6 / 7: j times t
        acc[3] -= 128u * (unsigned)C[1] + 512u * (unsigned)C[3];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned s = wave_sum(acc[i]);
        if (lane == 0) sums[((size_t)seg * nblocks + b) * 4 + i] = (int32_t)s;
    }
  }
}

// Integrators at the decimation instants by parallel prefix sums (exact: arithmetic mod 2^32 is
// associative).  With P1 = inclusive scan of S:  I1(b) = P1[b],  I2(b) = scan_b( R*P1[b-1] + W_b ).
// One workgroup per (segment, rail); thread t owns a contiguous chunk of blocks.
constexpr int kScanThreads = 1024;

__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* lds, int tid) {
    lds[tid] = v;
    __syncthreads();
    for (int o = 1; o < kScanThreads; o <<= 1) {
        const unsigned add = (tid >= o) ? lds[tid - o] : 0u;
        __syncthreads();
        lds[tid] += add;
        __syncthreads();
    }
    const unsigned incl = lds[tid];
    __syncthreads();
    return incl - v;
}

__global__ __launch_bounds__(kScanThreads)
void cic_scan_kernel(const int32_t* __restrict__ sums, int nblocks, size_t seg_samples, uint32_t* __restrict__ x2out,
                     DecimState* __restrict__ states) {
    __shared__ unsigned lds[kScanThreads];
    const int seg = blockIdx.x, rail = blockIdx.y, tid = threadIdx.x;
    const int32_t* __restrict__ s = sums + (size_t)seg * nblocks * 4;
    uint32_t* __restrict__ out = x2out + ((size_t)seg * 2 + rail) * nblocks;
    const int phase = states ? (int)states[seg].phase : 0;
    const unsigned i1_0 = states ? states[seg].x1[rail] : 0u, i2_0 = states ? states[seg].x2[rail] : 0u;
    const int nb = states ? (int)((phase + seg_samples + kR - 1) / kR) : nblocks;    // incl. a partial tail
    const int nfull = states ? (int)((phase + seg_samples) / kR) : nblocks;
    const unsigned mult0 = (unsigned)(kR - phase);                // block 0 only has R - phase samples left
    const int per = (nb + kScanThreads - 1) / kScanThreads;
    const int lo = min(nb, tid * per), hi = min(nb, lo + per);

    unsigned sumS = 0;
    for (int b = lo; b < hi; ++b) sumS += (unsigned)s[4 * b + rail];
    const unsigned x1_start = i1_0 + block_exclusive_scan(sumS, lds, tid);     // I1 before block lo

    unsigned x1 = x1_start, sumT = 0;
    for (int b = lo; b < hi; ++b) {
        sumT += (b == 0 ? mult0 : (unsigned)kR) * x1 + (unsigned)s[4 * b + 2 + rail];
        x1 += (unsigned)s[4 * b + rail];
    }
    const unsigned x2_start = i2_0 + block_exclusive_scan(sumT, lds, tid);     // I2 before block lo

    x1 = x1_start;
    unsigned x2 = x2_start;
    for (int b = lo; b < hi; ++b) {
        x2 += (b == 0 ? mult0 : (unsigned)kR) * x1 + (unsigned)s[4 * b + 2 + rail];
        x1 += (unsigned)s[4 * b + rail];
        if (b < nfull) out[b] = x2;
        if (states && b == nb - 1) {                              // the integrators after the chunk's last sample
            const unsigned n_tail = (unsigned)((phase + seg_samples) - (size_t)nfull * kR);
            states[seg].x1[rail] = x1;
            states[seg].x2[rail] = (nb > nfull) ? x2 - ((unsigned)kR - n_tail) * x1 : x2;
        }
    }
}

// Two combs with a two-output delay (rtlsdr_wsprd.c:204-218): y2[b] = x2[b] - 2 x2[b-2] + x2[b-4]
// (mod 2^32, zero before the start), then the 33-tap FIR: 32 previous comb outputs (oldest first)
// on taps 0..31 and the new one on tap 32, summed in that order (rtlsdr_wsprd.c:220-234).
// x2 at decimation instant b of this chunk; b < 0 reaches into the carried history (zero without one)
__device__ __forceinline__ uint32_t x2_at(const uint32_t* __restrict__ x2, const uint32_t* __restrict__ hist, int b) {
    if (b >= 0) return x2[b];
    return (hist && b >= -kDecimHist) ? hist[kDecimHist + b] : 0u;
}
__device__ __forceinline__ float comb_out(const uint32_t* __restrict__ x2, const uint32_t* __restrict__ hist, int b) {
    if (b < 0 && !hist) return 0.0f;
    const uint32_t c0 = x2_at(x2, hist, b), c2 = x2_at(x2, hist, b - 2), c4 = x2_at(x2, hist, b - 4);
    const uint32_t y1 = c0 - c2, y1d = c2 - c4;       // y1[b], y1[b-2]
    return (float)(int32_t)(y1 - y1d);
}

__global__ __launch_bounds__(256)
void cic_fir_kernel(const uint32_t* __restrict__ x2all, int nblocks, size_t seg_samples, float* __restrict__ dI,
                    float* __restrict__ dQ, int* __restrict__ n_out, const DecimState* __restrict__ states) {
    const int seg = blockIdx.y;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    const int nfull = states ? (int)((states[seg].phase + seg_samples) / kR) : nblocks;
    if (m == 0 && n_out) n_out[seg] = nfull < kMaxSamples ? nfull : kMaxSamples;
    if (m >= nfull || m >= kMaxSamples) return;
#pragma unroll
    for (int rail = 0; rail < 2; ++rail) {
        const uint32_t* __restrict__ x2 = x2all + ((size_t)seg * 2 + rail) * nblocks;
        const uint32_t* __restrict__ hist = states ? states[seg].hist[rail] : nullptr;
        float acc = 0.0f;
        for (int j = 0; j < 32; ++j) {
            const float p = comb_out(x2, hist, m - 32 + j) * kFirTaps[j];
            acc += p;
        }
        const float p = comb_out(x2, hist, m) * kFirTaps[32];
        acc += p;
        (rail == 0 ? dI : dQ)[(size_t)seg * kIqStride + m] = acc;
    }
}

// Whole segments from a zero state need no integrator scan: the combs difference what the integrators summed,
//     y2[b] = x2[b] - 2 x2[b-2] + x2[b-4]
//           = R (S[b-1] + 2 S[b-2] + S[b-3]) + W[b] + W[b-1] - W[b-2] - W[b-3]        (mod 2^32, S = W = 0 for b < 0)
// (x2[b] - x2[b-2] = sum over the two blocks of R x1(before) + W, and the x1 differences are block sums S),
// exactly the value the scan + comb path produces because every step is arithmetic modulo 2^32.  One workgroup
// forms 256 + 32 comb outputs per rail from coalesced 16-byte loads of the block sums and runs the 33-tap FIR
// from LDS in the reference's tap order (rtlsdr_wsprd.c:220-234).
__global__ __launch_bounds__(256)
void cic_comb_fir_kernel(const int32_t* __restrict__ sums, int nblocks, float* __restrict__ dI, float* __restrict__ dQ,
                         int* __restrict__ n_out) {
    __shared__ int4 sb[256 + 32 + 3];                  // block sums of blocks m0-35 .. m0+255
    __shared__ float y2[2][256 + 32];                  // comb outputs of blocks m0-32 .. m0+255
    const int seg = blockIdx.y, tid = threadIdx.x, m0 = blockIdx.x * 256;
    if (blockIdx.x == 0 && tid == 0 && n_out) n_out[seg] = nblocks < kMaxSamples ? nblocks : kMaxSamples;
    const int4* __restrict__ s4 = reinterpret_cast<const int4*>(sums) + (size_t)seg * nblocks;
    for (int e = tid; e < 256 + 35; e += 256) {
        const int b = m0 - 35 + e;
        sb[e] = (b >= 0 && b < nblocks) ? s4[b] : make_int4(0, 0, 0, 0);
    }
    __syncthreads();
    for (int e = tid; e < 256 + 32; e += 256) {        // comb output of block m0 - 32 + e = sums index e + 3
        const int4 a0 = sb[e + 3], a1 = sb[e + 2], a2 = sb[e + 1], a3 = sb[e];
        const unsigned yi = (unsigned)kR * ((unsigned)a1.x + 2u * (unsigned)a2.x + (unsigned)a3.x) +
                            (unsigned)a0.z + (unsigned)a1.z - (unsigned)a2.z - (unsigned)a3.z;
        const unsigned yq = (unsigned)kR * ((unsigned)a1.y + 2u * (unsigned)a2.y + (unsigned)a3.y) +
                            (unsigned)a0.w + (unsigned)a1.w - (unsigned)a2.w - (unsigned)a3.w;
        const bool live = (m0 - 32 + e) >= 0;          // before the start the reference's delay line holds 0.0f
        y2[0][e] = live ? (float)(int32_t)yi : 0.0f;
        y2[1][e] = live ? (float)(int32_t)yq : 0.0f;
    }
    __syncthreads();
    const int m = m0 + tid;
    if (m >= nblocks || m >= kMaxSamples) return;
#pragma unroll
    for (int rail = 0; rail < 2; ++rail) {
        float acc = 0.0f;
        for (int j = 0; j < 32; ++j) {
            const float p = y2[rail][tid + j] * kFirTaps[j];
            acc += p;
        }
        const float p = y2[rail][tid + 32] * kFirTaps[32];
        acc += p;
        (rail == 0 ? dI : dQ)[(size_t)seg * kIqStride + m] = acc;
    }
}

// carry on: the last 36 integrator samples and the phase
__global__ __launch_bounds__(128)
void cic_carry_kernel(const uint32_t* __restrict__ x2all, int nblocks, size_t seg_samples, DecimState* __restrict__ states) {
    const int seg = blockIdx.x, tid = threadIdx.x;
    DecimState& st = states[seg];
    const int phase = (int)st.phase;
    const int nfull = (int)((phase + seg_samples) / kR);
    uint32_t v = 0;
    const int rail = tid / kDecimHist, k = tid % kDecimHist;
    if (tid < 2 * kDecimHist) {
        const int i = nfull - kDecimHist + k;
        v = (i >= 0) ? x2all[((size_t)seg * 2 + rail) * nblocks + i] : st.hist[rail][kDecimHist + i];
    }
    __syncthreads();
    if (tid < 2 * kDecimHist) st.hist[rail][k] = v;
    if (tid == 0) st.phase = (uint32_t)((phase + seg_samples) - (size_t)nfull * kR);
}
}  // namespace

// blocks a chunk of nsamp samples can touch (scratch pitch); without a state only whole blocks count
void front_end_constants(float* taps33, int* samples_per_output) {
    for (int i = 0; i < 33; ++i) taps33[i] = kFirTapsHost[i];
    *samples_per_output = kR;
}

int decimate_blocks(size_t nsamp, bool stateful) {
    return stateful ? (int)((nsamp + 2 * (size_t)kR - 2) / kR) : (int)(nsamp / kR);
}

// scratch: nseg * decimate_blocks() * (4 int32 + 2 uint32)
void launch_decimate(const uint8_t* raw, size_t bytes_per_seg, int nseg, float* dI, float* dQ,
                     int* n_out, int32_t* scratch, hipStream_t st, DecimState* states) {
    if (nseg <= 0) return;
    const size_t nsamp = bytes_per_seg / 2;
    const int nblocks = decimate_blocks(nsamp, states != nullptr);
    if (nblocks <= 0) return;
    int32_t* sums = scratch;
    uint32_t* x2 = reinterpret_cast<uint32_t*>(scratch + (size_t)nseg * nblocks * 4);
    // whole segments take the wave-per-block kernel (sample indices fit 31 bits: rows below 4 GiB);
    // WSPR_K0_KERNEL=general keeps them on the kernel that also serves carried states
    // (lab build only)
    static const bool general = [] { const char* e = lab_env("WSPR_K0_KERNEL"); return e && e[0] == 'g'; }();
    if (!states && !general && bytes_per_seg < ((size_t)1 << 32)) {
        {
            // WSPR_K0_RESIDENT=p (lab build only): a resident grid of p workgroups per CU instead of one workgroup per four blocks
            static const int per_cu = [] { const char* e = lab_env("WSPR_K0_RESIDENT"); return e ? atoi(e) : 0; }();
            const int quads = (nblocks + 3) / 4;
            if (per_cu > 0 && nseg > 1 && (long)quads * nseg > 256L * per_cu)
                hipLaunchKernelGGL(cic_block_sums_mfma_kernel, dim3(256 * per_cu, 1), dim3(256), 0, st, raw, bytes_per_seg,
                                   nblocks, nseg, sums);
            else
                hipLaunchKernelGGL(cic_block_sums_mfma_kernel, dim3(quads, nseg), dim3(256), 0, st, raw, bytes_per_seg,
                                   nblocks, nseg, sums);
        }
        hipLaunchKernelGGL(cic_comb_fir_kernel, dim3((nblocks + 255) / 256, nseg), dim3(256), 0, st, sums, nblocks,
                           dI, dQ, n_out);
        return;
    }
    hipLaunchKernelGGL(cic_block_sums_kernel, dim3((nblocks + 1) / 2, nseg), dim3(256), 0, st, raw,
                       bytes_per_seg, nblocks, sums, states);
    hipLaunchKernelGGL(cic_scan_kernel, dim3(nseg, 2), dim3(kScanThreads), 0, st, sums, nblocks, nsamp, x2, states);
    hipLaunchKernelGGL(cic_fir_kernel, dim3((nblocks + 255) / 256, nseg), dim3(256), 0, st, x2, nblocks, nsamp,
                       dI, dQ, n_out, states);
    if (states) hipLaunchKernelGGL(cic_carry_kernel, dim3(nseg), dim3(128), 0, st, x2, nblocks, nsamp, states);
}

#ifdef WSPR_LAB   // calibration kernel: lab build only
// Calibration for K0's roofline: the same read pattern as cic_block_sums_kernel (one workgroup per 12 802
// bytes of a row, aligned 16-byte non-temporal loads, four in flight per lane) with one add per dword instead
// of the mixer and block-sum arithmetic -- what the memory system delivers to a read-only stream of this shape.
namespace {
__global__ __launch_bounds__(256)
void calib_read_kernel(const uint8_t* __restrict__ raw, size_t bytes_per_seg, int nblocks, unsigned* __restrict__ out) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4* __restrict__ vec = reinterpret_cast<const u4*>(raw + (size_t)blockIdx.y * bytes_per_seg);
    const long first = (long)blockIdx.x * 2 * kR, last = min(first + 2 * kR, (long)(bytes_per_seg / 2));
    const long v_lo = first >> 3, v_end = (last + 7) >> 3;
    unsigned acc = 0;
    for (long v0 = v_lo + threadIdx.x; v0 < v_end; v0 += 4 * 256) {
        u4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long v = v0 + 256 * u;
            q[u] = __builtin_nontemporal_load(vec + ((v < v_end) ? v : v_lo));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += q[u].x + q[u].y + q[u].z + q[u].w;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0 && acc == 0x9e3779b9u) out[0] = acc;      // keeps the loads alive; practically never true
    (void)nblocks;
}
}  // namespace
void launch_calib_read(const uint8_t* raw, size_t bytes_per_seg, int nseg, unsigned* out, hipStream_t st) {
    const int nblocks = decimate_blocks(bytes_per_seg / 2, false);
    if (nseg <= 0 || nblocks <= 0) return;
    hipLaunchKernelGGL(calib_read_kernel, dim3((nblocks + 1) / 2, nseg), dim3(256), 0, st, raw, bytes_per_seg, nblocks, out);
}
#endif  // WSPR_LAB

}  // namespace wspr

// K2 -- candidate peak picker, K3 -- coarse (frequency, lag, drift) sync search.
//
// K2 replaces reference wsprd/wsprd.c:555-631, K3 replaces :646-678.  Both read
// the bin-major power spectrogram written by K1.  The time average is bound by that read
// (HBM: 578 796 algorithmic bytes per segment per pass); the coarse sync by its arithmetic
// (~250 000 separately rounded adds per candidate) once its rows arrive as contiguous lines.
//
// Float evaluation order is the reference's: every sum that feeds a threshold or
// an argmax is accumulated by ONE lane in the reference's loop order; lanes
// parallelise over independent sums (bins, or (freq, lag, drift) hypotheses).
#include "wspr_device.h"
#include <cstdlib>
#include <stdint.h>

#pragma clang fp contract(off)

namespace wspr {
const unsigned char* sync_vector();           // host copy of the sync vector (wspr_message.cpp)
namespace {

constexpr double kHalfDf = 375.0 / 256.0 / 2.0;      // (DF / 2.0), wsprd.c:615

// ------------------------------------------------------------------ K2 ------
// K2a: time-averaged spectrum (wsprd.c:556-561: psavg[bin] = sum over the time blocks, in block order).
// One wave per 64 bins.  A bin's row is contiguous in HBM (bin-major spectrogram), so the wave streams
// 64 rows x 128 bytes per chunk with 16-byte loads (8 lanes per row), parks the chunk in LDS and each
// lane then adds ITS bin's 32 values in time order -- the reference's serial sum, fully coalesced.
// The next chunk's loads are in flight while the current one is summed.
constexpr int kAvgPitch = 33;
__global__ __launch_bounds__(64)
void time_average_kernel(const float* __restrict__ ps, const int* __restrict__ seg_list, int blocks,
                         float* __restrict__ psavg) {
    __shared__ float tile[64 * kAvgPitch];
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int seg = seg_list ? seg_list[blockIdx.y] : (int)blockIdx.y;
    const int lane = threadIdx.x, b0 = blockIdx.x * 64;
    const int nb = min(64, kPsBins - b0);
    const int part = lane & 7, rsub = lane >> 3;
    const float* __restrict__ P = ps + ((size_t)seg * kPsBins + b0) * kPsTPitch + 4 * part;
    const int nchunks = (blocks + 31) / 32;
    f4 v[8];
    auto fetch = [&](int c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rsub + 8 * i;
            v[i] = (row < nb) ? *reinterpret_cast<const f4*>(P + (size_t)row * kPsTPitch + 32 * c) : f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch(0);
    float acc = 0.0f;
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float* t = tile + (rsub + 8 * i) * kAvgPitch + 4 * part;
            t[0] = v[i].x; t[1] = v[i].y; t[2] = v[i].z; t[3] = v[i].w;
        }
        __builtin_amdgcn_wave_barrier();
        if (c + 1 < nchunks) fetch(c + 1);
        const int nt = min(32, blocks - 32 * c);
        const float* __restrict__ mine = tile + lane * kAvgPitch;
        if (nt == 32) {
#pragma unroll
            for (int j = 0; j < 32; ++j) acc += mine[j];
        } else {
            for (int j = 0; j < nt; ++j) acc += mine[j];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < nb) psavg[(size_t)seg * kPsStride + b0 + lane] = acc;
}

// K2b: one workgroup per segment, everything after the time average (wsprd.c:565-631)
__global__ __launch_bounds__(512)
void pick_peaks_kernel(const float* __restrict__ psavg, const int* __restrict__ seg_list,
                       DevCand* __restrict__ cand, int* __restrict__ npk_out,
                       float* __restrict__ noise_out, float* __restrict__ smspec_out,
                       float min_snr, float floor_snr) {
    __shared__ float avg[kPsBins];
    __shared__ float sm[kSmooth];
    __shared__ float nrm[kSmooth];
    __shared__ float noise_s;
    __shared__ int   pk_bin[kMaxCand];
    __shared__ float pk_snr[kMaxCand];
    __shared__ int   kept_s;

    const int tid = threadIdx.x;
    const int seg = seg_list ? seg_list[blockIdx.x] : (int)blockIdx.x;
    if (tid < kPsBins) avg[tid] = psavg[(size_t)seg * kPsStride + tid];
    if (tid == 0) noise_s = 0.0f;
    __syncthreads();

    // 7-bin boxcar, bins 51+i-3 .. 51+i+3 (wsprd.c:565-573); column = bin - 48
    if (tid < kSmooth) {
        float acc = 0.0f;
#pragma unroll
        for (int d = 0; d < 7; ++d) acc += avg[tid + d];
        sm[tid] = acc;
        if (smspec_out) smspec_out[(size_t)seg * kSmooth + tid] = acc;
    }
    __syncthreads();

    // 30th percentile = element 122 of the ascending sort (wsprd.c:576-583).  Only the VALUE at that position
    // is used, so any correct ascending sort gives the reference's number: a bitonic network over 512 LDS
    // words (411 values + infinities), one compare-exchange per thread pair and step (45 steps; the rank
    // counting of round 1 cost every thread 411 comparisons and was a fifth of the stage at 8 192 segments).
    // Non-finite inputs only have to terminate: NaN fails every comparison and stays where it is.
    __shared__ float srt[512];
    srt[tid] = (tid < kSmooth) ? sm[tid] : __int_as_float(0x7f800000);
    __syncthreads();
    for (int len = 2; len <= 512; len <<= 1) {
        for (int stride = len >> 1; stride > 0; stride >>= 1) {
            if (tid < 256) {
                const int lo = ((tid & ~(stride - 1)) << 1) | (tid & (stride - 1));
                const int hi = lo | stride;
                const bool up = (lo & len) == 0;                      // ascending block
                const float a = srt[lo], b = srt[hi];
                if ((a > b) == up) { srt[lo] = b; srt[hi] = a; }
            }
            __syncthreads();
        }
    }
    if (tid == 0) noise_s = srt[122];
    __syncthreads();
    const float noise = noise_s;

    // snr-like normalisation with floor (wsprd.c:590-597)
    if (tid < kSmooth) {
        const float q = sm[tid] / noise;
        float v = (float)((double)q - 1.0);
        if (v < min_snr) v = floor_snr;
        nrm[tid] = v;
    }
    __syncthreads();

    // strict local maxima, first 200 only, then the +-110 Hz window (wsprd.c:608-629): ordered
    // compaction with wave ballots (the reference walks j = 1..409 once)
    __shared__ unsigned long long bal_peak[8], bal_keep[8];
    const int wv = tid >> 6, ln = tid & 63;
    const unsigned long long below = (ln == 0) ? 0ull : (~0ull >> (64 - ln));
    bool peak = false;
    if (tid >= 1 && tid < kSmooth - 1) {
        const float v = nrm[tid];
        peak = (v > nrm[tid - 1]) && (v > nrm[tid + 1]);
    }
    const unsigned long long bp = __ballot(peak);
    if (ln == 0) bal_peak[wv] = bp;
    __syncthreads();
    int found_before = __popcll(bp & below);
    for (int q = 0; q < wv; ++q) found_before += __popcll(bal_peak[q]);
    const float fj = (float)((double)(tid - 205) * kHalfDf);
    const bool keep = peak && (found_before < kMaxCand) && (fj >= -110.0f) && (fj <= 110.0f);
    const unsigned long long bk = __ballot(keep);
    if (ln == 0) bal_keep[wv] = bk;
    __syncthreads();
    if (keep) {
        int idx = __popcll(bk & below);
        for (int q = 0; q < wv; ++q) idx += __popcll(bal_keep[q]);
        pk_bin[idx] = tid;
        pk_snr[idx] = (float)(10.0 * (double)log10f(nrm[tid]) - (double)26.3f);
    }
    if (tid == 0) {
        int kept = 0;
        for (int q = 0; q < 8; ++q) kept += __popcll(bal_keep[q]);
        kept_s = kept;
        npk_out[seg] = kept;
        if (noise_out) noise_out[seg] = noise;
    }
    __syncthreads();

    // stable sort by snr, strongest first (wsprd.c:631; glibc qsort is a merge sort)
    const int kept = kept_s;
    if (tid < kept) {
        const float mine = pk_snr[tid];
        int rank = 0;
        for (int j = 0; j < kept; ++j) {
            const float o = pk_snr[j];
            rank += (o > mine) || (o == mine && j < tid);
        }
        const int j = pk_bin[tid];
        DevCand cd;
        cd.freq  = (float)((double)(j - 205) * kHalfDf);
        cd.snr   = mine;
        cd.peak  = nrm[j];
        cd.shift = 0;
        cd.drift = 0.0f;
        cd.sync  = 0.0f;
        cd.bin   = j;
        cand[(size_t)seg * kMaxCand + rank] = cd;
    }
}

// ------------------------------------------------------------------ K3 ------
// Coarse (frequency, lag, drift) search, wsprd.c:646-678: per candidate 3 frequency bins x 32 lags x
// (2 maxdrift + 1) drifts, each a 162-term serial float sum of sqrt(ps) values.
//
// Reference quirks reproduced:
//  * "/ DF" expands to "/375.0/256.0", so the per-symbol drift offset only ever lowers the bin by one,
//    for (k > 81, drift < 0) or (k < 81, drift > 0): three distinct patterns; with a strict '>' the
//    first drift of each sign wins, i.e. the label is -maxdrift, 0 or +1.
//  * a negative time index reads the previous bin's row, 347 + index.
//
// Mapping: one workgroup = two candidates x three waves; wave = frequency bin ifr, lane = (candidate,
// lag).  The 11 bin rows a candidate can touch (bins if0-6 .. if0+4, contiguous 1408-byte rows of the
// bin-major spectrogram) are staged once as sqrt(ps) in LDS with 16-byte loads.  A lane then walks the
// 162 symbols once for the THREE drift patterns of its (frequency, lag): the eight amplitudes it reads
// per symbol serve all three, and the patterns share their history -- up to symbol 80 a pattern's sums
// depend only on the bin it reads (ifr for drift <= 0, ifr-1 for drift > 0), so two running sums stand
// for the three and fan out at symbol 81.  Every sum is accumulated term by term in the reference's
// order; two independent sums share one packed-fp32 instruction (each half an ordinary IEEE add).
constexpr int kCsRows = 11;
constexpr int kCsPitch = kPsTPitch;
constexpr int kCsThreads = 192;
typedef float cs2 __attribute__((ext_vector_type(2)));
struct SyncBits { uint32_t w[6]; };          // the 162-bit sync vector (wsprd.c:84-93), bit k of word k/32

template <bool kFull>
__global__ __launch_bounds__(kCsThreads)
void coarse_sync_kernel(const float* __restrict__ ps, const int* __restrict__ seg_list, int blocks,
                        DevCand* __restrict__ cand, const int* __restrict__ npk, int maxdrift, const SyncBits pr3) {
    __shared__ __attribute__((aligned(16))) float amp[2][kCsRows * kCsPitch];
    __shared__ float res_best[2][3];
    __shared__ int res_arg[2][3];
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, fi = tid >> 6, lane = tid & 63, half = lane >> 5, k0 = (lane & 31) - 10;
    const int seg = seg_list ? seg_list[blockIdx.x] : (int)blockIdx.x;
    const float* __restrict__ P = ps + (size_t)seg * kPsBins * kPsTPitch;
    const int ncand = min(npk[seg], kMaxCand);
    const int npat = (maxdrift > 0) ? 3 : 1;
    constexpr int kWords = kCsRows * (kCsPitch / 4);              // 968 16-byte words per candidate, rows contiguous
    constexpr int kPerThread = (2 * kWords + kCsThreads - 1) / kCsThreads;   // 11
    // the sync bits in six scalar registers (indexing the argument struct dynamically would become a memory
    // load whose wait also drains the LDS queue)
    const uint32_t w0 = pr3.w[0], w1 = pr3.w[1], w2 = pr3.w[2], w3 = pr3.w[3], w4 = pr3.w[4], w5 = pr3.w[5];

    for (int pair = blockIdx.y; 2 * pair < ncand; pair += gridDim.y) {
        __syncthreads();                                 // the previous pair's reads are done
        // ---- stage sqrt(ps) of both candidates' bin rows: every load of the pair in flight at once ----
        const int c_a = 2 * pair, c_b = 2 * pair + 1;
        const int if0_a = (int)((double)cand[(size_t)seg * kMaxCand + c_a].freq / kHalfDf + 256.0);
        const int if0_b = (c_b < ncand) ? (int)((double)cand[(size_t)seg * kMaxCand + c_b].freq / kHalfDf + 256.0) : if0_a;
        const float* __restrict__ src_a = P + (size_t)(if0_a - 6 - kPsBin0) * kPsTPitch;
        const float* __restrict__ src_b = P + (size_t)(if0_b - 6 - kPsBin0) * kPsTPitch;
        f4 v[kPerThread];
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) {
            const int e = min(tid + kCsThreads * u, 2 * kWords - 1);
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>((e < kWords ? src_a + 4 * e : src_b + 4 * (e - kWords))));
        }
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) {
            const int e = tid + kCsThreads * u;
            if (e < 2 * kWords) {
                f4 r;
                r.x = sqrtf(v[u].x); r.y = sqrtf(v[u].y); r.z = sqrtf(v[u].z); r.w = sqrtf(v[u].w);
                *reinterpret_cast<f4*>(&amp[0][0] + 4 * e) = r;             // amp[1] follows amp[0]
            }
        }
        __syncthreads();

        const int c = 2 * pair + half;
        // this wave: ifr = if0 - 1 + fi.  Bin "lo" = ifr - 1 (tones in staged rows fi+1, +3, +5, +7), bin
        // "hi" = ifr (rows fi+2, +4, +6, +8); a[q] below = staged row fi + 1 + q, so (a[2t], a[2t+1]) is tone
        // t of (lo, hi) -- the natural operand pair of the packed adds.
        const float* __restrict__ A = amp[half] + (fi + 1) * kCsPitch + k0;
        auto load_any = [&](int k, float (&a)[8]) {              // any symbol (negative time index, short record)
            const int kidx = k0 + 2 * k;
            int off = 2 * k;
            if (kidx < 0) off += blocks - kCsPitch;              // previous bin's row, 347 + index (Q2)
            if (!kFull && kidx >= blocks) off = -k0;
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = A[q * kCsPitch + off];
        };
        // Even and odd symbols read through two base pointers the compiler cannot relate: otherwise it fuses the
        // loads of symbols k and k+1 (two dwords apart) into ds_read2_b32 pairs, whose halves then have to be
        // shuffled into the (lo, hi) operand pairs with a dozen moves per step.
        int odd_off = 2;
        asm volatile("" : "+v"(odd_off));
        auto load = [&](int k, float (&a)[8]) {                  // symbols >= 5: the time index is never negative
            if (!kFull) { load_any(k, a); return; }
            const int off = (k & 1) ? odd_off + 2 * (k - 1) : 2 * k;
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = A[q * kCsPitch + off];
        };
        // sync-vector sign as a scalar float (wave-uniform: no vector select)
        auto sign_of = [&](int k) {
            const uint32_t w = k < 96 ? (k < 32 ? w0 : (k < 64 ? w1 : w2)) : (k < 128 ? w3 : (k < 160 ? w4 : w5));
            return __int_as_float(__builtin_amdgcn_readfirstlane((int)(((w >> (k & 31)) & 1u) ? 0x3f800000u : 0xbf800000u)));
        };
        // halves: .x = bin lo, .y = bin hi
        cs2 S = {0.f, 0.f}, W = {0.f, 0.f};
        auto step1 = [&](int k, const float (&a)[8]) {
            const float sg = sign_of(k);
            const cs2 P0 = {a[0], a[1]}, P1 = {a[2], a[3]}, P2 = {a[4], a[5]}, P3 = {a[6], a[7]};
            const cs2 m = ((P1 + P3) - (P0 + P2)) * sg;          // (p1 + p3) - (p0 + p2), signed by the sync vector
            const cs2 nS = S + m;
            const cs2 nW = (((W + P0) + P1) + P2) + P3;
            if (kFull || k0 + 2 * k < blocks) { S = nS; W = nW; }
        };
        // symbols 0..80, the next symbol's amplitudes in flight while the current one is folded in
        float a0[8], a1[8];
        load_any(0, a0);
#pragma unroll
        for (int k = 0; k < 6; k += 2) {                         // 0..5: lags < 0 still reach before the record
            load_any(k + 1, a1);
            step1(k, a0);
            load_any(k + 2, a0);
            step1(k + 1, a1);
        }
        for (int k = 6; k < 80; k += 2) {
            load(k + 1, a1);
            step1(k, a0);
            load(k + 2, a0);
            step1(k + 1, a1);
        }
        load(81, a1);
        step1(80, a0);
        // fan out at symbol 81.  U = (pattern 0, pattern 2): pattern 0 (drift < 0) continues bin hi's history
        // and reads bin hi at symbol 81, bin lo after it; pattern 2 (drift > 0) continues bin lo's history and
        // reads bin hi from 81 on.  V = pattern 1 (no drift): bin hi throughout.
        cs2 SU = {S.y, S.x}, WU = {W.y, W.x};
        float SV = S.y, WV = W.y;
        {   // symbol 81: all three read bin hi
            const float sg = sign_of(81);
            const cs2 P0 = {a1[0], a1[1]}, P1 = {a1[2], a1[3]}, P2 = {a1[4], a1[5]}, P3 = {a1[6], a1[7]};
            const cs2 m = ((P1 + P3) - (P0 + P2)) * sg;
            const cs2 nSU = SU + m.y;
            const cs2 nWU = (((WU + P0.y) + P1.y) + P2.y) + P3.y;
            const float nSV = SV + m.y;
            const float nWV = (((WV + P0.y) + P1.y) + P2.y) + P3.y;
            if (kFull || k0 + 2 * 81 < blocks) { SU = nSU; WU = nWU; SV = nSV; WV = nWV; }
        }
        auto step2 = [&](int k, const float (&a)[8]) {           // symbols 82..161: U reads (lo, hi), V reads hi
            const float sg = sign_of(k);
            const cs2 P0 = {a[0], a[1]}, P1 = {a[2], a[3]}, P2 = {a[4], a[5]}, P3 = {a[6], a[7]};
            const cs2 m = ((P1 + P3) - (P0 + P2)) * sg;
            const cs2 nSU = SU + m;
            const cs2 nWU = (((WU + P0) + P1) + P2) + P3;
            const float nSV = SV + m.y;
            const float nWV = (((WV + P0.y) + P1.y) + P2.y) + P3.y;
            if (kFull || k0 + 2 * k < blocks) { SU = nSU; WU = nWU; SV = nSV; WV = nWV; }
        };
        load(82, a0);
        for (int k = 82; k < kNSymD; k += 2) {               // 82 .. 161
            load(k + 1, a1);
            step2(k, a0);
            if (k + 2 < kNSymD) load(k + 2, a0);
            step2(k + 1, a1);
        }
        // ---- first hypothesis (in the reference's loop order) with the strictly largest metric ------
        float best = -1e30f;
        int arg = -1;
        {
            const float r0 = SU.x / WU.x, r1 = SV / WV, r2 = SU.y / WU.y;
            const int hb = fi * 32 * npat + (k0 + 10) * npat;
            if (npat == 3) {
                if (r0 > best) { best = r0; arg = hb; }
                if (r1 > best) { best = r1; arg = hb + 1; }
                if (r2 > best) { best = r2; arg = hb + 2; }
            } else {
                if (r1 > best) { best = r1; arg = hb; }
            }
        }
        if (c >= ncand) { best = -1e30f; arg = -1; }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {               // within the candidate's 32 lanes
            const float ob = __shfl_xor(best, o);
            const int oa = __shfl_xor(arg, o);
            const bool take = (oa >= 0) && (arg < 0 || ob > best || (ob == best && oa < arg));
            if (take) { best = ob; arg = oa; }
        }
        if ((lane & 31) == 0) { res_best[half][fi] = best; res_arg[half][fi] = arg; }
        __syncthreads();
        if (tid < 2) {                                   // one lane per candidate: the three frequency bins in order
            const int cc = 2 * pair + tid;
            float b = -1e30f;
            int ar = -1;
            for (int f = 0; f < 3; ++f)
                if (res_arg[tid][f] >= 0 && (ar < 0 || res_best[tid][f] > b)) { b = res_best[tid][f]; ar = res_arg[tid][f]; }
            if (cc < ncand && ar >= 0) {
                DevCand cd = cand[(size_t)seg * kMaxCand + cc];
                const int i0 = tid ? if0_b : if0_a;
                const int f = ar / (32 * npat);
                const int rem = ar - f * 32 * npat;
                const int kk = rem / npat - 10;
                const int pat = (npat == 3) ? rem % 3 : 1;
                cd.shift = 128 * (kk + 1);
                cd.drift = (pat == 0) ? (float)(-maxdrift) : (pat == 2 ? 1.0f : 0.0f);
                cd.freq  = (float)((double)(i0 - 1 + f - 256) * kHalfDf);
                cd.sync  = b;
                cand[(size_t)seg * kMaxCand + cc] = cd;
            }
        }
    }
}

// The same search with ONE lane per (candidate, lag) holding all three frequency bins (full-length records).
// coarse_sync_kernel issues eight LDS reads per lane and symbol for ten packed instructions, three waves per
// candidate pair: with four SIMDs on one LDS the return path is busy 64 clocks per 40 of arithmetic.  The three
// bins a candidate searches read only FOUR distinct tone sets D_j = bin if0 - 2 + j (rows j + 2 t of the ten
// bins if0 - 5 .. if0 + 4): a lane keeps one running (sync, power) pair per tone set for the whole frame --
// that is pattern 1 (no drift) of bin ifr = D_(fi+1) and, up to symbol 80, the shared history of the two
// drifting patterns -- plus the six drifting sums after the fan-out at symbol 81.  Per symbol: TEN amplitudes
// read once, 18 packed instructions before the fan-out and 38 after it for nine hypotheses (28 % fewer vector
// instructions, 58 % fewer LDS reads).  Every sum still receives the reference's terms one by one in its order;
// the packed pairs are (D_0, D_1) and (D_2, D_3), whose operands are adjacent rows = adjacent registers.
// One wave = two candidates; the amplitudes are staged in two halves of 81 symbols (196 columns of the ten
// rows: 15.7 KB per wave, ten waves per CU instead of five with whole rows).
constexpr int kCsChunkSyms = 81;                                 // the fan-out symbol 81 opens the second half
constexpr int kCsChunkCols = 196;                                // 2 x 80 + 32 lags, widened to whole 16-byte words
// The ten staged rows are kept as FIVE rows of pairs (bins if0 - 5 + 2 i and + 2 i + 1 side by side): the sums below
// work on exactly these pairs, so one 8-byte LDS read delivers a packed operand in one register pair (with ten rows of
// floats the compiler read ten words per symbol and moved them together: a quarter of the kernel's vector
// instructions were moves).  A staging lane owns two time columns of one pair row: two 8-byte loads, four square
// roots, one 16-byte store -- consecutive lanes, consecutive 16-byte words.
constexpr int kCsPairWords = 5 * (kCsChunkCols / 2);             // 490 16-byte words per candidate and half
constexpr int kCsChunkLoads = (2 * kCsPairWords + 63) / 64;      // 16 per lane
static_assert(kNSymD == 2 * kCsChunkSyms, "two halves of 81 symbols");

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4)))
void coarse_sync_lane_kernel(const float* __restrict__ ps, const int* __restrict__ seg_list,
                             DevCand* __restrict__ cand, const int* __restrict__ npk, int maxdrift, const SyncBits pr3) {
    __shared__ __attribute__((aligned(16))) cs2 amp[2][5 * kCsChunkCols];     // [candidate of the wave][pair row][time]
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int blocks = kMaxBlocks;
    const int lane = threadIdx.x, half = lane >> 5, k0 = (lane & 31) - 10;
    const int seg = seg_list ? seg_list[blockIdx.x] : (int)blockIdx.x;
    const float* __restrict__ P = ps + (size_t)seg * kPsBins * kPsTPitch;
    const int ncand = min(npk[seg], kMaxCand);
    const int npat = (maxdrift > 0) ? 3 : 1;
    const uint32_t w0 = pr3.w[0], w1 = pr3.w[1], w2 = pr3.w[2], w3 = pr3.w[3], w4 = pr3.w[4], w5 = pr3.w[5];
    auto sign_of = [&](int k) {                                  // sync-vector sign as a scalar float (wave-uniform)
        const uint32_t w = k < 96 ? (k < 32 ? w0 : (k < 64 ? w1 : w2)) : (k < 128 ? w3 : (k < 160 ? w4 : w5));
        return __int_as_float(__builtin_amdgcn_readfirstlane((int)(((w >> (k & 31)) & 1u) ? 0x3f800000u : 0xbf800000u)));
    };

    for (int pair = blockIdx.y; 2 * pair < ncand; pair += gridDim.y) {
        const int c_a = 2 * pair, c_b = 2 * pair + 1;
        const int if0_a = (int)((double)cand[(size_t)seg * kMaxCand + c_a].freq / kHalfDf + 256.0);
        const int if0_b = (c_b < ncand) ? (int)((double)cand[(size_t)seg * kMaxCand + c_b].freq / kHalfDf + 256.0) : if0_a;
        // staged row q = bin if0 - 5 + q, q = 0..9
        const float* __restrict__ src_a = P + (size_t)(if0_a - 5 - kPsBin0) * kPsTPitch;
        const float* __restrict__ src_b = P + (size_t)(if0_b - 5 - kPsBin0) * kPsTPitch;
        // half h covers columns t0 .. t0 + 195 with t0 = -12 (lags down to -10) or 152
        auto stage = [&](int t0) {
            constexpr int kRound = 8;                            // words in flight per lane
            static_assert(kCsChunkLoads % kRound == 0, "whole rounds");
            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll 1
            for (int u0 = 0; u0 < kCsChunkLoads; u0 += kRound) {
                f2 lo[kRound], hi[kRound];                       // two time columns of the even and of the odd row of a pair
#pragma unroll
                for (int u = 0; u < kRound; ++u) {
                    const int e = min(lane + 64 * (u0 + u), 2 * kCsPairWords - 1);
                    const int cb = e >= kCsPairWords, el = e - cb * kCsPairWords, i = el / (kCsChunkCols / 2), w = el - i * (kCsChunkCols / 2);
                    const int t = t0 + 2 * w;
                    const float* __restrict__ row = (cb ? src_b : src_a) + (size_t)(2 * i) * kPsTPitch;
                    // a negative time index reads the previous bin's row, 347 + index (Q2): t = -12 .. -2 of the first half
                    const float* __restrict__ from = (t >= 0) ? row + t : row - kPsTPitch + blocks + t;
                    if (t >= 0) {
                        lo[u] = __builtin_nontemporal_load(reinterpret_cast<const f2*>(from));
                        hi[u] = __builtin_nontemporal_load(reinterpret_cast<const f2*>(from + kPsTPitch));
                    } else {
                        lo[u].x = from[0]; lo[u].y = from[1];
                        hi[u].x = from[kPsTPitch]; hi[u].y = from[kPsTPitch + 1];
                    }
                }
#pragma unroll
                for (int u = 0; u < kRound; ++u) {
                    const int e = lane + 64 * (u0 + u);
                    if (e < 2 * kCsPairWords) {
                        f4 r;                                     // (even row, odd row) at t, then at t + 1
                        r.x = sqrtf(lo[u].x); r.y = sqrtf(hi[u].x); r.z = sqrtf(lo[u].y); r.w = sqrtf(hi[u].y);
                        *reinterpret_cast<f4*>(&amp[0][0] + 2 * e) = r;       // amp[1] follows amp[0]
                    }
                }
            }
        };
        // per-tone-set sums for the whole frame: S01 = (D_0, D_1), S23 = (D_2, D_3); tone t of D_j = a[j + 2 t]
        cs2 S01 = {0.f, 0.f}, W01 = {0.f, 0.f}, S23 = {0.f, 0.f}, W23 = {0.f, 0.f};
        cs2 m01 = {0.f, 0.f}, m23 = {0.f, 0.f};
        auto bins = [&](int k, const float (&a)[10]) {
            const float sg = sign_of(k);
            const cs2 A0 = {a[0], a[1]}, A1 = {a[2], a[3]}, A2 = {a[4], a[5]}, A3 = {a[6], a[7]}, A4 = {a[8], a[9]};
            m01 = ((A1 + A3) - (A0 + A2)) * sg;                   // tones of (D_0, D_1): pairs 0..3
            m23 = ((A2 + A4) - (A1 + A3)) * sg;                   // tones of (D_2, D_3): pairs 1..4
            S01 = S01 + m01;
            S23 = S23 + m23;
            W01 = (((W01 + A0) + A1) + A2) + A3;
            W23 = (((W23 + A1) + A2) + A3) + A4;
        };
        // ---- symbols 0..80 ----
        __syncthreads();                                         // the previous pair has been consumed
        stage(-12);
        __syncthreads();
        {
            const cs2* __restrict__ A = amp[half] + (k0 + 12);
            auto load = [&](int kk, float (&a)[10]) {
#pragma unroll
                for (int i = 0; i < 5; ++i) { const cs2 v = A[i * kCsChunkCols + 2 * kk]; a[2 * i] = v.x; a[2 * i + 1] = v.y; }
            };
            float a0[10], a1[10];
            load(0, a0);
#pragma unroll 1
            for (int k = 0; k < 80; k += 2) {
                load(k + 1, a1);
                bins(k, a0);
                load(k + 2, a0);
                bins(k + 1, a1);
            }
            bins(80, a0);
        }
        // fan-out: pattern 0 of bin fi (drift < 0) continues D_(fi+1), pattern 2 (drift > 0) continues D_fi
        cs2 X0S = {S01.y, S23.x}, X0W = {W01.y, W23.x};          // pattern 0 of fi = 0, 1 (reads D_0, D_1 from 82 on)
        float X0S2 = S23.y, X0W2 = W23.y;                        // pattern 0 of fi = 2 (reads D_2)
        float X2S0 = S01.x, X2W0 = W01.x;                        // pattern 2 of fi = 0 (reads D_1)
        cs2 X2S = {S01.y, S23.x}, X2W = {W01.y, W23.x};          // pattern 2 of fi = 1, 2 (reads D_2, D_3)
        // ---- symbols 81..161 ----
        __syncthreads();
        stage(152);
        __syncthreads();
        {
            const cs2* __restrict__ A = amp[half] + (k0 + 2 * kCsChunkSyms - 152);
            auto load = [&](int kk, float (&a)[10]) {            // kk = symbol - 81
#pragma unroll
                for (int i = 0; i < 5; ++i) { const cs2 v = A[i * kCsChunkCols + 2 * kk]; a[2 * i] = v.x; a[2 * i + 1] = v.y; }
            };
            auto drifting = [&](const float (&a)[10]) {          // after bins(k, a): m01 / m23 are this symbol's
                const cs2 A0 = {a[0], a[1]}, A1 = {a[2], a[3]}, A2 = {a[4], a[5]}, A3 = {a[6], a[7]}, A4 = {a[8], a[9]};
                X0S = X0S + m01;                                 // (D_0, D_1)
                X0W = (((X0W + A0) + A1) + A2) + A3;
                X2S = X2S + m23;                                 // (D_2, D_3)
                X2W = (((X2W + A1) + A2) + A3) + A4;
                X0S2 = X0S2 + m23.x;                             // D_2
                X0W2 = (((X0W2 + a[2]) + a[4]) + a[6]) + a[8];
                X2S0 = X2S0 + m01.y;                             // D_1
                X2W0 = (((X2W0 + a[1]) + a[3]) + a[5]) + a[7];
            };
            float a0[10], a1[10];
            load(0, a1);
            load(1, a0);
            bins(81, a1);                                        // symbol 81: every pattern of bin fi reads D_(fi+1)
            {
                const float p0[3] = {a1[1], a1[2], a1[3]}, p1[3] = {a1[3], a1[4], a1[5]};
                const float p2[3] = {a1[5], a1[6], a1[7]}, p3[3] = {a1[7], a1[8], a1[9]};
                const float mh[3] = {m01.y, m23.x, m23.y};
                X0S.x = X0S.x + mh[0]; X0S.y = X0S.y + mh[1]; X0S2 = X0S2 + mh[2];
                X2S0 = X2S0 + mh[0];   X2S.x = X2S.x + mh[1]; X2S.y = X2S.y + mh[2];
                X0W.x = (((X0W.x + p0[0]) + p1[0]) + p2[0]) + p3[0];
                X0W.y = (((X0W.y + p0[1]) + p1[1]) + p2[1]) + p3[1];
                X0W2  = (((X0W2  + p0[2]) + p1[2]) + p2[2]) + p3[2];
                X2W0  = (((X2W0  + p0[0]) + p1[0]) + p2[0]) + p3[0];
                X2W.x = (((X2W.x + p0[1]) + p1[1]) + p2[1]) + p3[1];
                X2W.y = (((X2W.y + p0[2]) + p1[2]) + p2[2]) + p3[2];
            }
#pragma unroll 1
            for (int k = 82; k < kNSymD; k += 2) {               // 82 .. 161
                load(k + 1 - 81, a1);
                bins(k, a0);
                drifting(a0);
                if (k + 2 < kNSymD) load(k + 2 - 81, a0);
                bins(k + 1, a1);
                drifting(a1);
            }
        }
        // ---- first hypothesis (in the reference's loop order) with the strictly largest metric ------
        float best = -1e30f;
        int arg = -1;
        {
            const float r0[3] = {X0S.x / X0W.x, X0S.y / X0W.y, X0S2 / X0W2};
            const float r1[3] = {S01.y / W01.y, S23.x / W23.x, S23.y / W23.y};
            const float r2[3] = {X2S0 / X2W0, X2S.x / X2W.x, X2S.y / X2W.y};
#pragma unroll
            for (int fi = 0; fi < 3; ++fi) {
                const int hb = fi * 32 * npat + (k0 + 10) * npat;
                if (npat == 3) {
                    if (r0[fi] > best) { best = r0[fi]; arg = hb; }
                    if (r1[fi] > best) { best = r1[fi]; arg = hb + 1; }
                    if (r2[fi] > best) { best = r2[fi]; arg = hb + 2; }
                } else {
                    if (r1[fi] > best) { best = r1[fi]; arg = hb; }
                }
            }
        }
        const int c = 2 * pair + half;
        if (c >= ncand) { best = -1e30f; arg = -1; }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {               // within the candidate's 32 lanes
            const float ob = __shfl_xor(best, o);
            const int oa = __shfl_xor(arg, o);
            const bool take = (oa >= 0) && (arg < 0 || ob > best || (ob == best && oa < arg));
            if (take) { best = ob; arg = oa; }
        }
        if ((lane & 31) == 0 && c < ncand && arg >= 0) {
            DevCand cd = cand[(size_t)seg * kMaxCand + c];
            const int i0 = half ? if0_b : if0_a;
            const int f = arg / (32 * npat);
            const int rem = arg - f * 32 * npat;
            const int kk = rem / npat - 10;
            const int pat = (npat == 3) ? rem % 3 : 1;
            cd.shift = 128 * (kk + 1);
            cd.drift = (pat == 0) ? (float)(-maxdrift) : (pat == 2 ? 1.0f : 0.0f);
            cd.freq  = (float)((double)(i0 - 1 + f - 256) * kHalfDf);
            cd.sync  = best;
            cand[(size_t)seg * kMaxCand + c] = cd;
        }
    }
}
}  // namespace

void launch_time_average(const float* ps, const int* seg_list, int nseg_active, int blocks, float* psavg,
                         hipStream_t st) {
    if (nseg_active <= 0) return;
    hipLaunchKernelGGL(time_average_kernel, dim3((kPsBins + 63) / 64, nseg_active), dim3(64), 0, st, ps, seg_list,
                       blocks, psavg);
}

void launch_pick_peaks(const float* ps, const int* seg_list, int nseg_active, int blocks, float* psavg,
                       DevCand* cand, int* npk, float* noise_out, float* smspec_out,
                       const DeviceTables& t, hipStream_t st, bool have_avg) {
    if (nseg_active <= 0) return;
    if (!have_avg) launch_time_average(ps, seg_list, nseg_active, blocks, psavg, st);
    hipLaunchKernelGGL(pick_peaks_kernel, dim3(nseg_active), dim3(512), 0, st, psavg, seg_list,
                       cand, npk, noise_out, smspec_out, t.min_snr, t.floor_snr);
}

void launch_coarse_sync(const float* ps, const int* seg_list, int nseg_active, int blocks,
                        DevCand* cand, const int* npk, int maxdrift,
                        const DeviceTables& t, hipStream_t st) {
    if (nseg_active <= 0) return;
    static const SyncBits bits = [] {
        SyncBits b{};
        const unsigned char* pr3 = sync_vector();
        for (int k = 0; k < kNSymD; ++k) if (pr3[k]) b.w[k >> 5] |= 1u << (k & 31);
        return b;
    }();
    constexpr int gy = 16;                                   // candidate pairs in flight per segment
    // Full-length records of large batches take the lane-per-(candidate, lag) kernel: its single-wave workgroups
    // pay two staging round trips each, which a launch of a few hundred candidate pairs cannot hide (1 024
    // single-signal segments: 59 vs 46 us; 8 192 x 10 signals: 0.93 vs 1.15 ms).  WSPR_K3_KERNEL=waves / lane force one.
    static const int forced = [] { const char* e = lab_env("WSPR_K3_KERNEL"); return !e ? 0 : (e[0] == 'w' ? 1 : 2); }();
    const bool lane_kernel = forced ? forced == 2 : nseg_active >= 1536;
    if (blocks == kMaxBlocks && lane_kernel)
        hipLaunchKernelGGL(coarse_sync_lane_kernel, dim3(nseg_active, gy), dim3(64), 0, st, ps, seg_list, cand, npk,
                           maxdrift, bits);
    else if (blocks == kMaxBlocks)
        hipLaunchKernelGGL(coarse_sync_kernel<true>, dim3(nseg_active, gy), dim3(kCsThreads), 0, st, ps, seg_list, blocks,
                           cand, npk, maxdrift, bits);
    else
        hipLaunchKernelGGL(coarse_sync_kernel<false>, dim3(nseg_active, gy), dim3(kCsThreads), 0, st, ps, seg_list, blocks,
                           cand, npk, maxdrift, bits);
}

}  // namespace wspr

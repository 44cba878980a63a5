// K7 -- coherent subtraction of a decoded signal from the segment's IQ.
//
// Replaces reference subtract_signal2(), wsprd/wsprd.c:316-413:
//   r(t) = exp(j*phi(t))  regenerated from the 162 channel symbols,
//   c(t) = LPF[s(t) * conj(r(t))]   360-tap normalised sine window,
//   s'(t) = s(t) - c(t) r(t) / edge_norm.
//
// Two kernels, each keeping the reference's evaluation order where it matters:
//   sub_runs_wave_kernel   phi is the reference's *float* running sum (41 472 serial adds, dphi changes every
//                          256 samples).  It is not walked: phase_runs.h decomposes it exactly into ~200 linear
//                          runs per signal (one wave per job, up to 64 symbols probed at once).
//   sub_fir_fused_kernel   per workgroup of 2 048 outputs: every input sample's phase from its run, glibc-exact
//                          sincos (one shared argument reduction), r and the products s*conj(r) straight into the
//                          LDS tile; each low-pass output is one serial 360-term sum in tap order, a lane owns 8
//                          consecutive outputs and slides an 8-sample register window (tile transposed by 8: the
//                          per-tap read is conflict-free), taps through the scalar cache; the epilogue subtracts
//                          c*r/norm in place with r from LDS.
// Bound: fp32 VALU (64 MFLOP of separately rounded mul/add per job), not HBM.
#include "wspr_device.h"
#include <cstdlib>
#include "glibc_sincosf.h"
#include "phase_runs.h"

#pragma clang fp contract(off)

namespace wspr {
namespace {

constexpr double kTwoPiDt = 2.0 * 3.14159265358979323846 * 1.0 / 375.0;

// scratch per job: the phase runs (PhaseTable)
struct PhaseTable {
    PhaseRun runs[kPhaseMaxRuns];
    float sym_phi[kNSymD];           // phase of the first sample of every symbol
    float dphi[kNSymD];
    uint16_t first_run[kNSymD + 2];  // first_run[0] == 0xffff: the runs did not fit, use sym_phi/dphi
};
static_assert(sizeof(PhaseTable) % 4 == 0, "PhaseTable is addressed in floats");
constexpr size_t kTableFloats = sizeof(PhaseTable) / 4;

__device__ __forceinline__ float dphi_of_symbol(float f0, float drift, int i, unsigned cs) {
    // wsprd.c:343, all double: TWOPIDT*(f0 + (drift/2.0)*(i - 81.0)/81.0 + (cs - 1.5)*375.0/256.0)
    const double arg = (double)f0 + ((double)drift / 2.0) * ((double)(float)i - 81.0) / 81.0
                       + ((double)(float)cs - 1.5) * 375.0 / 256.0;
    return (float)(kTwoPiDt * arg);
}

// The run decomposition of the serial float phase walk, one WAVE per job (phase_runs_build_chained, phase_runs.h):
// up to 64 symbols are probed at once, the leading ones whose 256 additions are all regular become one run each with prefix-summed
// significands, the first that is not (a binade crossing, typically) is walked run by run; ~20 rounds per
// signal instead of ~200-350 serial trips, and a thousand jobs are a thousand waves instead of sixteen
// (0.34-0.43 ms of latency per launch with lane = job, during which the GPU held 16 waves).
__global__ __launch_bounds__(64)
void sub_runs_wave_kernel(const SubJob* __restrict__ jobs, int njobs, PhaseTable* __restrict__ tables) {
    __shared__ float dph[kNSymD];
    const int job = blockIdx.x, lane = threadIdx.x;
    if (job >= njobs) return;
    const SubJob* __restrict__ jb = jobs + job;
    PhaseTable& tb = tables[job];
    const float f0 = jb->f0, drift = jb->drift;
    for (int i = lane; i < kNSymD; i += 64) {
        const float v = dphi_of_symbol(f0, drift, i, jb->sym[i]);
        dph[i] = v;
        tb.dphi[i] = v;
    }
    __syncthreads();
    // everything below is wave-uniform except the per-lane probe of symbol i + lane
    float phi = 0.0f;
    int nr = 0, i = 0;
    bool full = false;
    while (i < kNSymD) {
        int32_t m = 0;
        int e = 0, taken = 0;
        if (phase_split(phi, &m, &e)) {
            const int si = i + lane;
            const bool in = si < kNSymD;
            const PhaseSymbolStep st = phase_symbol_probe(e, (m & 1) != 0, in ? dph[si] : 0.0f);
            const int inc = (in && st.ok) ? kSps * st.q : 0;
            int incl = inc;                                      // inclusive prefix sum over the lanes
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            const int32_t ms = m + (incl - inc);                 // significand at the start of symbol i + lane
            const bool good = in && st.ok && phase_symbol_room(ms, st, kSps);
            const unsigned long long bad = ~__ballot(good);
            taken = bad ? (int)__ffsll((long long)bad) - 1 : 64;
            if (lane < taken) {
                tb.sym_phi[si] = phase_join(ms, e);
                const int slot = nr + lane;
                if (!full && slot < kPhaseMaxRuns) {
                    tb.first_run[si] = (uint16_t)slot;
                    tb.runs[slot] = PhaseRun{si * kSps, ms, st.q, e};
                }
            }
            if (taken) {
                const int32_t mb = taken < 64 ? __shfl(ms, taken) : __shfl(ms + inc, 63);
                phi = phase_join(mb, e);
                if (!full) {
                    if (nr + taken > kPhaseMaxRuns) { nr = kPhaseMaxRuns; full = true; } else nr += taken;
                }
            }
        }
        i += taken;
        if (taken == 64 || i >= kNSymD) continue;
        // symbol i does not pass: walk it run by run (every lane the same walk, lane 0 stores)
        const float d = dph[i];
        if (lane == 0) {
            tb.sym_phi[i] = phi;
            if (!full) tb.first_run[i] = (uint16_t)nr;
        }
        int left = kSps, pos = i * kSps;
        while (left > 0) {
            PhaseRun r;
            const int n = phase_next_run(phi, d, pos, left - 1, r);
            if (nr >= kPhaseMaxRuns) full = true;
            if (!full) {
                if (lane == 0) tb.runs[nr] = r;
                ++nr;
            }
            pos += n + 1;
            left -= n + 1;
        }
        ++i;
    }
    if (lane == 0) {
        tb.first_run[kNSymD] = (uint16_t)nr;
        if (full) tb.first_run[0] = 0xffffu;
    }
}

constexpr int kFirThreads = 256;
// Eight outputs per lane, taps from scalar registers.  (With four outputs per lane and the taps in LDS a wave
// issued 32 packed multiply/adds per four taps against one 16-byte and four 8-byte LDS reads, and with four SIMDs
// on one LDS the return path was ~75 % busy: 1.22 vs 0.95 ms per 1024 jobs.)  A lane slides an eight-sample register window (16 packed instructions per 8-byte LDS read) and the taps, the
// same for every lane, arrive through the scalar cache eight at a time, issued a stage (eight taps = 128
// packed instructions) before they are needed and waited for when a stage old (as the lag scan's table loads).  The
// tile is stored transposed by 8, so the per-tap read of consecutive lanes is consecutive 8-byte words.
// Each output is still one serial 360-term sum in tap order: identical bits.
constexpr int kFir8PerLane = 8;
constexpr int kFir8Out = kFirThreads * kFir8PerLane;              // 2048 outputs per workgroup
constexpr int kFir8Span = kFir8Out + kLpfTaps - 1;                // 2407 inputs
constexpr int kFir8Pitch = (kFir8Span + 7) / 8 + 1;               // transposed-by-8 row pitch (8-byte words)
static_assert(kLpfTaps % 8 == 0, "taps are consumed eight at a time");
constexpr int kFirWgs = (kSigLen + kFir8Out - 1) / kFir8Out;      // 21 tiles per signal
constexpr int kHalo = kLpfTaps / 2;                               // 180 samples of a neighbour's range on either side

// Reference r(t), the products s(t) conj(r(t)) and the FIR in ONE kernel (round 3): neither r nor s conj(r) travels
// through HBM.  Rounds 1-2 had a kernel write both for the whole signal (16 bytes per sample) and the FIR read them
// back: 2 GB per 2 048 jobs at the mixed-traffic ceiling of the memory system (0.545 ms of the set's 3.13 ms).  Here
// the workgroup that filters outputs n0 .. n0 + 2047 forms s conj(r) for the 2 407 inputs it needs while it stages
// them -- per sample: phase from the run table (every sample's phase from its run, phase_runs.h), glibc-exact sincos,
// four products, two adds, exactly the operations of the former kernel -- and keeps r of its own 2 048 outputs in LDS
// for the epilogue.  Set 3.13 -> 2.89 ms per 2 048 jobs, 664 KB of scratch per job gone.
// Staging goes symbol by symbol (thread j = sample j of the symbol), so that a wave sits in ONE symbol and fetches
// its few runs with wave-uniform (scalar) loads.
__global__ __launch_bounds__(kFirThreads)
void sub_fir_fused_kernel(float* __restrict__ dI, float* __restrict__ dQ, int np,
                          const SubJob* __restrict__ jobs, const PhaseTable* __restrict__ tables,
                          const float* __restrict__ lpf, const float* __restrict__ lpf_part, float2* __restrict__ halo,
                          int parity) {
    __shared__ float2 tile[8 * kFir8Pitch];
    __shared__ float2 rref[kFir8Out];
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x;
    // staging (loads, sincos: short dependent chains that want to issue at once) runs at a higher wave priority than the
    // FIR loop, which fills whatever issue slots are left: 2.58 -> 2.50 ms per 2 048 jobs
    __builtin_amdgcn_s_setprio(3);
    const SubJob* job = jobs + blockIdx.y;
    const PhaseTable& tb = tables[blockIdx.y];
    const bool dense = tb.first_run[0] == 0xffffu;
    const int shift = job->shift;
    float* __restrict__ xi = dI + (size_t)job->seg * kIqStride;
    float* __restrict__ xq = dQ + (size_t)job->seg * kIqStride;
    // The subtraction is IN PLACE and a tile reads 180 samples on either side of its own outputs -- samples its two
    // neighbours overwrite, as it overwrites theirs.  The tiles therefore run in two launches: the EVEN tiles first
    // (their halos lie in odd tiles' ranges, which nobody writes in that launch), and while they stage they save
    // s conj(r) of the first and last 180 samples of their own range (2.9 KB per tile); the ODD tiles second: their
    // own range is still untouched, and the halos -- by now overwritten -- come from what the even tiles saved
    // (the same values bit for bit: the same operations on the same samples).  No workgroup ever waits for another
    // (round 3's first form had neighbours exchange flags: correct, and as fast alone, but its forward progress hung
    // on the dispatcher -- with the front end on a CU-masked stream beside it a step took 8.9 s).
    const int tile_idx = 2 * (int)blockIdx.x + parity;
    float2* __restrict__ my_halo = halo + ((size_t)blockIdx.y * kFirWgs + tile_idx) * 2 * kHalo;
    const int n0 = tile_idx * kFir8Out;
    const int n_lo = n0 - kLpfTaps / 2, n_hi = n_lo + kFir8Span;            // inputs [n_lo, n_hi)
    // the reference filters a zero-padded copy (360 leading zeros); outside the signal the products are zero
    if (n_lo < 0 || n_hi > kSigLen)
        for (int e = tid; e < kFir8Span; e += kFirThreads) {
            const int n = n_lo + e;
            if (n < 0 || n >= kSigLen) tile[(e & 7) * kFir8Pitch + (e >> 3)] = make_float2(0.0f, 0.0f);
        }
    // what this tile forms itself: an even tile all its inputs, an odd tile its own range only
    const int c_lo = parity ? n0 : n_lo, c_hi = parity ? min(n0 + kFir8Out, n_hi) : n_hi;
    if (parity) {
        for (int e = tid; e < kFir8Span; e += kFirThreads) {
            const int n = n_lo + e;
            if (n < 0 || n >= kSigLen) continue;
            if (n < n0) tile[(e & 7) * kFir8Pitch + (e >> 3)] = my_halo[-kHalo + (n - n_lo)];          // left neighbour's LAST 180
            else if (n >= n0 + kFir8Out) tile[(e & 7) * kFir8Pitch + (e >> 3)] = my_halo[2 * kHalo + (n - (n0 + kFir8Out))];   // right neighbour's FIRST 180
        }
    }
    const int sym_lo = max(c_lo, 0) >> 8, sym_hi = (min(c_hi, kSigLen) - 1) >> 8;
#pragma unroll 1
    for (int sym = sym_lo; sym <= sym_hi; ++sym) {
        const int n = sym * kSps + tid;
        if (n < c_lo || n >= c_hi) continue;                               // whole waves at a time except at the two ends
        float phi;
        if (!dense) {
            const int r0 = __builtin_amdgcn_readfirstlane((int)tb.first_run[sym]);
            const int r1 = __builtin_amdgcn_readfirstlane((int)tb.first_run[sym + 1]);
            PhaseRun pick = tb.runs[r0];
            for (int r = r0 + 1; r < r1; ++r) {
                const PhaseRun c = tb.runs[r];
                if (c.start <= n) pick = c;
            }
            phi = phase_of(pick, n - pick.start);
        } else {
            phi = phase_from_symbol(tb.sym_phi[sym], tb.dphi[sym], tid);
        }
        float sr, cr;
        glibc_sincosf_pair(phi, &sr, &cr);
        const int k = shift + n;
        float a = 0.0f, b = 0.0f;
        if (k > 0 && k < np) {
            const float x = xi[k], y = xq[k];
            const float p1 = x * cr, p2 = y * sr, p3 = y * cr, p4 = x * sr;
            a = p1 + p2;                  // Re{s conj(r)}
            b = p3 - p4;                  // Im{s conj(r)}
        }
        const int e = n - n_lo;
        tile[(e & 7) * kFir8Pitch + (e >> 3)] = make_float2(a, b);
        if (n >= n0 && n < n0 + kFir8Out) {
            rref[n - n0] = make_float2(cr, sr);
            if (!parity) {                                                 // what the odd neighbours will need
                if (n - n0 < kHalo) my_halo[n - n0] = make_float2(a, b);
                else if (n - n0 >= kFir8Out - kHalo) my_halo[kHalo + (n - n0) - (kFir8Out - kHalo)] = make_float2(a, b);
            }
        }
    }
    __syncthreads();
    // the last tile reaches past the end of the signal: a wave without outputs has nothing to filter
    if (n0 + 8 * (tid & ~63) >= kSigLen) return;
    // outputs n0 + 8 tid + r, r = 0..7; input of tap j for output r: e = 8 tid + r + j.
    // The eight-sample window lives in THREE register sets that take turns (no register is ever copied): during
    // stage k (taps 8k .. 8k + 7) set k % 3 holds the window as the stage found it, set (k + 1) % 3 the eight
    // samples that enter under these taps (at tap u the window is in[0..u-1] | old[u..7]), and set (k + 2) % 3
    // receives the samples of stage k + 1, issued before stage k is consumed.
    v2f acc[8], W[3][8];
    float4 Ta[3], Tb[3];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        acc[r] = (v2f){0.0f, 0.0f};
        const float2 t = tile[r * kFir8Pitch + tid];
        W[0][r] = (v2f){t.x, t.y};
    }
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(lpf);
    auto issue = [&](v2f (&in)[8], float4& wa, float4& wb, int k) {   // taps 8k .. 8k + 7 and the samples that enter under them
        const int kk = 8 * k < kLpfTaps ? k : 0;             // clamped past the end (values unused)
        wa = w4[2 * kk];
        wb = w4[2 * kk + 1];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float2 t = tile[u * kFir8Pitch + tid + 1 + kk];
            in[u] = (v2f){t.x, t.y};
        }
    };
    auto consume = [&](const v2f (&old)[8], const v2f (&in)[8], const float4 wa, const float4 wb) {
        const float w[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const v2f wu = {w[u], w[u]};
            v2f p[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = (u + r) & 7;
                p[r] = wu * (i < u ? in[i] : old[i]);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = acc[r] + p[r];
        }
    };
    __builtin_amdgcn_s_setprio(0);
    issue(W[1], Ta[0], Tb[0], 0);
    static_assert(kLpfTaps % 24 == 0, "three stages of eight taps per trip");
#pragma unroll 1
    for (int k0 = 0; k0 < kLpfTaps / 8; k0 += 3) {
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): this stage's taps and samples have landed
            __builtin_amdgcn_sched_barrier(0);
            issue(W[(s3 + 2) % 3], Ta[(s3 + 1) % 3], Tb[(s3 + 1) % 3], k0 + s3 + 1);
            __builtin_amdgcn_sched_barrier(0);
            consume(W[s3], W[(s3 + 1) % 3], Ta[s3], Tb[s3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // wsprd.c:397-404: only the first and last 180 outputs of the signal are normalised by a partial tap sum; for
    // all the others norm == 1 and x / 1.0f == x exactly, so interior tiles do not divide
    const bool edge_tile = n0 < kLpfTaps / 2 || n0 + kFir8Out > kSigLen - kLpfTaps / 2;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = n0 + 8 * tid + r;
        if (n >= kSigLen) break;
        const int k = shift + n;
        if (k > 0 && k < np) {
            const float2 rr = rref[8 * tid + r];
            const float si = acc[r].x, sq = acc[r].y;
            const float a = si * rr.x, b = sq * rr.y, c = si * rr.y, d = sq * rr.x;
            float ri = a - b, rq = c + d;
            if (edge_tile) {
                float norm = 1.0f;
                if (n < kLpfTaps / 2)                    norm = lpf_part[kLpfTaps / 2 + n];
                else if (n > kSigLen - 1 - kLpfTaps / 2) norm = lpf_part[kLpfTaps / 2 + kSigLen - 1 - n];
                ri = ri / norm;
                rq = rq / norm;
            }
            xi[k] = xi[k] - ri;
            xq[k] = xq[k] - rq;
        }
    }
}

// ---- receiver normalisation, rtlsdr_wsprd.c:284-305 -------------------------
__global__ __launch_bounds__(1024)
void normalise_kernel(float* __restrict__ dI, float* __restrict__ dQ, const int* __restrict__ n_valid,
                      int n_total) {
    __shared__ float red[1024];
    const int seg = blockIdx.x, tid = threadIdx.x;
    float* __restrict__ xi = dI + (size_t)seg * kIqStride;
    float* __restrict__ xq = dQ + (size_t)seg * kIqStride;
    const int nv = n_valid ? n_valid[seg] : n_total;
    float peak = 1e-24f;
    for (int i = tid; i < n_total; i += 1024) {
        float a = xi[i], b = xq[i];
        if (i >= nv) { a = 0.0f; b = 0.0f; xi[i] = 0.0f; xq[i] = 0.0f; }
        peak = fmaxf(peak, fmaxf(fabsf(a), fabsf(b)));       // max is order-independent
    }
    red[tid] = peak;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    const float scale = (float)(0.5 / (double)red[0]);
    for (int i = tid; i < n_total; i += 1024) { xi[i] = xi[i] * scale; xq[i] = xq[i] * scale; }
}

// ---- subtract_signal(), wsprd/wsprd.c:263-312 (symbol-by-symbol subtraction) --------------------------------
// Exported by the reference's header (wsprd.h:83-89) but never called by its decoder.  The 162 symbols touch
// disjoint 256-sample spans, so a workgroup owns one symbol: the span is staged, thread 0 walks the reference's
// two serial loops (phasor recurrence, then the correlation sums in sample order), all threads subtract.
__global__ __launch_bounds__(256)
void sub_symbolwise_kernel(float* __restrict__ xi, float* __restrict__ xq, int np, float f0, int shift, float drift,
                           const unsigned char* __restrict__ sym) {
    __shared__ float c0[kSps], s0[kSps], si[kSps], sq[kSps], amp[2];
    const int i = blockIdx.x, j = threadIdx.x;
    const int k = shift + i * kSps + j;
    const bool in = (k > 0) && (k < np);
    si[j] = in ? xi[k] : 0.0f;
    sq[j] = in ? xq[k] : 0.0f;
    __syncthreads();
    if (j == 0) {
        // float fp = f0 + ((float)drift / 2.0) * ((float)i - (float)NBITS) / (float)NBITS;
        const float fp = (float)((double)f0 + ((double)drift / 2.0) * (double)((float)i - 81.0f) / (double)81.0f);
        // float dphi = TWOPIDT * (fp + ((float)cs - 1.5) * DF), the macros expanded in place (all double)
        const float dphi = (float)(kTwoPiDt * ((double)fp + ((double)(float)sym[i] - 1.5) * 375.0 / 256.0));
        float sd, cd;
        glibc_sincosf_pair(dphi, &sd, &cd);
        float c = 1.0f, s = 0.0f;
        c0[0] = c; s0[0] = s;
        for (int t = 1; t < kSps; ++t) {
            const float cn = c * cd - s * sd, sn = c * sd + s * cd;
            c = cn; s = sn;
            c0[t] = c; s0[t] = s;
        }
        float a = 0.0f, b = 0.0f;
        for (int t = 0; t < kSps; ++t) {
            const int kk = shift + i * kSps + t;
            if ((kk > 0) && (kk < np)) {
                a = a + si[t] * c0[t] + sq[t] * s0[t];
                b = b - si[t] * s0[t] + sq[t] * c0[t];
            }
        }
        amp[0] = a / (float)kSps;
        amp[1] = b / (float)kSps;
    }
    __syncthreads();
    if (in) {
        const float a = amp[0], b = amp[1];
        xi[k] = si[j] - (a * c0[j] - b * s0[j]);
        xq[k] = sq[j] - (b * c0[j] + a * s0[j]);
    }
}
}  // namespace

void launch_subtract_symbolwise(float* dI, float* dQ, int samples, float f0, int shift, float drift,
                                const unsigned char* d_sym, hipStream_t st) {
    hipLaunchKernelGGL(sub_symbolwise_kernel, dim3(kNSymD), dim3(kSps), 0, st, dI, dQ, samples, f0, shift, drift, d_sym);
}

// scratch floats needed for njobs jobs: the phase-run tables + the halos the even tiles save for the odd ones
size_t subtract_scratch_floats(int njobs) { return (size_t)njobs * (kTableFloats + (size_t)kFirWgs * 2 * kHalo * 2); }

void launch_subtract(float* dI, float* dQ, int samples, const SubJob* jobs, int njobs,
                     float* scratch, const DeviceTables& t, hipStream_t st) {
    if (njobs <= 0) return;
    PhaseTable* tables = reinterpret_cast<PhaseTable*>(scratch);
    float2* halo = reinterpret_cast<float2*>(scratch + (size_t)njobs * kTableFloats);
    hipLaunchKernelGGL(sub_runs_wave_kernel, dim3(njobs), dim3(64), 0, st, jobs, njobs, tables);
    hipLaunchKernelGGL(sub_fir_fused_kernel, dim3((kFirWgs + 1) / 2, njobs), dim3(kFirThreads), 0, st,
                       dI, dQ, samples, jobs, tables, t.lpf, t.lpf_part, halo, 0);
    hipLaunchKernelGGL(sub_fir_fused_kernel, dim3(kFirWgs / 2, njobs), dim3(kFirThreads), 0, st,
                       dI, dQ, samples, jobs, tables, t.lpf, t.lpf_part, halo, 1);
}

// Working copy of resident input: rows of `samples` floats (16-byte aligned, stride a multiple of 4) into
// rows of kIqStride floats with a zero tail, I and Q in one launch of 16-byte accesses.
namespace {
__global__ __launch_bounds__(256)
void load_rows_kernel(const float4* __restrict__ sI, const float4* __restrict__ sQ, size_t src_row4, int n4,
                      float4* __restrict__ dI, float4* __restrict__ dQ) {
    const int seg = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    constexpr int kRow4 = kIqStride / 4;
    if (e >= kRow4) return;
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    dI[(size_t)seg * kRow4 + e] = e < n4 ? sI[(size_t)seg * src_row4 + e] : z;
    dQ[(size_t)seg * kRow4 + e] = e < n4 ? sQ[(size_t)seg * src_row4 + e] : z;
}
}  // namespace

bool launch_load_rows(const float* sI, const float* sQ, size_t stride, int samples, int nseg, float* dI, float* dQ,
                      hipStream_t st) {
    if ((samples & 3) || (stride & 3) || (reinterpret_cast<uintptr_t>(sI) & 15) || (reinterpret_cast<uintptr_t>(sQ) & 15))
        return false;                                        // the caller falls back to strided copies
    if (nseg <= 0) return true;
    hipLaunchKernelGGL(load_rows_kernel, dim3((kIqStride / 4 + 255) / 256, nseg), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(sI), reinterpret_cast<const float4*>(sQ), stride / 4, samples / 4,
                       reinterpret_cast<float4*>(dI), reinterpret_cast<float4*>(dQ));
    return true;
}

void launch_normalise(float* dI, float* dQ, const int* n_valid, int nseg, int n_total, hipStream_t st) {
    if (nseg <= 0) return;
    hipLaunchKernelGGL(normalise_kernel, dim3(nseg), dim3(1024), 0, st, dI, dQ, n_valid, n_total);
}

}  // namespace wspr

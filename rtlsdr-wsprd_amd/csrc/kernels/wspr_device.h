// Device-side data layout and kernel launchers of the MI355X WSPR decoder.
//
// HBM layout (all float32 unless noted), sized for thousands of resident
// 2-minute segments (one segment = 45 000 complex samples at 375 sps):
//   iq      I[nseg][kIqStride], Q[nseg][kIqStride]   planar, rows 256-B aligned
//   ps      [nseg][417 bins][kPsTPitch]               |X|^2, bin-major like the reference's
//                                                     ps[512][blocks]: row b = fft-shifted bin 48+b,
//                                                     its time blocks contiguous (pitch 352 floats)
//   cand    DevCand[nseg][200] + npk[nseg]            peak list, strongest first
// Everything a kernel needs besides these is a small constant table uploaded once
// (window, twiddles, sync vector, subtraction low-pass taps).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>

namespace wspr {

// Kernel-selection and measurement switches exist in the LAB build only (libwspr_mi355x_lab.so, -DWSPR_LAB: what the
// parity suite and bench.py's kernel-level timings load); in the product library every one of them reads as unset, so
// the product carries no way to pick a different kernel, repeat a stage or fold virtual devices.
#ifdef WSPR_LAB
inline const char* lab_env(const char* name) { return getenv(name); }
#else
inline const char* lab_env(const char*) { return nullptr; }
#endif

constexpr int kMaxSamples = 45000;
constexpr int kIqStride   = 45056;      // 176 * 256
constexpr int kFftSize    = 512;
constexpr int kHop        = 128;
constexpr int kMaxBlocks  = 347;        // 4*floor(45000/512) - 1
constexpr int kPsBin0     = 48;         // first fft-shifted bin kept
constexpr int kPsBins     = 417;        // bins 48..464 (all that any consumer reads)
constexpr int kPsStride   = 432;        // floats per segment of the time-averaged spectrum (psavg)
constexpr int kPsTPitch   = 352;        // floats per (segment, bin) row of the spectrogram: 347 time blocks + padding
                                        // (1408 bytes = 11 x 128: every row starts on a 128-byte line)
constexpr int kSmooth     = 411;        // smoothed-spectrum length
constexpr int kMaxCand    = 200;
constexpr int kNSymD      = 162;
constexpr int kNBitsD     = 81;
constexpr int kSps        = 256;
constexpr int kSigLen     = kNSymD * kSps;   // 41 472 samples of signal
constexpr int kLpfTaps    = 360;
constexpr int kMaxLags    = 43;

struct DevCand {
    float freq;     // Hz relative to 1500 Hz
    float snr;      // dB in 2500 Hz (device log10f)
    float peak;     // normalised smoothed-spectrum value the snr derives from
    int   shift;    // samples
    float drift;    // Hz over the frame
    float sync;
    int   bin;      // index of the peak in the smoothed spectrum (the reference's list order before its sort)
};

// One candidate being refined/demodulated (fine sync, soft symbols)
struct FineState {
    int   seg;
    float freq;         // in: coarse freq  -> after mode 1: refined freq
    float drift;
    int   shift;        // in: coarse shift -> after mode 0: refined shift
    float sync;         // sync after the latest mode-0/1 search
    int   shift_coarse; // kept so that the lag window can be rebuilt
    float freq_coarse;
    int   pad;
};

// One coherent subtraction job
struct SubJob {
    int   seg;
    float f0;
    int   shift;
    float drift;
    unsigned char sym[kNSymD];
    unsigned char pad[2];
};

struct DeviceTables {
    const float*  window;      // [512]  sinf(0.006147931*j)
    const float2* twiddle;     // [256]  exp(-2*pi*i*k/512), float
    const unsigned char* sync; // [162]
    const float*  lpf;         // [360]  normalised sine taps
    const float*  lpf_part;    // [360]  running sums for the edge correction
    float min_snr;             // powf(10, -0.8)
    float floor_snr;           // 0.1 * min_snr
};

// ---- launchers (all asynchronous on `st`) ----------------------------------
void launch_fft_bank(const float* dI, const float* dQ, const int* seg_list, int nseg_active,
                     int samples, float* ps, const DeviceTables& t, hipStream_t st);
// K1 fused with the time average (one workgroup per segment): ps as above, psavg[seg][kPsStride]
void launch_fft_bank_avg(const float* dI, const float* dQ, const int* seg_list, int nseg_active,
                         int samples, float* ps, float* psavg, const DeviceTables& t, hipStream_t st);
void launch_calib_copy(const float* src, float* dst, size_t n, hipStream_t st);
void launch_calib_copy16(const float* src, float* dst, size_t n, hipStream_t st, int variant = 0);

// Opt a kernel in to `bytes` of dynamic LDS on the CURRENT device, once per device (a process that drives
// several devices through wspr_set_device reaches every launch site from each of them).
inline void lds_opt_in(const void* kernel, size_t bytes, std::atomic<unsigned>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 31) dev = 0;
    if (done.load(std::memory_order_relaxed) & (1u << dev)) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    done.fetch_or(1u << dev, std::memory_order_relaxed);
}
double launch_calib_valu(float* out, int iters, hipStream_t st);
void launch_calib_read(const uint8_t* raw, size_t bytes_per_seg, int nseg, unsigned* out, hipStream_t st);
// psavg: scratch, nseg * kPsStride floats (time-averaged spectrum per segment)
// K2a alone: psavg[seg][kPsStride] = sum over time blocks of ps, in block order (wsprd.c:556-561)
void launch_time_average(const float* ps, const int* seg_list, int nseg_active, int blocks, float* psavg,
                         hipStream_t st);
// have_avg: psavg was already filled by launch_time_average
void launch_pick_peaks(const float* ps, const int* seg_list, int nseg_active, int blocks, float* psavg,
                       DevCand* cand, int* npk, float* noise_out, float* smspec_out,
                       const DeviceTables& t, hipStream_t st, bool have_avg = false);
void launch_coarse_sync(const float* ps, const int* seg_list, int nseg_active, int blocks,
                        DevCand* cand, const int* npk, int maxdrift,
                        const DeviceTables& t, hipStream_t st);
// nhyp hypotheses per item, results in sync_out[item][nhyp]:
// mode 0: lags shift_coarse-128 + lagstep*h at freq_coarse
// mode 1: frequencies state.freq + (ifmin+h)*fstep at state.shift
// mode 2: lags state.shift + jitter[h] at state.freq; also soft symbols and their rms;
//         skipped (nothing written) for items whose state.sync <= minsync1
void launch_demod(const float* dI, const float* dQ, int samples, const FineState* items, int nitems,
                  int mode, int nhyp, int lagstep, int ifmin, float fstep, const int* jitter,
                  float minsync1, float* sync_out, unsigned char* sym_out, float* rms_out,
                  const DeviceTables& t, hipStream_t st, int symfac = 50);
// Tiled fast path for the wide searches.  FineState.pad must hold the index of the item's
// first phasor table in `tabs` (1 table if drift == 0, else 162); list_shared/list_own are the
// item indices without / with drift.  mode 0: nlag lags shift_coarse-128 + lagstep*m;
// mode 2: 43 lags shift-63+3*m (lagstep must be 3).  pw: nitems*nlag*162 float4 of scratch.
void launch_phasor_tables(const FineState* items, int nitems, int mode, float* tabs, hipStream_t st);
void launch_demod_tiled(const float* dI, const float* dQ, int samples, const FineState* items, int nitems,
                        const int* list_shared, int n_shared, const int* list_own, int n_own, int mode,
                        int nlag, int lagstep, float minsync1, const float* tabs, float* pw,
                        float* sync_out, unsigned char* sym_out, float* rms_out,
                        const DeviceTables& t, hipStream_t st);
// mode 1 (5 frequencies) + first ladder rung; see k4_demod.hip.  tabs: n_shared*5 tables,
// pw: n_shared*5*162 float4, scratch_sync: nitems*5 floats.
void launch_freq_scan_and_first_rung(const float* dI, const float* dQ, int samples, FineState* items,
                                     const int* list_shared, int n_shared, const int* list_own, int n_own,
                                     int lagstep, float minsync1, const int* jitter0, float* tabs, float* pw,
                                     float* scratch_sync, float* sync_out, unsigned char* sym_out,
                                     float* rms_out, const DeviceTables& t, hipStream_t st,
                                     const float* pw_lag = nullptr, int nlag_lag = 0);
void launch_pick_lag(FineState* items, int nitems, const float* sync_in, int nlag, int lagstep, hipStream_t st);
void launch_pick_freq(FineState* items, int nitems, const float* sync_in, int nfreq, int ifmin,
                      float fstep, hipStream_t st);
size_t subtract_scratch_floats(int njobs);
void launch_subtract(float* dI, float* dQ, int samples, const SubJob* jobs, int njobs,
                     float* scratch /* subtract_scratch_floats(njobs) */, const DeviceTables& t, hipStream_t st);
// subtract_signal() of the reference (wsprd.c:263-312): one segment row, symbols in device memory
void launch_subtract_symbolwise(float* dI, float* dQ, int samples, float f0, int shift, float drift,
                                const unsigned char* d_sym, hipStream_t st);
// Device Fano search (K6) for n soft-symbol vectors symbols[offsets[i]*162 ...] (interleaved order, as the
// demodulator writes them); metric0 = the 256-entry "sent 0" branch-metric row.
// One wavefront per vector, 64 tree visits per step (k6_fano_wave.hip): exact return
// code, cycle count and decoded bytes; metric/maxnp only for decoded frames.  ret -2 = the wave's
// pending-visit store overflowed (the caller decodes that vector some other way).  steps may be null.
// scratch: fano_wave_scratch_words(n) words of device memory (the waves' stack slices; their tops live in LDS).
size_t fano_wave_scratch_words(int n);
void launch_fano_wave(const unsigned char* symbols, const int* offsets, int n, const short* metric0,
                      unsigned maxcycles, int* ret, unsigned* cycles, unsigned* metric, unsigned* maxnp,
                      unsigned char* data, unsigned* steps, uint32_t* scratch, hipStream_t st);
void launch_normalise(float* dI, float* dQ, const int* n_valid, int nseg, int n_total, hipStream_t st);
// resident input rows -> working rows (zero tail); false if the input is not 16-byte friendly
bool launch_load_rows(const float* sI, const float* sQ, size_t stride, int samples, int nseg, float* dI, float* dQ,
                      hipStream_t st);
// Front-end state of one receiver between chunks of its sample stream (all zero at start-up):
// samples into the open decimation block, both integrators per rail, and the integrator values at
// the last 36 decimation instants (what the two combs and the 33-tap FIR still need).
constexpr int kDecimHist = 36;
struct DecimState {
    uint32_t phase;
    uint32_t x1[2], x2[2];
    uint32_t hist[2][kDecimHist];
};
int decimate_blocks(size_t nsamp, bool stateful);
void front_end_constants(float* taps33, int* samples_per_output);   // the FIR taps and R as the kernels use them
// states: null = zero state, whole blocks only; else one DecimState per segment row, read and updated
void launch_decimate(const uint8_t* raw, size_t bytes_per_seg, int nseg, float* dI, float* dQ,
                     int* n_out, int32_t* scratch, hipStream_t st, DecimState* states = nullptr);

}  // namespace wspr

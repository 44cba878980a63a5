/* ============================================================================
 * wspr_mi355x_bench.h -- the LAB side of the library: stage-level parity hooks, the per-candidate trace of the fine
 * search, kernel timings and calibration kernels.  None of this is in libwspr_mi355x.so (the drop-in for wsprd.h:
 * include/wspr_mi355x.h is all it exports).  These entry points are compiled under -DWSPR_LAB into
 * libwspr_mi355x_lab.so -- the same sources, kernels and scheduler plus what is declared here, plus the environment
 * switches that select alternative kernels or repeat stages (WSPR_K0_KERNEL, WSPR_K0_RESIDENT, WSPR_K0_CUS,
 * WSPR_K1_FUSED, WSPR_K3_KERNEL, WSPR_K4_LAG, WSPR_K4_FREQ, WSPR_K4_DRIFT, WSPR_REPEAT_LAG / _FREQ / _FANO,
 * WSPR_FANO_WAVE_CAP, WSPR_NODE_VIRTUAL; docs/HISTORY.md section 4 "Switches"), which read as unset in the product.
 * tests/ and bench.py load the lab library for these calls and the product library for everything else.
 * ==========================================================================*/
#ifndef WSPR_MI355X_BENCH_H
#define WSPR_MI355X_BENCH_H

#include "wspr_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Per-candidate trace of the fine search (the reference's candidate loop, wsprd.c:697-822): what the production
 * kernels -- lag scan, frequency scan fused with rung 0, the 43-lag ladder block, the Fano search -- produced for
 * EVERY candidate the loop entered, whether it decoded or not.  Same launches as wspr_decode_batch(); the only
 * differences are extra device-to-host copies and that the Fano budget split is off (every attempt runs the
 * reference's full budget where it is first met).  For parity tests against the oracle's trace. */
#define WSPR_TRACE_PASSES 3
typedef struct wspr_cand_trace {
    int   visited;              /* the loop entered this candidate (it may have left the segment early, :786-793) */
    int   mode0_shift;          /* after sync_and_demodulate(mode 0), wsprd.c:709-719 */
    float mode0_sync;
    float freq;                 /* after mode 1, :721-726: what the reference writes back to candidates[j] */
    int   shift;
    float drift;
    float sync;
    int   attempts;             /* mode-2 calls of the jitter ladder, :739-766 (0: sync <= minsync1) */
    int   fano_calls;           /* of which passed the sync/rms gates (:759) and reached fano() */
    float first_sync;           /* rung 0 of the ladder: sync, rms and the 162 soft symbols (transmission order) */
    float first_rms;
    int   decoded;
    int   subtracted;
    int   jitter;
    unsigned cycles;
    unsigned char first_symbols[162];
    unsigned char decdata[11];
    unsigned char pad[3];
} wspr_cand_trace;
typedef struct wspr_trace {
    int passes_run;
    int npk[WSPR_TRACE_PASSES];
    int n_visited[WSPR_TRACE_PASSES];
    wspr_cand_trace cand[WSPR_TRACE_PASSES][MAX_CANDIDATES];
} wspr_trace;
/* wspr_decode_batch() (host buffers, inputs untouched) that also fills trace[0..nseg). */
int wspr_decode_batch_trace(float *idat, float *qdat, int nseg, int samples, size_t seg_stride,
                            struct decoder_options options, struct decoder_results *decodes,
                            int max_results, int *n_results, wspr_trace *trace);


/* FFT bank + power spectrogram (reference wsprd/wsprd.c:509-553) for nseg host
 * segments; ps_out[s][bin 0..511][t 0..blocks) in the reference's bin-major
 * layout, bins outside 48..464 are zero. */
int wspr_stage_fft_bank(const float *idat, const float *qdat, int nseg, int samples,
                        size_t seg_stride, float *ps_out);
/* Peak picker + coarse sync (reference wsprd/wsprd.c:555-678) for nseg host
 * segments: cand_out[s][200] strongest first (after coarse sync when coarse != 0),
 * npk_out[s], noise_out[s] (may be NULL), smspec_out[s][411] (may be NULL). */
int wspr_stage_candidates(const float *idat, const float *qdat, int nseg, int samples,
                          size_t seg_stride, int coarse, int maxdrift, struct cand *cand_out,
                          int *npk_out, float *noise_out, float *smspec_out);

/* Times `iters` passes of the FFT+sync stage (K1,K2,K3) on resident data with HIP events on the
 * launch stream, after one untimed pass.  ms must hold 8 doubles: ms[0] = K1 (sum over the segment chunks of a pass),
 * ms[1] = K2 (time average of every chunk + peak picking), ms[2] = K3, ms[3] = K1 launches per
 * pass, ms[4] = wall time of one pass (first launch to last kernel end), all in milliseconds. */
int wspr_bench_fft_sync(const void *d_idat, const void *d_qdat, int nseg, int samples,
                        size_t seg_stride, int iters, double *ms);
/* Times the two fp32-VALU-bound stages on resident data with HIP events on the launch stream: the
 * strongest candidate of every segment through the tiled lag scan (K4 mode 0, reference wsprd.c:709-719)
 * and the fused frequency scan + first ladder rung (wsprd.c:721-758), and one coherent subtraction
 * (K7, wsprd.c:316-413) per segment.  ms must hold 8 doubles: ms[0] = lag scan, ms[1] = subtraction,
 * ms[2] = candidates, ms[3] = subtraction jobs, ms[4] = frequency scan + first rung (milliseconds per
 * launch set, averaged over `iters` passes after one untimed pass, as wspr_bench_fft_sync does). */
int wspr_bench_valu(const void *d_idat, const void *d_qdat, int nseg, int samples, size_t seg_stride,
                    int iters, double *ms);

/* CUs the whole-segment front end (K0) may occupy: 0 = all (default; env WSPR_K0_CUS), else its kernels run on a
 * stream restricted to that many CUs (hipExtStreamCreateWithCUMask, spread over the XCDs), so that a decoder running
 * on another lane keeps the rest of the chip: K0 is HBM-bound and needs bandwidth, not every CU.  Returns the
 * previous value.  Results never depend on it. */
int wspr_set_front_end_cus(int ncus);
/* Times `iters` launches of the front end (K0 + normalise) on resident raw data with HIP events;
 * ms[0] = average milliseconds per launch. */
int wspr_bench_decimate(const void *d_raw, size_t bytes_per_seg, int nseg, void *d_idat, void *d_qdat,
                        int iters, double *ms);
/* Read-bandwidth calibration for that roofline: `iters` launches of a kernel with K0's access pattern
 * (one workgroup per pair of CIC blocks, 16-byte non-temporal loads) and no arithmetic; ms[0] = average
 * milliseconds per launch over the same resident raw rows. */
int wspr_calib_read(const void *d_raw, size_t bytes_per_seg, int nseg, int iters, double *ms);
/* PMC calibration: `iters` launches of a 4-byte-per-lane stream copy of nfloats floats on the
 * library's stream (known traffic: 4*nfloats bytes read and written per launch). */
int wspr_calib_copy(const void *d_src, void *d_dst, size_t nfloats, int iters);
/* The tuned copy: 16 bytes per lane, four loads in flight per lane, resident grid (nfloats a multiple of 4, 16-byte
 * aligned buffers).  variant: 0 = non-temporal loads and stores, 1 = default loads + non-temporal stores, 2 = default
 * both, 3 = write only (d_dst is filled with 1.0f, d_src is not read).  ms (may be NULL) = average milliseconds per launch, HIP events on the launch stream. */
int wspr_calib_copy16(const void *d_src, void *d_dst, size_t nfloats, int iters, int variant, double *ms);
/* Vector-pipe calibration for the VALU rooflines: `launches` launches of register-only chains of separately
 * rounded packed multiplies and adds (v_pk_mul_f32 + v_pk_add_f32, no FMA) that fill every SIMD; *tflops =
 * sustained TFLOP/s (one flop per multiply or add), i.e. the practical ceiling of the decoder's arithmetic at the
 * clock the GPU holds under that load (theoretical: half of the fp32 FMA vector peak). */
int wspr_calib_valu(int launches, double *tflops);

#ifdef __cplusplus
}
#endif
#endif /* WSPR_MI355X_BENCH_H */

/* ============================================================================
 * wspr_mi355x.h -- C ABI of libwspr_mi355x.so, the MI355X-native WSPR decoder.
 *
 * Drop-in boundary for Guenael/rtlsdr-wsprd v0.5.6: the library exports the
 * reference decoder's own entry point and spot struct, so rtlsdr_wsprd.c can be
 * linked against it instead of wsprd/wsprd.c.  Plain C
 * types only; device pointers are passed as void*.
 *
 * This header is EVERYTHING libwspr_mi355x.so exports: the reference's symbols, their batched / resident / sharded
 * forms, the front end, the receiver session, device selection and a few observability calls.  Stage-level parity
 * hooks, the per-candidate trace, kernel timings and calibration kernels live in wspr_mi355x_bench.h and exist only
 * in the lab build (libwspr_mi355x_lab.so, the same sources with -DWSPR_LAB), which is what the test suite and
 * bench.py's kernel-level measurements load next to the product.
 *
 * Runtime knobs of the product (environment, read once): WSPR_HOST_THREADS (CPUs this process may count on: a
 * rank's share), WSPR_SLOTS (pipelines per call, default 3), WSPR_BLOCKING_SYNC (how host threads wait: 2 = poll and
 * sleep, default; 1 = runtime blocking events; 0 = spin), WSPR_FANO_DEVICE and WSPR_FANO_FAST (initial values of
 * wspr_set_fano_device_mode() / wspr_set_fano_fast_budget()).  There are no others.
 *
 * Every declaration cites the reference interface it replaces.
 * ==========================================================================*/
#ifndef WSPR_MI355X_H
#define WSPR_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants the caller uses (reference wsprd/wsprd.h:36-41) ------------ */
#define HASHTAB_SIZE       32768
#define HASHTAB_ENTRY_LEN  13
#define LOCTAB_ENTRY_LEN   5
#define FFT_SIZE           512
#define MAX_CANDIDATES     200
#define MAX_UNIQUES        100

/* ---- structs, identical layout (reference wsprd/wsprd.h:44-74) ------------ */
struct decoder_options {            /* 40 bytes, passed BY VALUE */
    int  freq;                      /* dial frequency, Hz */
    char rcall[13];
    char rloc[7];
    int  quickmode;
    int  usehashtable;
    int  npasses;
    int  subtraction;
};

struct cand {                       /* 20 bytes */
    float freq;
    float snr;
    int   shift;
    float drift;
    float sync;
};

struct decoder_results {            /* 80 bytes: the spot record */
    double freq;
    float  sync;
    float  snr;
    float  dt;
    float  drift;
    int    jitter;
    char   message[23];
    char   call[13];
    char   loc[7];
    char   pwr[3];
    int    cycles;
};

/* ---- THE BOUNDARY --------------------------------------------------------- */
/* Replaces wspr_decode(), reference wsprd/wsprd.h:106-111 / wsprd/wsprd.c:416.
 * Same contract: idat/qdat (length `samples`, <= 45000) are overwritten with the
 * residual after coherent subtraction, decodes[0..*n_results) is filled strongest
 * first, returns 0.
 *
 * DIFFERENCE FROM THE REFERENCE -- CHECK THE RETURN VALUE.  The reference's wspr_decode() cannot fail and always
 * returns 0 (wsprd.c:854).  This one returns a NEGATIVE value, with a message on stderr and *n_results = 0, when
 *   -1  no HIP device is usable or a HIP call failed (there is no CPU fallback), or
 *   -2  samples > 45000: the reference sizes its FFT bank from `samples` (wsprd.c:516) and would read beyond the
 *       45 000 samples its callers hold; this library's working rows are 45 000 samples and a longer record is
 *       refused rather than silently cut (rounds 1-3 cut it).
 * The same codes come back from every wspr_decode_batch*() entry point.
 *
 * One call = a batch of one segment on the GPU.  With options.usehashtable the
 * reference's hashtable.txt side effect is kept (read before, written after the decode, wsprd.c:481-494,
 * 842-852); fftw_wisdom.dat is not (there is no FFTW). */
int wspr_decode(float *idat, float *qdat, int samples, struct decoder_options options,
                struct decoder_results *decodes, int *n_results);

/* Batched form of the same call (build-defined): nseg independent
 * segments, planar host buffers idat/qdat[s*seg_stride + i]; results for segment s
 * go to decodes[s*max_results ...], n_results[s]: a segment's unique spots are ranked by SNR first and
 * the strongest max_results are returned (the reference allows 100, its caller holds 50).  Inputs are not
 * modified unless writeback != 0.  With options.usehashtable the hash memory orders the segments
 * (wsprd.c:481-494, 842-852): the result -- spots and hashtable.txt -- is that of nseg reference calls in index
 * order, obtained in parallel (see wspr_decode_batch_hashed() below; rounds 2-4 decoded such a batch one segment at a
 * time).  Calls with the option are ordered as they enter the library (each reads the file the previous one wrote);
 * a call decodes its first round ahead of its turn, so several may be in flight on different lanes. */
int wspr_decode_batch(float *idat, float *qdat, int nseg, int samples, size_t seg_stride,
                      struct decoder_options options, struct decoder_results *decodes,
                      int max_results, int *n_results, int writeback);

/* usehashtable on a batch, the general form.  The reference reads hashtable.txt before and writes it after
 * every decode (wsprd.c:481-494, 842-852), so what a type-3 "<call>" message resolves to (wsprd_utils.c:296-300)
 * depends on the ORDER of the segments.  wspr_decode_batch*() with options.usehashtable decodes the batch in parallel all
 * the same: each segment logs its hash stores and look-ups, the logs are checked in index order, and only the segments
 * whose look-ups would have seen something else are decoded again, to the fixed point -- same spots and same
 * hashtable.txt as nseg reference calls in index order.  This entry adds what a job sharded over several processes /
 * GPUs needs: the call's segments are global indices seg_index0 .. seg_index0 + nseg - 1, `prior` holds the stores of
 * the other shards (any order; those of earlier segments take part), and the call's own stores come back in
 * stores_out[0 .. *n_stores) (segment order).  cap too small: -3 with *n_stores = the capacity needed, nothing written to
 * hashtable.txt, the result arrays filled; the same call again with WSPR_HASH_REVISIT, the same prior and a buffer of
 * that size completes it (nothing is decoded twice).  *n_redecoded: segments decoded more than once.
 * Protocol for R shards (rtlsdr-wsprd_amd/dist.py decode_hashed_sharded): every shard calls with no prior and
 * WSPR_HASH_KEEP_FILE; the stores are exchanged; a shard whose predecessors' stores changed calls again with
 * WSPR_HASH_REVISIT (same buffers and result arrays: only the affected segments are decoded again); when no store list
 * changes any more, ONE process calls wspr_hash_commit() with all stores in segment order.
 * WSPR_HASH_REVISIT is refused (negative return, results untouched) unless the calling thread's previous hashed call
 * completed over the same segments, samples and slot layout and the library's working buffers have not been released
 * since.  Calls with WSPR_HASH_KEEP_FILE or a prior are ordered by a process-wide turn which they keep while they decode:
 * shards driven from several threads of one process run one after the other; give every shard its own process (dist.py). */
typedef struct wspr_hash_op {
    int32_t seg;                /* global segment index */
    int32_t slot;               /* 0 .. HASHTAB_SIZE - 1 */
    int32_t kind;               /* 1 = type-1 store (call + locator), 2 = type-2 store (call only) */
    char    call[13];
    char    grid[5];
    char    pad[2];
} wspr_hash_op;
#define WSPR_HASH_KEEP_FILE 1   /* do not write hashtable.txt (a shard of a larger job) */
#define WSPR_HASH_REVISIT   2   /* the calling thread's previous hashed call once more, under a new `prior` */
int wspr_decode_batch_hashed(float *idat, float *qdat, int nseg, int samples, size_t seg_stride,
                             struct decoder_options options, struct decoder_results *decodes, int max_results,
                             int *n_results, int writeback, int seg_index0, const wspr_hash_op *prior, int n_prior,
                             int flags, wspr_hash_op *stores_out, int cap, int *n_stores, int *n_redecoded);
/* hashtable.txt := hashtable.txt + stores, applied in the order given (the job's stores in segment order). */
int wspr_hash_commit(const wspr_hash_op *stores, int n);

/* Host buffers at speed.  wspr_decode() / wspr_decode_batch() take the caller's HOST memory as the reference does
 * (wsprd.h:106-111; call sites rtlsdr_wsprd.c:316, :689).  Pageable memory is gathered through pinned chunks of the
 * library (host memcpy + DMA, chunk k+1 gathered under the DMA of chunk k); PINNED memory goes to the device as a plain
 * DMA with no host work.  The reference's callers allocate their I/Q buffers once (rtlsdr_wsprd.c:78-90, 331-336):
 * pin them once with this (hipHostRegister; no HIP headers needed on the caller's side) and unpin before free().
 * Returns 0 on success (also when the range is pinned already), -1 otherwise. */
int wspr_pin_host_buffer(void *p, size_t bytes);
int wspr_unpin_host_buffer(void *p);

/* Same, with the IQ already resident in HBM (device pointers, same layout).
 * The input is copied to a working buffer and left untouched.  This is the
 * entry point bench.py times.
 * Stream contract (every entry point that takes device pointers): the library works on streams of its own
 * and does not know the caller's; the buffers must be COMPLETE when the call is made (synchronise the stream
 * or event that produces them first), and everything the call writes is complete when it returns. */
int wspr_decode_batch_device(const void *d_idat, const void *d_qdat, int nseg, int samples,
                             size_t seg_stride, struct decoder_options options,
                             struct decoder_results *decodes, int max_results, int *n_results);

/* Node-level form (SURVEY §8e "one host process, 8 devices"): the nseg host segments are split into contiguous
 * blocks by wspr_shard_range() over `ndevices` HIP devices (0 = every visible device; more than are visible is an
 * error), one host thread per device, each block decoded by wspr_decode_batch() on its device straight into the
 * caller's arrays.  The segments are independent (hashtab/loctab are locals of wspr_decode, wsprd.c:478-479), so no
 * collective is involved and the result equals wspr_decode_batch() on one device.  The host's CPUs are shared
 * between the devices (pools of contexts created from then on are sized for a 1/ndevices share).  With
 * options.usehashtable the segments are ordered and the call is wspr_decode_batch() on the current device. */
int wspr_decode_batch_node(float *idat, float *qdat, int nseg, int samples, size_t seg_stride,
                           struct decoder_options options, struct decoder_results *decodes,
                           int max_results, int *n_results, int ndevices);
/* The same for IQ already RESIDENT on device `src_device` (rows of seg_stride floats, e.g. what
 * wspr_decimate_u8_batch_device() left there): every other device pulls its wspr_shard_range() block over xGMI
 * with a peer copy (the scatter of 360 000 B per segment for real inputs, here inside one process),
 * decodes it on its own host thread and writes its spots into the caller's arrays.  The stream contract of
 * wspr_decode_batch_device() applies to d_idat / d_qdat.
 * Failure semantics of both node-level calls: ndevices larger than the visible devices, or a source device that does not
 * exist: -1 before anything runs.  Refused peer access is NOT a failure: the block then travels source device -> pinned
 * host memory -> target device (a line on stderr says so).  A shard that fails (its device cannot be selected, its copy
 * or its decode fails) fails the WHOLE call: the return value is negative and every n_results[] is 0 -- the other shards'
 * spots are not reported, so that a caller never mistakes a partial result for the slot's spots; the call can be
 * repeated (with fewer devices) as it stands, the inputs are untouched. */
int wspr_decode_batch_node_device(const void *d_idat, const void *d_qdat, int src_device, int nseg, int samples,
                                  size_t seg_stride, struct decoder_options options,
                                  struct decoder_results *decodes, int max_results, int *n_results, int ndevices);
/* The partitioning rule of the node-level call and of the multi-process driver (rtlsdr-wsprd_amd/dist.py
 * shard_range): shard k of n owns segments [*lo, *hi), the first nseg % n shards one segment more. */
void wspr_shard_range(int nseg, int shard, int nshards, int *lo, int *hi);

/* Front end, replaces the static rtlsdr_callback() + decoder() normalisation
 * (reference rtlsdr_wsprd.c:126-244, :284-305).  iq = interleaved unsigned 8-bit
 * I/Q at 2.4 Msps, nbytes a multiple of 8, decimator state zero at the start.
 * Writes min(floor(nbytes/2/6401), 45000) samples to I/Q (capacity 45000 each;
 * the rest is zero-filled when normalise != 0, which also scales to peak 0.5). */
int wspr_decimate_u8(const uint8_t *iq, size_t nbytes, float *I, float *Q, uint32_t *n_out,
                     int normalise);
/* nseg raw segments resident in HBM -> planar float IQ in HBM (rows of
 * wspr_iq_stride() floats), normalised, ready for wspr_decode_batch_device.  d_raw and bytes_per_seg must be
 * multiples of 16 (rows are read with aligned 16-byte loads); -1 otherwise. */
int wspr_decimate_u8_batch_device(const void *d_raw, size_t bytes_per_seg, int nseg,
                                  void *d_idat, void *d_qdat, int normalise);

/* Streaming form of the front end.  The reference keeps the decimator state in `static` variables
 * of rtlsdr_callback() (rtlsdr_wsprd.c:135-160: integrators, comb delay lines, FIR line, decimation
 * counter), so it runs on across callbacks and across 2-minute segments; this state object carries
 * exactly that between calls.  All-zero = the receiver at start-up.  Fields: samples into the open
 * decimation block, integrators I1/I2 per rail (I, Q), and the second integrator at the last 36
 * decimation instants (from which both combs and the FIR history follow). */
typedef struct wspr_decim_state {
    uint32_t phase;
    uint32_t x1[2], x2[2];
    uint32_t hist[2][36];
} wspr_decim_state;
void wspr_decim_stream_reset(wspr_decim_state *st);
/* One callback's worth of samples (rtlsdr_wsprd.c:126: `buf`, `len`; nbytes a multiple of 16, the
 * mixer phase restarts with every call as in the reference): appends the new outputs to I/Q at
 * index `fill`, never beyond `capacity` (rtlsdr_wsprd.c:236-242), and updates the state.  Any
 * chunking of a stream gives the same outputs as one call over the whole of it. */
int wspr_decimate_u8_stream(wspr_decim_state *st, const uint8_t *iq, size_t nbytes, float *I, float *Q,
                            uint32_t fill, uint32_t capacity, uint32_t *new_fill);
/* Many receivers at once, resident data: row s of d_raw is the next bytes_per_seg bytes of receiver
 * s, d_states[s] its state (device memory, read and updated), outputs start at column 0 of row s of
 * d_idat/d_qdat (row stride wspr_iq_stride()), n_out[s] (host) = outputs produced. */
int wspr_decimate_u8_batch_device_stateful(const void *d_raw, size_t bytes_per_seg, int nseg, void *d_states,
                                           void *d_idat, void *d_qdat, int *n_out);
size_t wspr_iq_stride(void);        /* floats per segment row of device IQ buffers */
/* The front end's constants as the kernels use them: the 33 compensation-FIR taps (zCoef, reference
 * rtlsdr_wsprd.c:142-152) and the input samples per output (DOWNSAMPLING + 1 = 6401, :41, :198-202). */
void wspr_front_end_constants(float *taps33, int *samples_per_output);

/* ---- receiver session --------------------------------------------------------------------
 * The reference's rx_state (two I/Q buffers of 45000 samples, their fill counters, the active index,
 * rtlsdr_wsprd.c:78-90), the decimator's static state (:135-160) and the body of its decoder thread (:263-328) as
 * one object.  The application keeps its three threads: the RX thread feeds librtlsdr callback buffers, the main
 * loop rolls the buffers over on the even minute, the decoder thread decodes the completed buffer.  feed() may
 * run beside decode(): feed() and rollover() exclude each other (a roll-over waits for the callback in flight, and
 * nothing is written into a buffer once rollover() has returned its index), and the front end runs on a lane of
 * the library that wspr_bind_thread_lane() never hands out. */
typedef struct wspr_session wspr_session;
wspr_session *wspr_session_create(struct decoder_options options);     /* initSampleStorage(), :331-336 */
void wspr_session_destroy(wspr_session *s);
/* rtlsdr_callback(buf, len), :126-244: mixer + CIC + FIR into the active buffer (outputs beyond 45000 are
 * dropped, :236-242).  len a multiple of 16 (librtlsdr delivers 65536).  Returns the buffer's fill, < 0 on error. */
int wspr_session_feed(wspr_session *s, const uint8_t *buf, uint32_t len);
/* One callback of EACH of n receivers at once (bufs[k]: len bytes for sessions[k]; the same len, a multiple of 16, for
 * all; the sessions distinct): what n wspr_session_feed() calls do, as one transfer and one launch set -- a callback's
 * cost is its round trips, not its arithmetic.  fills[k] (optional): sessions[k]'s fill afterwards.  0, < 0 on error. */
int wspr_session_feed_many(wspr_session *const *sessions, const uint8_t *const *bufs, uint32_t len, int n, int *fills);
/* Main loop on the 2-minute boundary, :1179-1182: switches to the other buffer (fill reset to 0) and returns the
 * index of the buffer that just completed. */
int wspr_session_rollover(wspr_session *s);
/* decoder(), :263-328, on a completed buffer: returns 0 without decoding if it holds fewer than 117 s of samples
 * (:277), else zeroes the tail (:284-288), normalises to a peak of 0.5 (:290-305), calls wspr_decode() (:312-317)
 * and returns 1 (negative: no usable device).  The buffer then holds what the reference's saveSample() would see. */
int wspr_session_decode(wspr_session *s, int buffer, struct decoder_results *decodes, int *n_results);
/* Many receivers, one slot: the completed buffers (buffers[k] of sessions[k]) of n sessions decoded TOGETHER -- one
 * batch call per distinct set of decoder options (the receivers of one band share theirs).  decodes: n rows of
 * max_results spots; n_results[k]; decoded[k] (optional) = 1 / 0 as wspr_session_decode() would have returned.  Spots
 * and what the buffers hold afterwards are those of wspr_session_decode() on each session in index order (with
 * usehashtable on any of them that is the order of the hash memory: then only RUNS of consecutive sessions with equal
 * options share a batch call, so sessions with options A, B, A are decoded 0, 1, 2).  Returns the number of buffers
 * decoded, negative on error. */
int wspr_session_decode_many(wspr_session *const *sessions, const int *buffers, int n, struct decoder_results *decodes,
                             int max_results, int *n_results, int *decoded);
uint32_t wspr_session_fill(const wspr_session *s, int buffer);
const float *wspr_session_samples(const wspr_session *s, int buffer, int rail /* 0 = I, 1 = Q */);
/* Microseconds until the next even UTC minute, :1170-1175 (what the main loop sleeps). */
uint32_t wspr_usec_to_next_slot(long tv_sec, long tv_usec);
/* UTC time stamp of the frame being decoded: gmtime(now - 120 + 1), :307-310; for wspr_format_spot_timestamped(). */
void wspr_frame_time(long unixtime_now, int *year, int *month, int *day, int *hour, int *minute);

/* ---- recorded files and the playback print format ----------- */
/* Replaces readRawIQfile(), reference rtlsdr_wsprd.c:555-592 (.iq: interleaved f32, Q negated,
 * normalised to peak 0.5).  I/Q: 45000 floats each.  Returns complex samples read, 0 on error. */
int wspr_read_iq_file(const char *filename, float *I, float *Q);
/* Replaces readC2file(), reference rtlsdr_wsprd.c:620-667 (14-byte name, int, double dial Hz). */
int wspr_read_c2_file(const char *filename, float *I, float *Q, double *dial_hz);
/* Replaces writeRawIQfile(), reference rtlsdr_wsprd.c:595-617. */
int wspr_write_iq_file(const char *filename, const float *I, const float *Q);
/* The "Spot : ..." line of decodeRecordedFile(), reference rtlsdr_wsprd.c:691-701. */
int wspr_format_spot(const struct decoder_results *r, char *out, size_t cap);

/* The daemon's stdout line, reference printSpots() rtlsdr_wsprd.c:447-474 (UTC frame time). */
int wspr_format_spot_timestamped(const struct decoder_results *r, int year, int month, int day, int hour,
                                 int minute, char *out, size_t cap);
/* Text of the wsprnet.org report URL, reference postSpots() rtlsdr_wsprd.c:390-397 (r == NULL: the
 * "no spot" status report) and :414-429 (one spot).  Formatting only -- nothing is sent. */
int wspr_format_wsprnet_url(const struct decoder_results *r, const struct decoder_options *opt,
                            double dial_hz, int year, int month, int day, int hour, int minute,
                            const char *app_version, char *out, size_t cap);

/* ---- the reference's other decoder entry points, GPU-backed ---------------- */
/* Replaces sync_and_demodulate(), reference wsprd/wsprd.h:76-91 (GPU-backed).  symfac is honoured (mode 2,
 * wsprd.c:250); the decoder itself always passes 50. */
void sync_and_demodulate(float *id, float *qd, long np, unsigned char *symbols, float *freq,
                         int ifmin, int ifmax, float fstep, int *shift, int lagmin, int lagmax,
                         int lagstep, float *drift, int symfac, float *sync, int mode);
/* Replaces subtract_signal(), reference wsprd/wsprd.h:83-89 / wsprd.c:263-312 (symbol-by-symbol subtraction;
 * declared by the reference's header, never called by its decoder; GPU-backed). */
void subtract_signal(float *id, float *qd, long np, float f0, int shift, float drift,
                     const unsigned char *channel_symbols);
/* Replaces subtract_signal2(), reference wsprd/wsprd.h:99-105 (GPU-backed). */
void subtract_signal2(float *id, float *qd, long np, float f0, int shift, float drift,
                      const unsigned char *channel_symbols);
/* Timing of the stages of the most recent batch call, milliseconds (HIP events on the library's
 * streams / host clock, summed over the slots).  Order: [0] FFT+sync stage, [1] host bookkeeping,
 * [2] device Fano tail (K6), [3] fine sync + demod, [4] subtract, [5] host Fano, [6] total wall time,
 * then counts: [7] Fano calls, [8] Fano time-outs, [9] Fano cycles, [10] candidates refined, [11] GPU
 * waves, [12] Fano attempts left to the device tail, [13] segments decoded a second time, [14] refined
 * candidates whose result was consumed (the others: speculation cut by a subtraction), [15] subtractions;
 * then the CPU time (not wall time) the calling thread spent in the call, milliseconds: [16] all of it, by phase [17] pass
 * start (candidate lists, re-ranking), [18] wave building, [19] fine search + first rung (launches, lists, gates),
 * [20] ladder, [21] bookkeeping (unpack, re-encode, de-dup), [22] subtraction launches, [23] result hand-over;
 * then [24] decoded messages looked up in the calling thread's message cache (what a 50-bit message unpacks and
 * re-encodes to, computed once per thread and message) and [25] how many of them it answered.
 * Returns the number of values written (<= capacity). */
int wspr_last_timings(double *ms, int capacity);
/* Worker threads of the library's host pools alive in this process (the threads that call into the library are
 * not counted): what a rank adds to the host's load besides its callers -- 0 when its CPU share (WSPR_HOST_THREADS,
 * or the share of a node-level call) is no larger than the slots it drives. */
int wspr_host_pool_workers(void);
/* Device Fano search (K6w) over n soft-symbol vectors of 162 bytes in transmission
 * (interleaved) order, i.e. deinterleave() + fano() of reference wsprd.c:759-761 (fano.c:87-238) with
 * delta 60: the exact wave-parallel search the decoder uses, one wavefront per vector (fano_wave.h).  Outputs per
 * vector: ret (0 / -1), cycles and data[10] are the reference's; metric/maxnp are filled for decoded frames only
 * (the reference's caller reads nothing else of a failed attempt, wsprd.c:759-766).  steps (may be NULL):
 * expansion steps per vector. */
int wspr_fano_batch_device_wave(const unsigned char *symbols, int n, unsigned maxcycles, int *ret,
                                unsigned *cycles, unsigned *metric, unsigned *maxnp, unsigned char *data,
                                unsigned *steps);
/* Devices.  A host thread decodes on the HIP device that is current for it; wspr_set_device() makes
 * `device` current for the calling thread (0 on success, -1 if there is no such device).  The library keeps
 * separate contexts (streams, buffers, host pools) per device, so one process can drive all GPUs of a node
 * with one host thread (or one per lane) per device, segments split by the caller -- no collective is
 * involved. */
int wspr_device_count(void);
int wspr_set_device(int device);
/* Concurrency.  Like the reference, the library is not re-entrant within one lane -- calls made on the same lane of
 * the same device from several threads TAKE TURNS (safe, but nothing overlaps); it keeps up to sixteen independent
 * lanes (streams, buffers, host pools).  A host thread is bound to lane 0 until it calls this (returns the lane
 * actually bound, 0..15; one more lane is reserved for receiver sessions); calls made from threads bound to different
 * lanes may overlap, e.g. to start the next batch under the tail of the current one. */
int wspr_bind_thread_lane(int lane);
/* A batch of >= 128 segments is split over up to WSPR_SLOTS (default 3) concurrent pipelines ("slots") of the
 * calling thread's lane, so that one call overlaps its own host phases with its own kernels.  A service that
 * already keeps several batches in flight on different lanes gets that overlap from the lanes: this caps the
 * slots of the calling thread's later calls at n (n <= 0: no cap) and returns the number they will use. */
int wspr_set_thread_slots(int n);
/* Memory.  The library keeps the work buffers of every (device, lane, slot) it has used, sized for the largest
 * batch seen there (about 1 MB of HBM per segment).  This returns the work buffers of the CURRENT device -- all
 * lanes, device and pinned host memory -- to the driver and reports the device bytes freed; tables, streams and
 * host pools stay, and the next call allocates what it needs.  Calls in flight on that device (any lane, any
 * receiver session's feed) finish first and calls arriving meanwhile wait (since round 5; before, none was allowed
 * to be in flight).  Nothing a caller can observe lives in these buffers between calls. */
size_t wspr_release_buffers(void);
/* Scheduler tuning for crowded bands (batches of >= 256 segments per slot): the host Fano pool gives
 * every attempt `cycles_per_bit` cycles per bit; attempts still running then are finished by K6 with
 * the reference's 10000, and a segment in which one of those decodes after all is decoded again with
 * the full budget everywhere, so results never depend on this value.  Default 10000 = split off
 * (env WSPR_FANO_FAST overrides); a few hundred pays once a batch carries thousands of Fano time-outs (the
 * device tail then takes tens of milliseconds for all of them).  Returns the previous value. */
unsigned wspr_set_fano_fast_budget(unsigned cycles_per_bit);
/* Where a batch's Fano attempts run: 0 = the host pool (the north star's partition; with the budget split
 * above if set), 1 = every attempt on the device (the exact wave-parallel search, straight from the soft
 * symbols in HBM: no host Fano, nothing postponed or decoded twice), -1 = automatic (default; env
 * WSPR_FANO_DEVICE): the device for batches of >= 32 segments per pipeline when the rank has fewer than four
 * host threads or the pipeline's previous batch met more than one time-out per ten segments (a crowded band),
 * the host otherwise (single calls always).  Results are identical in every mode.  Returns the previous value. */
int wspr_set_fano_device_mode(int mode);
/* Library / device description, e.g. for bench logs. */
const char *wspr_mi355x_version(void);
int wspr_device_ready(void);        /* 1 if a HIP device and the kernels are usable */

/* ---- message layer, same names as the reference exports ------------------- */
/* reference wsprd/wsprsim_utils.h:1-9 */
char get_locator_character_code(char ch);
char get_callsign_character_code(char ch);
long unsigned int pack_grid4_power(char const *grid4, int power);
long unsigned int pack_call(char const *callsign);
void pack_prefix(char *callsign, int32_t *n, int32_t *m, int32_t *nadd);
void interleave(unsigned char *sym);
int  get_wspr_channel_symbols(char *message, char *hashtab, char *loctab, unsigned char *symbols);
/* reference wsprd/wsprd_utils.h:32-42 */
void unpack50(signed char *dat, int32_t *n1, int32_t *n2);
int  unpackcall(int32_t ncall, char *call);
int  unpackgrid(int32_t ngrid, char *grid);
int  unpackpfx(int32_t nprefix, char *call);
void deinterleave(unsigned char *sym);
int  doublecomp(const void *elem1, const void *elem2);
int  floatcomp(const void *elem1, const void *elem2);
int  unpk_(signed char *message, char *hashtab, char *loctab, char *call_loc_pow, char *call,
           char *loc, char *pwr, char *callsign);
/* reference wsprd/fano.h:14-28 */
int  fano(unsigned int *metric, unsigned int *cycles, unsigned int *maxnp, unsigned char *data,
          unsigned char *symbols, unsigned int nbits, int mettab[2][256], int delta,
          unsigned int maxcycles);
int  encode(unsigned char *symbols, unsigned char *data, unsigned int nbytes);
extern unsigned char Partab[];
/* reference wsprd/nhash.h:3 */
uint32_t nhash(const void *key, size_t length, uint32_t initval);
/* The soft-decision metric tables, reference wsprd/metric_tables.h:8 (a data symbol of the reference's wsprd.o;
 * rows = Es/No 0, 3, 6, 9 dB and a fifth row whose last eight entries are uninitialised-memory values in the
 * reference source -- kept as they are).  wspr_decode uses row 2 only (wsprd.c:471-472). */
extern float metric_tables[5][256];
/* the integer branch-metric table wspr_decode derives at wsprd/wsprd.c:467-473 */
void wspr_fano_metric_table(int mettab[2][256]);

#ifdef __cplusplus
}
#endif
#endif /* WSPR_MI355X_H */

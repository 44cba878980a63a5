/* ============================================================================
 * wspr_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Plain-C restatement of the reference WSPR decode hot path
 * (Guenael/rtlsdr-wsprd v0.5.6: wsprd/wsprd.c, fano.c, wsprd_utils.c,
 * wsprsim_utils.c, nhash.c and the decimator/normaliser of rtlsdr_wsprd.c).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this library, and only as the checker.  The product
 * (libwspr_mi355x.so) never includes, links or calls anything in oracle/.
 *
 * Every function cites the reference file:line it restates.
 * ==========================================================================*/
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NSYM        162
#define ORC_NBITS       81
#define ORC_SPS         256          /* samples per symbol at 375 sps      */
#define ORC_FFT         512
#define ORC_MAXSAMPLES  45000        /* 120 s * 375 sps                    */
#define ORC_MAXCAND     200
#define ORC_MAXUNIQ     100
#define ORC_HASH_N      32768
#define ORC_HASH_W      13
#define ORC_LOC_W       5

/* Same memory layout as struct decoder_options / decoder_results / cand
 * (reference wsprd/wsprd.h:44-74). */
typedef struct {
    int  freq;
    char rcall[13];
    char rloc[7];
    int  quickmode;
    int  usehashtable;
    int  npasses;
    int  subtraction;
} orc_options;

typedef struct {
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
} orc_spot;

typedef struct {
    float freq;
    float snr;
    int   shift;
    float drift;
    float sync;
} orc_cand;

/* Optional per-stage trace of one orc_wspr_decode() call (tests only). */
#define ORC_TRACE_PASSES 3
typedef struct {
    int      passes_run;
    int      blocks;
    float    noise_level[ORC_TRACE_PASSES];
    int      npk[ORC_TRACE_PASSES];
    float    smspec_raw[ORC_TRACE_PASSES][411];       /* before normalisation */
    orc_cand cand_peaks[ORC_TRACE_PASSES][ORC_MAXCAND];   /* after snr sort   */
    orc_cand cand_coarse[ORC_TRACE_PASSES][ORC_MAXCAND];  /* after coarse sync*/
    orc_cand cand_fine[ORC_TRACE_PASSES][ORC_MAXCAND];    /* after mode 0+1   */
    int      mode0_shift[ORC_TRACE_PASSES][ORC_MAXCAND];
    float    mode0_sync[ORC_TRACE_PASSES][ORC_MAXCAND];
    int      n_visited[ORC_TRACE_PASSES];             /* candidates entered   */
    int      attempts[ORC_TRACE_PASSES][ORC_MAXCAND]; /* mode-2 calls         */
    int      fano_calls[ORC_TRACE_PASSES][ORC_MAXCAND];
    int      decoded[ORC_TRACE_PASSES][ORC_MAXCAND];
    int      subtracted[ORC_TRACE_PASSES][ORC_MAXCAND];
    float    first_rms[ORC_TRACE_PASSES][ORC_MAXCAND];
    float    first_sync2[ORC_TRACE_PASSES][ORC_MAXCAND];
    unsigned char first_symbols[ORC_TRACE_PASSES][ORC_MAXCAND][ORC_NSYM];
    unsigned fano_metric[ORC_TRACE_PASSES][ORC_MAXCAND];
    unsigned fano_cycles[ORC_TRACE_PASSES][ORC_MAXCAND];
    unsigned fano_maxnp[ORC_TRACE_PASSES][ORC_MAXCAND];
    unsigned char decdata[ORC_TRACE_PASSES][ORC_MAXCAND][11];
    long     fano_cycles_total;                        /* sum over all calls  */
} orc_trace;

/* ---- message layer ------------------------------------------------------ */
uint32_t orc_nhash(const void *key, size_t length, uint32_t initval);
char     orc_call_char_code(char ch);
char     orc_loc_char_code(char ch);
unsigned long orc_pack_call(const char *callsign);
unsigned long orc_pack_grid4_power(const char *grid4codes, int power);
void     orc_pack_prefix(char *callsign, int32_t *n, int32_t *m, int32_t *nadd);
void     orc_interleave(unsigned char *sym);
void     orc_deinterleave(unsigned char *sym);
int      orc_conv_encode(unsigned char *symbols, const unsigned char *data, unsigned nbytes);
int      orc_fano(unsigned *metric, unsigned *cycles, unsigned *maxnp,
                  unsigned char *data, const unsigned char *symbols, unsigned nbits,
                  const int mettab[2][256], int delta, unsigned maxcycles);
void     orc_build_mettab(int mettab[2][256]);
void     orc_unpack50(const signed char *dat, int32_t *n1, int32_t *n2);
int      orc_unpackcall(int32_t ncall, char *call);
int      orc_unpackgrid(int32_t ngrid, char *grid);
int      orc_unpackpfx(int32_t nprefix, char *call);
int      orc_unpk(const signed char *message, char *hashtab, char *loctab,
                  char *call_loc_pow, char *call, char *loc, char *pwr, char *callsign);
int      orc_channel_symbols(const char *rawmessage, char *hashtab, char *loctab,
                             unsigned char *symbols);
extern unsigned char orc_sync_vector[ORC_NSYM];

/* ---- DSP stages --------------------------------------------------------- */
/* 512-point forward FFT, float32, radix-2 DIF, output in natural order. */
void orc_fft512(float *re, float *im);
/* Robustness study (orc_fft_alt.c): 0 = the FFT above (default, the product's arithmetic), 1 = float64 rounded to
 * float32, 2..7 = other float32 factorisations.  Process-wide; set it before decoding, not while other threads decode. */
int  orc_set_fft_variant(int variant);
int  orc_get_fft_variant(void);
void orc_fft512_variant(float *re, float *im);
/* ps[bin*blocks + t], bin 0..511 (fft-shifted), t 0..blocks-1. */
int  orc_blocks_for(int samples);
void orc_fft_bank(const float *idat, const float *qdat, int samples, float *ps);
/* returns npk; cand sorted by snr desc; smspec_raw (411) optional. */
int  orc_pick_peaks(const float *ps, int blocks, orc_cand *cand,
                    float *noise_level, float *smspec_raw, float *smspec_norm);
void orc_coarse_sync(const float *ps, int blocks, orc_cand *cand, int npk, int maxdrift);
void orc_sync_demod(const float *id, const float *qd, long np, unsigned char *symbols,
                    float *freq, int ifmin, int ifmax, float fstep,
                    int *shift, int lagmin, int lagmax, int lagstep,
                    const float *drift, int symfac, float *sync, int mode);
void orc_subtract(float *id, float *qd, long np, float f0, int shift, float drift,
                  const unsigned char *channel_symbols);
void orc_subtract_simple(float *id, float *qd, long np, float f0, int shift, float drift,
                         const unsigned char *channel_symbols);
int  orc_wspr_decode(float *idat, float *qdat, int samples, orc_options opt,
                     orc_spot *spots, int *n_results, orc_trace *trace);

/* ---- front end ---------------------------------------------------------- */
typedef struct orc_decim_state orc_decim_state;
orc_decim_state *orc_decim_new(void);
void   orc_decim_free(orc_decim_state *);
/* Feed interleaved u8 IQ (nbytes multiple of 8; the buffer is NOT modified),
 * append outputs to I/Q (capacity cap), returns new fill count. */
uint32_t orc_decim_feed(orc_decim_state *, const unsigned char *iq, size_t nbytes,
                        float *I, float *Q, uint32_t fill, uint32_t cap);
void   orc_normalise(float *I, float *Q, int n_valid, int n_total);
void   orc_front_end_constants(float *taps33, int *samples_per_output);
/* .iq file semantics: interleaved f32 -> planar with Q negated, then normalise */
int    orc_iq_from_interleaved(const float *file_f32, int nfloats, float *I, float *Q);

#ifdef __cplusplus
}
#endif

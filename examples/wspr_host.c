/* wspr_host -- the reference application's decoder-side modes as a C program over the C ABI of libwspr_mi355x.so.
 *
 * The reference's host side is C (rtlsdr_wsprd.c); this is the same language over include/wspr_mi355x.h and nothing
 * else: no HIP header, no C++, no Python.  It covers what rtlsdr_wsprd does once samples exist -- the parts that need a
 * dongle (librtlsdr) and the network (curl) stay in the reference application, which links the library unchanged
 * (INTEGRATION.md section 2):
 *
 *   -r FILE [FILE ...]   playback, decodeRecordedFile() of rtlsdr_wsprd.c:668-703: .iq / .c2 by extension, the same
 *                        lines on stdout.  Several files are ONE wspr_decode_batch() call (the MI355X form of the loop).
 *   -t                   self-test, decoderSelfTest() of rtlsdr_wsprd.c:727-790: "K1JT FN20QI 20" at 50 Hz, 2.0 s,
 *                        amplitude 1 in noise of sigma 0.02, saved as selftest.iq, decoded; exit status 0 iff the first
 *                        spot is K1JT / FN20 / 20 (the reference's check).  The noise generator is this file's own
 *                        (the reference draws from rand(); its exact sample values are pinned in tests/, not here).
 *   -i FILE|-            raw unsigned 8-bit I/Q at 2.4 Msps (what `rtl_sdr -s 2400000 -f <dial+1500+600000> -` writes)
 *                        through the receiver session: 65 536-byte callbacks (rtlsdr_wsprd.c:126), a roll-over every
 *                        576 000 000 bytes (the two minutes the main loop counts off the wall clock, :1170-1185), the
 *                        decoder thread's body on the completed buffer (:263-328), printSpots()' lines (:447-474) with
 *                        frame times counted from -T (UTC seconds of the first sample's slot; default 0).  Repeated,
 *                        every -i is one more RECEIVER: one callback of each goes to the GPU together
 *                        (wspr_session_feed_many()), and at the even minute all completed buffers are decoded together
 *                        (wspr_session_decode_many()); lines carry "[k] " in front.
 *   decoder options      -f dial Hz, -c call, -l locator, -H, -Q, -S as rtlsdr_wsprd.c:862-970.
 *
 * Build: make -C examples      (gcc, links ../rtlsdr-wsprd_amd/libwspr_mi355x.so with an rpath)
 */
#include <errno.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "wspr_mi355x.h"

#define SLOT_SAMPLES 45000                 /* 120 s at 375 sps: SIGNAL_LENGHT * SIGNAL_SAMPLE_RATE, rtlsdr_wsprd.h */
#define CALLBACK_BYTES 65536u              /* librtlsdr's buffer, rtlsdr_wsprd.c:1136 */
#define SLOT_BYTES 576000000ull            /* 120 s * 2.4 Msps * 2 bytes */
#define MAX_SPOTS 50                       /* dec_results[50], rtlsdr_wsprd.c:117 */

static const char kHeader[] = "        SNR      DT        Freq Dr    Call    Loc Pwr";

static void usage(const char *argv0) {
    fprintf(stderr,
            "use: %s [-f dial_hz] [-c call] [-l locator] [-H] [-Q] [-S] (-r FILE [FILE ...] | -t | -i RAWFILE|- [-i RAWFILE ...] [-T utc_seconds])\n",
            argv0);
}

static int has_suffix(const char *name, const char *suffix) {
    const size_t n = strlen(name), m = strlen(suffix);
    return n >= m && strcmp(name + n - m, suffix) == 0;
}

/* ---- -r: recorded files ---------------------------------------------------------------------------------------- */
static int playback(int nfiles, char **files, struct decoder_options opt) {
    float *I = calloc((size_t)nfiles * SLOT_SAMPLES, sizeof *I);
    float *Q = calloc((size_t)nfiles * SLOT_SAMPLES, sizeof *Q);
    struct decoder_results *spots = calloc((size_t)nfiles * MAX_SPOTS, sizeof *spots);
    int *nspots = calloc((size_t)nfiles, sizeof *nspots);
    int *nread = calloc((size_t)nfiles, sizeof *nread);
    if (!I || !Q || !spots || !nspots || !nread) { fprintf(stderr, "out of memory\n"); return 2; }
    int rc = 0;
    for (int k = 0; k < nfiles; ++k) {
        float *i = I + (size_t)k * SLOT_SAMPLES, *q = Q + (size_t)k * SLOT_SAMPLES;
        if (has_suffix(files[k], ".iq")) {
            nread[k] = wspr_read_iq_file(files[k], i, q);
        } else if (has_suffix(files[k], ".c2")) {
            double dial = 0.0;
            nread[k] = wspr_read_c2_file(files[k], i, q, &dial);
        } else {
            fprintf(stderr, "Not a valid extension!! (only .iq & .c2 files)\n");
            rc = 2;
        }
    }
    if (rc == 0) {
        /* a single file is the reference's call as it stands (its `samples` argument is what was read); several files
         * are one batch of whole records -- the readers zero-fill short files -- decoded together */
        const int r = nfiles == 1 ? wspr_decode(I, Q, nread[0] > 0 ? nread[0] : 0, opt, spots, nspots)
                                  : wspr_decode_batch(I, Q, nfiles, SLOT_SAMPLES, SLOT_SAMPLES, opt, spots, MAX_SPOTS, nspots, 0);
        if (nfiles == 1 && nread[0] <= 0) nspots[0] = 0;
        if (r < 0) { fprintf(stderr, "decode failed (%d): no usable MI355X\n", r); rc = 3; }
    }
    for (int k = 0; k < nfiles && rc == 0; ++k) {
        if (nfiles > 1) printf("%s\n", files[k]);
        printf("Number of samples: %d\n", nread[k]);
        if (nread[k] <= 0) continue;
        printf("%s\n", kHeader);
        for (int s = 0; s < nspots[k]; ++s) {
            char line[128];
            wspr_format_spot(&spots[(size_t)k * MAX_SPOTS + s], line, sizeof line);
            printf("%s\n", line);
        }
    }
    free(I); free(Q); free(spots); free(nspots); free(nread);
    return rc;
}

/* ---- -t: self-test ---------------------------------------------------------------------------------------------- */
static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static double uniform01(void) {                       /* xorshift64*: 53 random bits in (0, 1) */
    g_rng ^= g_rng >> 12; g_rng ^= g_rng << 25; g_rng ^= g_rng >> 27;
    return ((double)((g_rng * 0x2545F4914F6CDD1Dull) >> 11) + 0.5) / 9007199254740992.0;
}
static void gaussian_pair(double sigma, float *a, float *b) {          /* Box-Muller, both outputs used */
    const double r = sigma * sqrt(-2.0 * log(uniform01())), t = 2.0 * M_PI * uniform01();
    *a = (float)(r * cos(t));
    *b = (float)(r * sin(t));
}

static int self_test(struct decoder_options opt) {
    static float I[SLOT_SAMPLES], Q[SLOT_SAMPLES];
    static char hashtab[HASHTAB_SIZE * HASHTAB_ENTRY_LEN], loctab[HASHTAB_SIZE * LOCTAB_ENTRY_LEN];
    unsigned char symbols[162];
    char message[] = "K1JT FN20QI 20";
    if (!get_wspr_channel_symbols(message, hashtab, loctab, symbols)) { fprintf(stderr, "message does not encode\n"); return 2; }
    const double tone_spacing = 375.0 / 256.0, sample_time = 1.0 / 375.0, f0 = 50.0, t0 = 2.0;
    const int first = (int)(t0 / sample_time);
    double phase = 0.0;
    for (int sym = 0; sym < 162; ++sym) {
        const double step = 2.0 * M_PI * sample_time * (f0 + ((double)symbols[sym] - 1.5) * tone_spacing);
        for (int j = 0; j < 256; ++j, phase += step) {
            float ni, nq;
            gaussian_pair(0.02, &ni, &nq);
            I[first + 256 * sym + j] = (float)cos(phase) + ni;
            Q[first + 256 * sym + j] = (float)sin(phase) + nq;
        }
    }
    wspr_write_iq_file("selftest.iq", I, Q);
    struct decoder_results spots[MAX_SPOTS];
    int n = 0;
    memset(spots, 0, sizeof spots);
    const int r = wspr_decode(I, Q, SLOT_SAMPLES, opt, spots, &n);
    if (r < 0) { fprintf(stderr, "decode failed (%d): no usable MI355X\n", r); return 3; }
    printf("%s\n", kHeader);
    for (int s = 0; s < n; ++s)
        printf("Spot(%i) %6.2f %6.2f %10.6f %2d %7s %6s %2s\n", s, spots[s].snr, spots[s].dt, spots[s].freq,
               (int)spots[s].drift, spots[s].call, spots[s].loc, spots[s].pwr);
    const int ok = n > 0 && !strcmp(spots[0].call, "K1JT") && !strcmp(spots[0].loc, "FN20") && !strcmp(spots[0].pwr, "20");
    printf("%s\n", ok ? "Self-test SUCCESS!" : "Self-test FAILED!");
    return ok ? 0 : 1;
}

/* ---- -i: raw receiver streams through sessions -------------------------------------------------------------------- */
#define MAX_RECEIVERS 64

static int decode_slot(wspr_session **rx, int nrx, const int *buffers, long slot_end_utc, int *total) {
    struct decoder_results *spots = calloc((size_t)nrx * MAX_SPOTS, sizeof *spots);
    int nspots[MAX_RECEIVERS], decoded[MAX_RECEIVERS];
    if (!spots) { fprintf(stderr, "out of memory\n"); return 2; }
    /* every receiver's completed buffer in one call: the decoder thread's body (rtlsdr_wsprd.c:263-328) for all of them */
    const int r = wspr_session_decode_many(rx, buffers, nrx, spots, MAX_SPOTS, nspots, decoded);
    if (r < 0) { fprintf(stderr, "decode failed (%d): no usable MI355X\n", r); free(spots); return 3; }
    int y, mo, d, h, mi;
    wspr_frame_time(slot_end_utc, &y, &mo, &d, &h, &mi);
    for (int k = 0; k < nrx; ++k) {
        char who[16] = "";
        if (nrx > 1) snprintf(who, sizeof who, "[%d] ", k);
        if (!decoded[k]) {
            printf("%sSignal too short, skipping (%u samples)\n", who, wspr_session_fill(rx[k], buffers[k]));
            continue;
        }
        if (nspots[k] == 0) printf("%sNo spot %04d-%02d-%02d %02d:%02dz\n", who, y, mo, d, h, mi);
        for (int s = 0; s < nspots[k]; ++s) {
            char line[160];
            wspr_format_spot_timestamped(&spots[(size_t)k * MAX_SPOTS + s], y, mo, d, h, mi, line, sizeof line);
            printf("%s%s\n", who, line);
        }
        *total += nspots[k];
    }
    fflush(stdout);
    free(spots);
    return 0;
}

static int stream(int nrx, char **paths, struct decoder_options opt, long t_first_slot) {
    FILE *in[MAX_RECEIVERS];
    wspr_session *rx[MAX_RECEIVERS];
    int live[MAX_RECEIVERS];
    for (int k = 0; k < nrx; ++k) {
        in[k] = strcmp(paths[k], "-") == 0 ? stdin : fopen(paths[k], "rb");
        if (!in[k]) { fprintf(stderr, "%s: %s\n", paths[k], strerror(errno)); return 2; }
        rx[k] = wspr_session_create(opt);
        if (!rx[k]) { fprintf(stderr, "no receiver session: no usable MI355X\n"); return 3; }
        live[k] = 1;
    }
    static uint8_t buf[MAX_RECEIVERS][CALLBACK_BYTES];
    unsigned long long in_slot = 0;                 /* bytes of the current slot every live receiver has delivered */
    long slot_end = t_first_slot + 120;
    int rc = 0, total = 0, nlive = nrx;
    while (nlive > 0 && rc == 0) {
        size_t want = CALLBACK_BYTES, fed = 0;
        if (SLOT_BYTES - in_slot < want) want = (size_t)(SLOT_BYTES - in_slot);   /* the slot ends inside this callback */
        wspr_session *full[MAX_RECEIVERS];          /* receivers whose callback is complete: fed together */
        const uint8_t *full_buf[MAX_RECEIVERS];
        int nfull = 0;
        for (int k = 0; k < nrx && rc == 0; ++k) {  /* one callback per receiver (each RX thread's rtlsdr_callback) */
            if (!live[k]) continue;
            const size_t got = fread(buf[k], 1, want, in[k]);
            if (got == want) {
                full[nfull] = rx[k];
                full_buf[nfull++] = buf[k];
                fed = want;
                continue;
            }
            const size_t whole = got & ~(size_t)15;                               /* a stream's last bytes: multiples of 16 */
            if (whole && wspr_session_feed(rx[k], buf[k], (uint32_t)whole) < 0) rc = 3;
            if (whole > fed) fed = whole;
            live[k] = 0;
            --nlive;
        }
        if (rc == 0 && nfull == 1 && wspr_session_feed(full[0], full_buf[0], (uint32_t)want) < 0) rc = 3;
        if (rc == 0 && nfull > 1 && wspr_session_feed_many(full, full_buf, (uint32_t)want, nfull, NULL) < 0) rc = 3;
        in_slot += fed;
        if (rc == 0 && (in_slot == SLOT_BYTES || (nlive == 0 && in_slot))) {      /* the even minute, or the streams' end */
            int done[MAX_RECEIVERS];
            for (int k = 0; k < nrx; ++k) done[k] = wspr_session_rollover(rx[k]);
            rc = decode_slot(rx, nrx, done, slot_end, &total);
            in_slot = 0;
            slot_end += 120;
        }
    }
    for (int k = 0; k < nrx; ++k) {
        wspr_session_destroy(rx[k]);
        if (in[k] != stdin) fclose(in[k]);
    }
    fprintf(stderr, "%d spot(s)\n", total);
    return rc;
}

int main(int argc, char **argv) {
    struct decoder_options opt;
    memset(&opt, 0, sizeof opt);
    opt.freq = 14095600;                                    /* 20 m; overridden by -f */
    opt.npasses = 2;                                        /* the reference's defaults, rtlsdr_wsprd.c:357-362 */
    opt.subtraction = 1;
    enum { NONE, PLAYBACK, SELFTEST, STREAM } mode = NONE;
    char *raw[MAX_RECEIVERS];
    int nraw = 0;
    long t0 = 0;
    int first_file = 0;
    for (int a = 1; a < argc; ++a) {
        const char *o = argv[a];
        if (o[0] != '-' || o[1] == '\0' || o[2] != '\0') { usage(argv[0]); return 2; }
        const int wants_value = strchr("fclirT", o[1]) != NULL;
        if (wants_value && a + 1 >= argc) { usage(argv[0]); return 2; }
        switch (o[1]) {
            case 'f': opt.freq = (int)strtod(argv[++a], NULL); break;
            case 'c': snprintf(opt.rcall, sizeof opt.rcall, "%.12s", argv[++a]); break;
            case 'l': snprintf(opt.rloc, sizeof opt.rloc, "%.6s", argv[++a]); break;
            case 'H': opt.usehashtable = 1; break;
            case 'Q': opt.quickmode = 1; break;
            case 'S': opt.subtraction = 0; opt.npasses = 1; break;
            case 't': mode = SELFTEST; break;
            case 'T': t0 = strtol(argv[++a], NULL, 10); break;
            case 'i':                                                           /* one receiver per -i */
                mode = STREAM;
                if (nraw == MAX_RECEIVERS) { fprintf(stderr, "at most %d receivers\n", MAX_RECEIVERS); return 2; }
                raw[nraw++] = argv[++a];
                break;
            case 'r': mode = PLAYBACK; first_file = ++a; a = argc; break;      /* everything behind -r is a file */
            default: usage(argv[0]); return 2;
        }
    }
    if (mode == NONE || (mode == PLAYBACK && first_file >= argc)) { usage(argv[0]); return 2; }
    if (!wspr_device_ready()) { fprintf(stderr, "%s: no usable MI355X (there is no CPU fallback)\n", wspr_mi355x_version()); return 3; }
    switch (mode) {
        case PLAYBACK: return playback(argc - first_file, argv + first_file, opt);
        case SELFTEST: return self_test(opt);
        case STREAM: return stream(nraw, raw, opt, t0);
        default: usage(argv[0]); return 2;
    }
}

/* ============================================================================
 * orc_message.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Message layer of the WSPR decode path, restated from the reference:
 *   callsign hash      wsprd/nhash.c:205-451     (Jenkins lookup3 "hashlittle")
 *   50-bit pack        wsprd/wsprsim_utils.c:16-316
 *   50-bit unpack      wsprd/wsprd_utils.c:40-313
 *   K=32 r=1/2 code    wsprd/fano.c:51-82, fano.h:35-44, tab.c:7
 *   Fano decoder       wsprd/fano.c:87-238
 *   (de)interleaver    wsprd/wsprd_utils.c:196-213, wsprsim_utils.c:144-161
 *
 * Pinned against the real reference objects (oracle/_ref/libwsprd_ref.so, built
 * from the FFTW-free reference sources where they lie) by tests/test_oracle_ref.py
 * and against the reference's own unit-test expectations (tests/test_wsprd.c).
 * ==========================================================================*/
#include "wspr_oracle.h"
#include "orc_tables.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* 162-symbol pseudo-random sync vector (wsprd/wsprd.c:84-93; the same data is
 * repeated at wsprsim_utils.c:167-176), expanded from orc_tables.h at load. */
unsigned char orc_sync_vector[ORC_NSYM];
__attribute__((constructor)) static void sync_expand(void) {
    for (int i = 0; i < ORC_NSYM; i++) orc_sync_vector[i] = (unsigned char)(orc_sync_bits[i] - '0');
}

/* ------------------------------------------------------------------ nhash -- */
#define ROL(x, k) (((x) << (k)) | ((x) >> (32 - (k))))

/* wsprd/nhash.c:205-451: lookup3 hashlittle() over bytes, masked to 15 bits.
 * The reference picks a 32-bit/16-bit/8-bit reader by pointer alignment; all
 * three read the same little-endian words, so one byte-wise reader suffices. */
uint32_t orc_nhash(const void *key, size_t length, uint32_t initval) {
    const uint8_t *k = (const uint8_t *)key;
    uint32_t a, b, c;
    a = b = c = 0xdeadbeefu + (uint32_t)length + initval;
    while (length > 12) {
        a += (uint32_t)k[0] | (uint32_t)k[1] << 8 | (uint32_t)k[2] << 16 | (uint32_t)k[3] << 24;
        b += (uint32_t)k[4] | (uint32_t)k[5] << 8 | (uint32_t)k[6] << 16 | (uint32_t)k[7] << 24;
        c += (uint32_t)k[8] | (uint32_t)k[9] << 8 | (uint32_t)k[10] << 16 | (uint32_t)k[11] << 24;
        a -= c; a ^= ROL(c, 4);  c += b;
        b -= a; b ^= ROL(a, 6);  a += c;
        c -= b; c ^= ROL(b, 8);  b += a;
        a -= c; a ^= ROL(c, 16); c += b;
        b -= a; b ^= ROL(a, 19); a += c;
        c -= b; c ^= ROL(b, 4);  b += a;
        length -= 12;
        k += 12;
    }
    if (length == 0) return c;   /* nhash.c:443-444: unmasked, unmixed */
    uint32_t w[3] = {0, 0, 0};
    for (size_t i = 0; i < length; i++) w[i >> 2] |= (uint32_t)k[i] << (8 * (i & 3));
    a += w[0]; b += w[1]; c += w[2];
    c ^= b; c -= ROL(b, 14);
    a ^= c; a -= ROL(c, 11);
    b ^= a; b -= ROL(a, 25);
    c ^= b; c -= ROL(b, 16);
    a ^= c; a -= ROL(c, 4);
    b ^= a; b -= ROL(a, 14);
    c ^= b; c -= ROL(b, 24);
    return c & 32767u;           /* nhash.c:448 */
}

/* ------------------------------------------------------------- char codes -- */
/* wsprsim_utils.c:28-39 */
char orc_call_char_code(char ch) {
    if (ch >= '0' && ch <= '9') return (char)(ch - '0');
    if (ch == ' ') return 36;
    if (ch >= 'A' && ch <= 'Z') return (char)(ch - 'A' + 10);
    return -1;
}
/* wsprsim_utils.c:16-26: locator letters are A..R only */
char orc_loc_char_code(char ch) {
    if (ch >= '0' && ch <= '9') return (char)(ch - '0');
    if (ch == ' ') return 36;
    if (ch >= 'A' && ch <= 'R') return (char)(ch - 'A');
    return -1;
}

/* wsprsim_utils.c:41-47 (argument = four already-coded characters) */
unsigned long orc_pack_grid4_power(const char *g, int power) {
    unsigned long m;
    m = (unsigned long)((179 - 10 * g[0] - g[2]) * 180 + 10 * g[1] + g[3]);
    m = m * 128 + (unsigned long)power + 64;
    return m;
}

/* wsprsim_utils.c:49-78.  The reference writes one byte past call6[] when a
 * 6-character call has its digit in position 1; that byte never
 * contributes to n, so the write is simply bounded here. */
unsigned long orc_pack_call(const char *callsign) {
    char six[6];
    size_t len = strlen(callsign);
    memset(six, ' ', sizeof six);
    if (len > 6) return 0;
    char c2 = (len >= 2) ? callsign[2] : 0;
    char c1 = (len >= 1) ? callsign[1] : 0;
    if (isdigit((unsigned char)c2)) {
        for (size_t i = 0; i < len; i++) six[i] = callsign[i];
    } else if (isdigit((unsigned char)c1)) {
        for (size_t i = 1; i < len + 1 && i < 6; i++) six[i] = callsign[i - 1];
    }
    for (int i = 0; i < 6; i++) six[i] = orc_call_char_code(six[i]);
    unsigned long n = (unsigned long)six[0];
    n = n * 36 + (unsigned long)six[1];
    n = n * 10 + (unsigned long)six[2];
    n = n * 27 + (unsigned long)six[3] - 10;
    n = n * 27 + (unsigned long)six[4] - 10;
    n = n * 27 + (unsigned long)six[5] - 10;
    return n;
}

static int pfx_char_value(int nc) {
    if (nc >= '0' && nc <= '9') return nc - '0';
    if (nc >= 'A' && nc <= 'Z') return nc - 'A' + 10;
    return -1;
}

/* wsprsim_utils.c:80-142 */
void orc_pack_prefix(char *callsign, int32_t *n, int32_t *m, int32_t *nadd) {
    char base[7] = {0};
    size_t slash = strcspn(callsign, "/");
    if (callsign[slash + 2] == 0) {                      /* CALL/x            */
        for (size_t i = 0; i < slash && i < 6; i++) base[i] = callsign[i];
        *n = (int32_t)orc_pack_call(base);
        *nadd = 1;
        int v = pfx_char_value(callsign[slash + 1]);
        *m = (v >= 0) ? v : 38;
        *m = 60000 - 32768 + *m;
    } else if (callsign[slash + 3] == 0) {               /* CALL/nn           */
        for (size_t i = 0; i < slash && i < 6; i++) base[i] = callsign[i];
        *n = (int32_t)orc_pack_call(base);
        *nadd = 1;
        *m = 10 * (callsign[slash + 1] - '0') + (callsign[slash + 2] - '0');
        *m = 60000 + 26 + *m;
    } else {                                             /* PFX/CALL          */
        const char *pfx  = strtok(callsign, "/");
        const char *call = strtok(NULL, " ");
        *n = (int32_t)orc_pack_call(call ? call : "");
        size_t plen = strlen(pfx);
        if (plen == 1)      *m = 37 * 36 + 36;
        else if (plen == 2) *m = 36;
        else                *m = 0;
        for (size_t i = 0; i < plen; i++) {
            int v = pfx_char_value(callsign[i]);
            if (v < 0) v = 36;
            *m = 37 * (*m) + v;
        }
        *nadd = 0;
        if (*m > 32768) { *m -= 32768; *nadd = 1; }
    }
}

/* ------------------------------------------------------------ interleaver -- */
/* Both directions walk i = 0..255, bit-reverse the 8-bit counter and keep
 * positions < 162 (wsprd_utils.c:196-213 / wsprsim_utils.c:144-161). */
static unsigned char bitrev8(unsigned char v) {
    v = (unsigned char)((v & 0xF0) >> 4 | (v & 0x0F) << 4);
    v = (unsigned char)((v & 0xCC) >> 2 | (v & 0x33) << 2);
    v = (unsigned char)((v & 0xAA) >> 1 | (v & 0x55) << 1);
    return v;
}
void orc_interleave(unsigned char *sym) {
    unsigned char out[ORC_NSYM];
    int p = 0;
    for (int i = 0; p < ORC_NSYM; i++) {
        unsigned char j = bitrev8((unsigned char)i);
        if (j < ORC_NSYM) out[j] = sym[p++];
    }
    memcpy(sym, out, ORC_NSYM);
}
void orc_deinterleave(unsigned char *sym) {
    unsigned char out[ORC_NSYM];
    int p = 0;
    for (int i = 0; p < ORC_NSYM; i++) {
        unsigned char j = bitrev8((unsigned char)i);
        if (j < ORC_NSYM) out[p++] = sym[j];
    }
    memcpy(sym, out, ORC_NSYM);
}

/* ------------------------------------------------- convolutional encoder -- */
#define ORC_POLY_A 0xf2d05351u   /* fano.c:51  Layland-Lushbaugh */
#define ORC_POLY_B 0xe4613c47u   /* fano.c:52 */

static inline unsigned parity32(uint32_t v) {
    v ^= v >> 16; v ^= v >> 8; v ^= v >> 4; v ^= v >> 2; v ^= v >> 1;
    return v & 1u;
}
/* fano.h:35-44: symbol pair = 2*parity(state & POLY1) + parity(state & POLY2)
 * (Partab[] of tab.c:7 is the 8-bit parity function). */
static inline unsigned conv_pair(uint64_t state) {
    uint32_t s = (uint32_t)state;
    return (parity32(s & ORC_POLY_A) << 1) | parity32(s & ORC_POLY_B);
}

/* fano.c:63-82 */
int orc_conv_encode(unsigned char *symbols, const unsigned char *data, unsigned nbytes) {
    uint64_t state = 0;
    for (unsigned b = 0; b < nbytes; b++) {
        for (int bit = 7; bit >= 0; bit--) {
            state = (state << 1) | ((data[b] >> bit) & 1u);
            unsigned pr = conv_pair(state);
            *symbols++ = (unsigned char)(pr >> 1);
            *symbols++ = (unsigned char)(pr & 1);
        }
    }
    return 0;
}

/* wsprd.c:467-473 */
void orc_build_mettab(int mettab[2][256]) {
    float bias = 0.45;
    for (int i = 0; i < 256; i++) {
        mettab[0][i] = (int)roundf(10.0 * (orc_metric_es6db[i] - bias));
        mettab[1][i] = (int)roundf(10.0 * (orc_metric_es6db[255 - i] - bias));
    }
}

/* fano.c:87-238, index-based instead of pointer-walking.  Node arrays are
 * sized for nbits <= 128. */
int orc_fano(unsigned *metric, unsigned *cycles, unsigned *maxnp,
             unsigned char *data, const unsigned char *symbols, unsigned nbits,
             const int mettab[2][256], int delta, unsigned maxcycles) {
    enum { MAXN = 130 };
    uint64_t enc[MAXN];
    long     gam[MAXN];
    int      bm[MAXN][4];
    int      tm[MAXN][2];
    int      br[MAXN];
    if (nbits + 1 > MAXN || nbits < 32) return 0;

    const int last = (int)nbits - 1;
    const int tail = (int)nbits - 31;
    *maxnp = 0;

    for (int k = 0; k <= last; k++) {            /* fano.c:118-124 */
        int a0 = mettab[0][symbols[2 * k]],     a1 = mettab[1][symbols[2 * k]];
        int b0 = mettab[0][symbols[2 * k + 1]], b1 = mettab[1][symbols[2 * k + 1]];
        bm[k][0] = a0 + b0; bm[k][1] = a0 + b1; bm[k][2] = a1 + b0; bm[k][3] = a1 + b1;
    }

    int pos = 0;
    enc[0] = 0;
    {
        unsigned ls = conv_pair(enc[0]);
        int m0 = bm[0][ls], m1 = bm[0][3 ^ ls];
        if (m0 > m1) { tm[0][0] = m0; tm[0][1] = m1; }
        else         { tm[0][0] = m1; tm[0][1] = m0; enc[0]++; }
    }
    br[0] = 0;
    const unsigned limit = maxcycles * nbits;
    gam[0] = 0;
    int t = 0;
    unsigned i;
    for (i = 1; i <= limit; i++) {
        if (pos > (int)*maxnp) *maxnp = (unsigned)pos;
        int ng = (int)(gam[pos] + tm[pos][br[pos]]);
        if (ng >= t) {
            if (gam[pos] < t + delta)
                while (ng >= t + delta) t += delta;
            gam[pos + 1] = ng;
            enc[pos + 1] = enc[pos] << 1;
            pos++;
            if (pos == last + 1) break;
            unsigned ls = conv_pair(enc[pos]);
            if (pos >= tail) {
                tm[pos][0] = bm[pos][ls];
            } else {
                int m0 = bm[pos][ls], m1 = bm[pos][3 ^ ls];
                if (m0 > m1) { tm[pos][0] = m0; tm[pos][1] = m1; }
                else         { tm[pos][0] = m1; tm[pos][1] = m0; enc[pos]++; }
            }
            br[pos] = 0;
            continue;
        }
        for (;;) {
            if (pos == 0 || gam[pos - 1] < t) {
                t -= delta;
                if (br[pos] != 0) { br[pos] = 0; enc[pos] ^= 1; }
                break;
            }
            pos--;
            if (pos < tail && br[pos] != 1) { br[pos]++; enc[pos] ^= 1; break; }
        }
    }
    *metric = (unsigned)gam[pos];
    for (unsigned k = 0, nb = nbits >> 3; k < nb; k++) data[k] = (unsigned char)enc[7 + 8 * k];
    *cycles = i + 1;
    return (i >= limit) ? -1 : 0;
}

/* ----------------------------------------------------------------- unpack -- */
/* wsprd_utils.c:40-71 */
void orc_unpack50(const signed char *dat, int32_t *n1, int32_t *n2) {
    uint32_t b[7];
    for (int i = 0; i < 7; i++) b[i] = (uint32_t)(unsigned char)dat[i];
    *n1 = (int32_t)((b[0] << 20) + (b[1] << 12) + (b[2] << 4) + ((b[3] >> 4) & 15));
    *n2 = (int32_t)(((b[3] & 15) << 18) + (b[4] << 10) + (b[5] << 2) + ((b[6] >> 6) & 3));
}

static const char k37[] = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ ";

/* wsprd_utils.c:73-118 */
int orc_unpackcall(int32_t ncall, char *call) {
    char tmp[7];
    int32_t n = ncall;
    snprintf(call, 13, "......");
    if (!(n < 262177560)) return 0;
    tmp[5] = k37[n % 27 + 10]; n /= 27;
    tmp[4] = k37[n % 27 + 10]; n /= 27;
    tmp[3] = k37[n % 27 + 10]; n /= 27;
    tmp[2] = k37[n % 10];      n /= 10;
    tmp[1] = k37[n % 36];      n /= 36;
    tmp[0] = k37[n];
    tmp[6] = '\0';
    int lead = 0;
    while (lead < 5 && tmp[lead] == ' ') lead++;
    snprintf(call, 13, "%-6s", &tmp[lead]);
    for (int i = 0; i < 6; i++)
        if (call[i] == ' ') call[i] = '\0';
    return 1;
}

/* wsprd_utils.c:120-150 */
int orc_unpackgrid(int32_t ngrid, char *grid) {
    ngrid = ngrid >> 7;
    if (!(ngrid < 32400)) {
        snprintf(grid, 5, "XXXX");
        return 0;
    }
    int dlat  = (ngrid % 180) - 90;
    int dlong = (ngrid / 180) * 2 - 180 + 2;
    if (dlong < -180) dlong += 360;
    if (dlong > 180)  dlong += 360;
    int nlong = 60.0 * (180.0 - dlong) / 5.0;
    int hi = nlong / 240, lo = (nlong - 240 * hi) / 24;
    grid[0] = k37[10 + hi];
    grid[2] = k37[lo];
    int nlat = 60.0 * (dlat + 90) / 2.5;
    hi = nlat / 240; lo = (nlat - 240 * hi) / 24;
    grid[1] = k37[10 + hi];
    grid[3] = k37[lo];
    return 1;
}

/* wsprd_utils.c:152-194 */
int orc_unpackpfx(int32_t nprefix, char *call) {
    char nc, pfx[4] = {0}, keep[13];
    snprintf(keep, sizeof keep, "%s", call);
    if (nprefix < 60000) {
        int32_t n = nprefix;
        for (int i = 2; i >= 0; i--) {
            nc = (char)(n % 37);
            if (nc >= 0 && nc <= 9)        pfx[i] = (char)(nc + '0');
            else if (nc >= 10 && nc <= 35) pfx[i] = (char)(nc + 'A' - 10);
            else                           pfx[i] = ' ';
            n /= 37;
        }
        char *sp = strrchr(pfx, ' ');
        snprintf(call, 13, "%s/%s", sp ? sp + 1 : pfx, keep);
    } else {
        nc = (char)(nprefix - 60000);
        if (nc >= 0 && nc <= 9)
            snprintf(call, 13, "%s/%c", keep, nc + '0');
        else if (nc >= 10 && nc <= 35)
            snprintf(call, 13, "%s/%c", keep, nc + 'A' - 10);
        else if (nc >= 36 && nc <= 125)
            snprintf(call, 13, "%s/%c%c", keep, (nc - 26) / 10 + '0', (nc - 26) % 10 + '0');
        else
            return 0;
    }
    return 1;
}

static int power_digit_ok(int v) { int u = v % 10; return u == 0 || u == 3 || u == 7; }

/* wsprd_utils.c:228-313 */
int orc_unpk(const signed char *message, char *hashtab, char *loctab,
             char *call_loc_pow, char *call, char *loc, char *pwr, char *callsign) {
    int32_t n1, n2;
    int noprint = 0;
    char grid[5], grid6[7], cdbm[4];

    orc_unpack50(message, &n1, &n2);
    if (!orc_unpackcall(n1, callsign)) return 1;
    if (!orc_unpackgrid(n2, grid)) return 1;
    int ntype = (n2 & 127) - 64;
    callsign[12] = 0;
    grid[4] = 0;

    if (ntype >= 0 && ntype <= 62) {
        int nu = ntype % 10;
        if (nu == 0 || nu == 3 || nu == 7) {                 /* type 1 */
            snprintf(cdbm, sizeof cdbm, "%02d", ntype);
            snprintf(call_loc_pow, 23, "%s %s %s", callsign, grid, cdbm);
            uint32_t h = orc_nhash(callsign, strlen(callsign), 146u);
            snprintf(hashtab + h * ORC_HASH_W, ORC_HASH_W, "%s", callsign);
            snprintf(loctab + h * ORC_LOC_W, ORC_LOC_W, "%s", grid);
            snprintf(call, ORC_HASH_W, "%s", callsign);
            snprintf(loc, 7, "%s", grid);
            snprintf(pwr, 3, "%s", cdbm);
        } else {                                             /* type 2 */
            int nadd = nu;
            if (nu > 3) nadd = nu - 3;
            if (nu > 7) nadd = nu - 7;
            int32_t n3 = n2 / 128 + ORC_HASH_N * (nadd - 1);
            if (!orc_unpackpfx(n3, callsign)) return 1;
            int ndbm = ntype - nadd;
            snprintf(cdbm, sizeof cdbm, "%2d", ndbm);
            snprintf(call_loc_pow, 23, "%s %s", callsign, cdbm);
            if (power_digit_ok(ndbm)) {
                uint32_t h = orc_nhash(callsign, strlen(callsign), 146u);
                snprintf(hashtab + h * ORC_HASH_W, ORC_HASH_W, "%s", callsign);
            } else {
                noprint = 1;
            }
        }
    } else if (ntype < 0) {                                  /* type 3 */
        int ndbm = -(ntype + 1);
        memset(grid6, 0, sizeof grid6);
        snprintf(grid6, sizeof grid6, "%c%.*s", callsign[5], 5, callsign);
        if (!power_digit_ok(ndbm) ||
            !isalpha((unsigned char)grid6[0]) || !isalpha((unsigned char)grid6[1]) ||
            !isdigit((unsigned char)grid6[2]) || !isdigit((unsigned char)grid6[3]))
            noprint = 1;
        int ih = (n2 - ntype - 64) / 128;
        if (hashtab[ih * ORC_HASH_W] != '\0')
            snprintf(callsign, ORC_HASH_W, "<%s>", hashtab + ih * ORC_HASH_W);
        else
            snprintf(callsign, ORC_HASH_W, "<...>");
        snprintf(cdbm, sizeof cdbm, "%2d", ndbm);
        snprintf(call_loc_pow, 23, "%s %s %s", callsign, grid6, cdbm);
        snprintf(call, ORC_HASH_W, "%s", callsign);
        snprintf(loc, 7, "%s", grid6);
        snprintf(pwr, 3, "%s", cdbm);
        if (ntype == -64) noprint = 1;
    }
    return noprint;
}

/* ------------------------------------------------------- channel symbols -- */
static const int power_round[10] = {0, -1, 1, 0, -1, 2, 1, 0, -1, 1};

/* wsprsim_utils.c:163-316 */
int orc_channel_symbols(const char *rawmessage, char *hashtab, char *loctab,
                        unsigned char *symbols) {
    char msg[23];
    memset(msg, 0, sizeof msg);
    /* the reference copies up to 23 characters (no terminator kept at 23);
     * a decoder-produced string is at most 22, so bound to 22 here */
    for (int i = 0; i < 22 && rawmessage[i]; i++) msg[i] = rawmessage[i];

    size_t sp = strcspn(msg, " "), sl = strcspn(msg, "/");
    size_t lt = strcspn(msg, "<"), gt = strcspn(msg, ">");
    size_t len = strlen(msg);
    unsigned long n = 0;
    int m = 0;

    if (sp > 3 && sp < 7 && sl == len && lt == len) {            /* type 1 */
        char *cs = strtok(msg, " ");
        char *gr = strtok(NULL, " ");
        char *pw = strtok(NULL, " ");
        if (!cs || !gr || !pw) return 0;       /* reference would dereference NULL */
        int power = atoi(pw);
        n = orc_pack_call(cs);
        char g4[4];
        for (int i = 0; i < 4; i++) g4[i] = orc_loc_char_code(gr[i]);
        m = (int)orc_pack_grid4_power(g4, power);
    } else if (lt == 0 && gt < len) {                            /* type 3 */
        char *cs = strtok(msg, "<> ");
        char *gr = strtok(NULL, " ");
        char *pw = strtok(NULL, " ");
        if (!cs || !gr || !pw) return 0;
        int power = atoi(pw);
        if (power < 0) power = 0;
        if (power > 60) power = 60;
        power += power_round[power % 10];
        int ntype = -(power + 1);
        int ih = (int)orc_nhash(cs, strlen(cs), 146u);
        m = 128 * ih + ntype + 64;
        char g6[7];
        memset(g6, 0, sizeof g6);
        int gl = (int)strlen(gr);
        for (int i = 0; i < gl - 1 && i < 6; i++) g6[i] = gr[i + 1];
        g6[5] = gr[0];
        n = orc_pack_call(g6);
    } else if (sl < len) {                                       /* type 2 */
        char *cs = strtok(msg, " ");
        if (sl == 0 || sl > strlen(cs)) return 0;
        char *pw = strtok(NULL, " ");
        if (!pw) return 0;
        int power = atoi(pw);
        if (power < 0) power = 0;
        if (power > 60) power = 60;
        power += power_round[power % 10];
        int32_t n1, ng, nadd;
        orc_pack_prefix(cs, &n1, &ng, &nadd);
        int ntype = power + 1 + nadd;
        m = 128 * ng + ntype + 64;
        n = (unsigned long)(long)n1;
    } else {
        return 0;
    }

    unsigned char data[11];
    memset(data, 0, sizeof data);
    data[0] = (unsigned char)(n >> 20);
    data[1] = (unsigned char)(n >> 12);
    data[2] = (unsigned char)(n >> 4);
    data[3] = (unsigned char)(((n & 0x0F) << 4) + ((m >> 18) & 0x0F));
    data[4] = (unsigned char)(m >> 10);
    data[5] = (unsigned char)(m >> 2);
    data[6] = (unsigned char)((m & 0x03) << 6);

    /* wsprsim_utils.c:280-300: the reference unpacks its own packing purely for
     * the hash-table side effect */
    {
        char clp[23], cs13[13], c13[13], l7[7], p3[3];
        signed char chk[11];
        memcpy(chk, data, 11);
        orc_unpk(chk, hashtab, loctab, clp, c13, l7, p3, cs13);
    }

    unsigned char bits[176];
    memset(bits, 0, sizeof bits);
    orc_conv_encode(bits, data, 11);
    orc_interleave(bits);
    const unsigned char *sv = orc_sync_vector;
    for (int i = 0; i < ORC_NSYM; i++) symbols[i] = (unsigned char)(2 * bits[i] + sv[i]);
    return 1;
}

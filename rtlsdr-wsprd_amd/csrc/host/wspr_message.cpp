// Host-side message layer (see wspr_message.h).  Written from the protocol
// description and the observable behaviour of the reference functions cited at
// each routine; results are bit-identical on every input the decoder can
// produce (tests/test_message_layer.py checks them against vectors produced by
// the reference objects and against the CPU oracle).
#include "wspr_message.h"

#include <vector>

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "wspr_tables.h"

namespace wspr {

// ---------------------------------------------------------------- sync vector
const unsigned char* sync_vector() {
    static unsigned char v[kNSym];
    static const bool once = [] {
        for (int i = 0; i < kNSym; ++i) v[i] = static_cast<unsigned char>(kSyncBits[i] - '0');
        return true;
    }();
    (void)once;
    return v;
}

// ---------------------------------------------------------------- callsign hash
// Jenkins lookup3 "hashlittle" reduced to 15 bits (reference wsprd/nhash.c:205-451).
namespace {
inline uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
inline uint32_t le32(const uint8_t* p, size_t avail) {
    uint32_t w = 0;
    for (size_t i = 0; i < 4 && i < avail; ++i) w |= static_cast<uint32_t>(p[i]) << (8 * i);
    return w;
}
}  // namespace

uint32_t nhash15(const void* key, size_t len, uint32_t seed) {
    const uint8_t* p = static_cast<const uint8_t*>(key);
    uint32_t a = 0xdeadbeefu + static_cast<uint32_t>(len) + seed, b = a, c = a;
    for (; len > 12; len -= 12, p += 12) {
        a += le32(p, 4); b += le32(p + 4, 4); c += le32(p + 8, 4);
        a -= c; a ^= rotl(c, 4);  c += b;
        b -= a; b ^= rotl(a, 6);  a += c;
        c -= b; c ^= rotl(b, 8);  b += a;
        a -= c; a ^= rotl(c, 16); c += b;
        b -= a; b ^= rotl(a, 19); a += c;
        c -= b; c ^= rotl(b, 4);  b += a;
    }
    if (len == 0) return c;
    a += le32(p, len);
    if (len > 4) b += le32(p + 4, len - 4);
    if (len > 8) c += le32(p + 8, len - 8);
    c ^= b; c -= rotl(b, 14);
    a ^= c; a -= rotl(c, 11);
    b ^= a; b -= rotl(a, 25);
    c ^= b; c -= rotl(b, 16);
    a ^= c; a -= rotl(c, 4);
    b ^= a; b -= rotl(a, 14);
    c ^= b; c -= rotl(b, 24);
    return c & 0x7fffu;
}

// ---------------------------------------------------------------- source coding
// wsprsim_utils.c:16-39
char callsign_code(char ch) {
    if (ch >= '0' && ch <= '9') return static_cast<char>(ch - '0');
    if (ch >= 'A' && ch <= 'Z') return static_cast<char>(ch - 'A' + 10);
    return ch == ' ' ? 36 : -1;
}
char locator_code(char ch) {
    if (ch >= '0' && ch <= '9') return static_cast<char>(ch - '0');
    if (ch >= 'A' && ch <= 'R') return static_cast<char>(ch - 'A');
    return ch == ' ' ? 36 : -1;
}

// wsprsim_utils.c:41-47
unsigned long pack_grid_power(const char* g, int power) {
    unsigned long m = static_cast<unsigned long>((179 - 10 * g[0] - g[2]) * 180 + 10 * g[1] + g[3]);
    return m * 128 + static_cast<unsigned long>(power) + 64;
}

// wsprsim_utils.c:49-78.  A callsign is right-aligned so that its digit lands in
// column 2 of a 6-column field; radix 37*36*10*27*27*27.
unsigned long pack_callsign(const char* call) {
    const size_t len = std::strlen(call);
    if (len > 6) return 0;
    char f[6] = {' ', ' ', ' ', ' ', ' ', ' '};
    const bool digit_at_2 = len >= 2 && std::isdigit(static_cast<unsigned char>(call[2]));
    const bool digit_at_1 = len >= 1 && std::isdigit(static_cast<unsigned char>(call[1]));
    if (digit_at_2)      std::memcpy(f, call, len);
    else if (digit_at_1) std::memcpy(f + 1, call, len < 5 ? len : 5);
    static const int radix[6] = {1, 36, 10, 27, 27, 27};
    static const int bias[6]  = {0, 0, 0, 10, 10, 10};
    unsigned long n = 0;
    for (int i = 0; i < 6; ++i) {
        const unsigned long code = static_cast<unsigned long>(static_cast<long>(callsign_code(f[i])));
        n = (i == 0) ? code : n * radix[i] + code - bias[i];
    }
    return n;
}

namespace {
inline int alnum36(int ch) {
    if (ch >= '0' && ch <= '9') return ch - '0';
    if (ch >= 'A' && ch <= 'Z') return ch - 'A' + 10;
    return -1;
}
// strtok()-compatible splitter without hidden global state
struct Splitter {
    char* cur;
    explicit Splitter(char* s) : cur(s) {}
    char* next(const char* delims) {
        if (!cur) return nullptr;
        cur += std::strspn(cur, delims);
        if (!*cur) { cur = nullptr; return nullptr; }
        char* tok = cur;
        cur += std::strcspn(cur, delims);
        if (*cur) { *cur = '\0'; ++cur; } else { cur = nullptr; }
        return tok;
    }
};
}  // namespace

// wsprsim_utils.c:80-142: compound callsigns (one/two character suffix or a
// 1..3 character prefix) are carried in the 15 "grid" bits plus nadd.
void pack_compound(char* call, int32_t* n, int32_t* m, int32_t* nadd) {
    const size_t slash = std::strcspn(call, "/");
    char base[7] = {0};
    if (call[slash + 2] == '\0') {                       // CALL/x
        std::memcpy(base, call, slash < 6 ? slash : 6);
        *n = static_cast<int32_t>(pack_callsign(base));
        *nadd = 1;
        const int v = alnum36(call[slash + 1]);
        *m = 60000 - 32768 + (v >= 0 ? v : 38);
    } else if (call[slash + 3] == '\0') {                // CALL/nn
        std::memcpy(base, call, slash < 6 ? slash : 6);
        *n = static_cast<int32_t>(pack_callsign(base));
        *nadd = 1;
        *m = 60000 + 26 + 10 * (call[slash + 1] - '0') + (call[slash + 2] - '0');
    } else {                                             // PFX/CALL
        Splitter sp(call);
        const char* pfx = sp.next("/");
        const char* rest = sp.next(" ");
        *n = static_cast<int32_t>(pack_callsign(rest ? rest : ""));
        const size_t plen = pfx ? std::strlen(pfx) : 0;
        int32_t acc = plen == 1 ? 37 * 36 + 36 : (plen == 2 ? 36 : 0);
        for (size_t i = 0; i < plen; ++i) {
            const int v = alnum36(pfx[i]);
            acc = 37 * acc + (v >= 0 ? v : 36);
        }
        *nadd = 0;
        if (acc > 32768) { acc -= 32768; *nadd = 1; }
        *m = acc;
    }
}

// ---------------------------------------------------------------- interleaver
// Symbol p (in transmission order 0..161) sits at the p-th value of the
// bit-reversed 8-bit counter that is < 162 (wsprd_utils.c:196-213,
// wsprsim_utils.c:144-161).
namespace {
struct Permutation {
    unsigned char at[kNSym];
    Permutation() {
        int p = 0;
        for (int i = 0; i < 256 && p < kNSym; ++i) {
            unsigned r = 0;
            for (int b = 0; b < 8; ++b) r |= ((i >> b) & 1u) << (7 - b);
            if (r < static_cast<unsigned>(kNSym)) at[p++] = static_cast<unsigned char>(r);
        }
    }
};
const Permutation& perm() { static const Permutation p; return p; }
}  // namespace

void interleave162(unsigned char* sym) {
    unsigned char tmp[kNSym];
    for (int p = 0; p < kNSym; ++p) tmp[perm().at[p]] = sym[p];
    std::memcpy(sym, tmp, kNSym);
}
void deinterleave162(unsigned char* sym) {
    unsigned char tmp[kNSym];
    for (int p = 0; p < kNSym; ++p) tmp[p] = sym[perm().at[p]];
    std::memcpy(sym, tmp, kNSym);
}

// ---------------------------------------------------------------- K=32 r=1/2 code
// Layland-Lushbaugh polynomials (fano.c:51-52); output pair = (parity(s&G1), parity(s&G2)),
// fano.h:35-44.
namespace {
constexpr uint32_t kG1 = 0xf2d05351u, kG2 = 0xe4613c47u;
inline unsigned branch_pair(uint32_t state) {
    return (static_cast<unsigned>(__builtin_parity(state & kG1)) << 1) |
           static_cast<unsigned>(__builtin_parity(state & kG2));
}
}  // namespace

int conv_encode(unsigned char* out, const unsigned char* data, unsigned nbytes) {
    uint32_t state = 0;
    for (unsigned b = 0; b < nbytes; ++b)
        for (int bit = 7; bit >= 0; --bit) {
            state = (state << 1) | ((data[b] >> bit) & 1u);
            const unsigned pr = branch_pair(state);
            *out++ = static_cast<unsigned char>(pr >> 1);
            *out++ = static_cast<unsigned char>(pr & 1u);
        }
    return 0;
}

FanoMetrics::FanoMetrics() {
    for (int i = 0; i < 256; ++i) {
        tab[0][i] = kFanoMetric[i];
        tab[1][i] = kFanoMetric[255 - i];
    }
}
const FanoMetrics& default_metrics() { static const FanoMetrics m; return m; }

// Fano sequential decoder (reference wsprd/fano.c:87-238; same search order,
// threshold schedule, cycle accounting and return convention).
//
// The search is a serial walk of up to 810 000 steps for an undecodable vector, and such
// time-outs dominate the host cost of crowded bands, so the walk is written for speed:
// structure-of-arrays node storage, the encoder output of a node computed once when the node is
// entered (both branch metrics are kept sorted), parity by popcount.
int fano_decode(unsigned* metric, unsigned* cycles, unsigned* maxnp, unsigned char* data,
                const unsigned char* symbols, unsigned nbits, const int mettab[2][256],
                int delta, unsigned maxcycles) {
    constexpr unsigned kMax = 128;
    if (nbits < 32 || nbits > kMax) return 0;
    uint32_t state[kMax + 2];      // encoder state including the hypothesised bit
    int gamma[kMax + 2];           // path metric up to the node
    int best[kMax + 2];            // larger branch metric (tail nodes: the 0-branch metric)
    int second[kMax + 2];          // smaller branch metric
    unsigned char pick[kMax + 2];  // 0: exploring the better branch, 1: the other
    int bm[kMax + 1][4];
    const int last = static_cast<int>(nbits) - 1;
    const int tail = static_cast<int>(nbits) - 31;   // the last 31 bits are forced to 0

    for (int k = 0; k <= last; ++k) {
        const int a0 = mettab[0][symbols[2 * k]], a1 = mettab[1][symbols[2 * k]];
        const int b0 = mettab[0][symbols[2 * k + 1]], b1 = mettab[1][symbols[2 * k + 1]];
        bm[k][0] = a0 + b0; bm[k][1] = a0 + b1; bm[k][2] = a1 + b0; bm[k][3] = a1 + b1;
    }
    auto enter = [&](int pos, uint32_t st) {           // rank the two branches of a fresh node
        const unsigned zp = branch_pair(st);
        const int m0 = bm[pos][zp];
        if (pos >= tail) { best[pos] = m0; second[pos] = m0; state[pos] = st; pick[pos] = 0; return; }
        const int m1 = bm[pos][3 ^ zp];
        if (m0 > m1) { best[pos] = m0; second[pos] = m1; state[pos] = st; }
        else         { best[pos] = m1; second[pos] = m0; state[pos] = st | 1u; }
        pick[pos] = 0;
    };

    int pos = 0, deepest = 0, threshold = 0;
    gamma[0] = 0;
    enter(0, 0u);
    const unsigned budget = maxcycles * nbits;
    unsigned it;
    for (it = 1; it <= budget; ++it) {
        if (pos > deepest) deepest = pos;
        const int g = gamma[pos];
        const int ahead = g + (pick[pos] ? second[pos] : best[pos]);
        if (ahead >= threshold) {
            if (g < threshold + delta)
                while (ahead >= threshold + delta) threshold += delta;
            gamma[pos + 1] = ahead;
            const uint32_t nst = state[pos] << 1;
            if (++pos == last + 1) { state[pos] = nst; break; }
            enter(pos, nst);
            continue;
        }
        for (;;) {
            if (pos == 0 || gamma[pos - 1] < threshold) {
                threshold -= delta;
                if (pick[pos] != 0) { pick[pos] = 0; state[pos] ^= 1u; }
                break;
            }
            --pos;
            if (pos < tail && pick[pos] != 1) { pick[pos] = 1; state[pos] ^= 1u; break; }
        }
    }
    *maxnp = static_cast<unsigned>(deepest);
    *metric = static_cast<unsigned>(gamma[pos]);
    for (unsigned k = 0; k < (nbits >> 3); ++k) data[k] = static_cast<unsigned char>(state[7 + 8 * k]);
    *cycles = it + 1;
    return it >= budget ? -1 : 0;
}

// ---------------------------------------------------------------- unpacking
// wsprd_utils.c:40-71: 28-bit callsign field n1, 22-bit grid/power field n2
void unpack_50bits(const signed char* dat, int32_t* n1, int32_t* n2) {
    uint64_t bits = 0;
    for (int i = 0; i < 7; ++i) bits = (bits << 8) | static_cast<unsigned char>(dat[i]);
    bits >>= 6;                                            // 50 payload bits
    *n1 = static_cast<int32_t>(bits >> 22);
    *n2 = static_cast<int32_t>(bits & 0x3fffff);
}

namespace { const char kAlphabet[] = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ "; }

// wsprd_utils.c:73-118
int unpack_callsign(int32_t ncall, char* call) {
    if (ncall >= 262177560) { std::memcpy(call, "......", 7); return 0; }
    char f[7];
    int32_t n = ncall;
    for (int i = 5; i >= 3; --i) { f[i] = kAlphabet[n % 27 + 10]; n /= 27; }
    f[2] = kAlphabet[n % 10]; n /= 10;
    f[1] = kAlphabet[n % 36]; n /= 36;
    f[0] = kAlphabet[n];
    f[6] = '\0';
    int lead = 0;
    while (lead < 5 && f[lead] == ' ') ++lead;
    // the field without its leading blanks, left-justified in six columns ("%-6s"), blanks then turned into NULs
    // (a decode's hot path on the host: written out instead of two snprintf)
    for (int i = 0; i < 6; ++i) {
        const char ch = i < 6 - lead ? f[lead + i] : ' ';
        call[i] = ch == ' ' ? '\0' : ch;
    }
    call[6] = '\0';
    return 1;
}

// wsprd_utils.c:120-150
int unpack_grid(int32_t ngrid, char* grid) {
    ngrid >>= 7;
    if (ngrid >= 32400) { std::snprintf(grid, 5, "XXXX"); return 0; }
    const int dlat = ngrid % 180 - 90;
    int dlong = (ngrid / 180) * 2 - 180 + 2;
    if (dlong < -180) dlong += 360;
    if (dlong > 180)  dlong += 360;
    const int nlong = static_cast<int>(60.0 * (180.0 - dlong) / 5.0);
    const int nlat  = static_cast<int>(60.0 * (dlat + 90) / 2.5);
    grid[0] = kAlphabet[10 + nlong / 240];
    grid[1] = kAlphabet[10 + nlat / 240];
    grid[2] = kAlphabet[(nlong % 240) / 24];
    grid[3] = kAlphabet[(nlat % 240) / 24];
    return 1;
}

// wsprd_utils.c:152-194
int unpack_prefix(int32_t nprefix, char* call) {
    char base[13];
    std::snprintf(base, sizeof base, "%s", call);
    if (nprefix < 60000) {
        char pfx[4] = {0, 0, 0, 0};
        int32_t n = nprefix;
        for (int i = 2; i >= 0; --i, n /= 37) {
            const int d = n % 37;
            pfx[i] = d <= 9 ? static_cast<char>('0' + d) : (d <= 35 ? static_cast<char>('A' + d - 10) : ' ');
        }
        const char* sp = std::strrchr(pfx, ' ');
        std::snprintf(call, 13, "%s/%s", sp ? sp + 1 : pfx, base);
        return 1;
    }
    const char d = static_cast<char>(nprefix - 60000);     // the reference narrows to char here
    if (d >= 0 && d <= 9)         std::snprintf(call, 13, "%s/%c", base, '0' + d);
    else if (d >= 10 && d <= 35)  std::snprintf(call, 13, "%s/%c", base, 'A' + d - 10);
    else if (d >= 36 && d <= 125) std::snprintf(call, 13, "%s/%c%c", base, '0' + (d - 26) / 10, '0' + (d - 26) % 10);
    else return 0;
    return 1;
}

namespace {
inline bool legal_power(int dbm) { const int u = dbm % 10; return u == 0 || u == 3 || u == 7; }
// a look-up view whose answers are not recorded (see HashTable::peek)
struct QuietView : HashTable {
    HashTable& t;
    explicit QuietView(HashTable& t_) : t(t_) {}
    const char* call_at(int slot) override { return t.peek(slot); }
    const char* peek(int slot) override { return t.peek(slot); }
    void put(int slot, const char* call, const char* grid) override { t.put(slot, call, grid); }
};
}  // namespace

void FlatHashTable::put(int slot, const char* call, const char* grid) {
    copy_text(hashtab + (size_t)slot * kHashWidth, kHashWidth, call);
    if (grid) copy_text(loctab + (size_t)slot * kLocWidth, kLocWidth, grid);
    if (dirty_vec) static_cast<std::vector<int>*>(dirty_vec)->push_back(slot);
}

int unpack_message(const signed char* msg, char* hashtab, char* loctab, char* call_loc_pow,
                   char* call, char* loc, char* pwr, char* callsign) {
    FlatHashTable tab(hashtab, loctab);
    return unpack_message(msg, tab, call_loc_pow, call, loc, pwr, callsign);
}
int channel_symbols(const char* text, char* hashtab, char* loctab, unsigned char* symbols) {
    FlatHashTable tab(hashtab, loctab);
    return channel_symbols(text, tab, symbols);
}

// wsprd_utils.c:228-313.  Returns the reference's "noprint" flag.
int unpack_message(const signed char* msg, HashTable& tab, char* call_loc_pow,
                   char* call, char* loc, char* pwr, char* callsign) {
    int32_t n1, n2;
    unpack_50bits(msg, &n1, &n2);
    char grid[5];
    if (!unpack_callsign(n1, callsign)) return 1;
    if (!unpack_grid(n2, grid)) return 1;
    const int ntype = (n2 & 127) - 64;
    callsign[12] = '\0';
    grid[4] = '\0';
    char dbm_txt[4];
    int noprint = 0;

    if (ntype >= 0 && ntype <= 62) {
        if (legal_power(ntype)) {                                  // type 1: CALL GRID dBm
            // "%02d" of 0..62, "%s %s %s", three "%s" copies: every decode of a plain message runs through here, so the
            // texts are put together by hand (same bytes up to and including each terminating NUL)
            dbm_txt[0] = static_cast<char>('0' + ntype / 10); dbm_txt[1] = static_cast<char>('0' + ntype % 10); dbm_txt[2] = '\0';
            const size_t cl = std::strlen(callsign);               // <= 12 (callsign[12] = 0 above); grid: 4 characters
            char* o = call_loc_pow;
            std::memcpy(o, callsign, cl); o += cl; *o++ = ' ';
            std::memcpy(o, grid, 4); o += 4; *o++ = ' ';
            o[0] = dbm_txt[0]; o[1] = dbm_txt[1]; o[2] = '\0';
            tab.put((int)nhash15(callsign, cl, 146u), callsign, grid);
            std::memcpy(call, callsign, cl + 1);
            std::memcpy(loc, grid, 5);
            std::memcpy(pwr, dbm_txt, 3);
        } else {                                                   // type 2: compound call + dBm
            const int nu = ntype % 10;
            const int nadd = nu > 7 ? nu - 7 : (nu > 3 ? nu - 3 : nu);
            if (!unpack_prefix(n2 / 128 + kHashSlots * (nadd - 1), callsign)) return 1;
            const int dbm = ntype - nadd;
            std::snprintf(dbm_txt, sizeof dbm_txt, "%2d", dbm);
            std::snprintf(call_loc_pow, 23, "%s %s", callsign, dbm_txt);
            if (legal_power(dbm)) tab.put((int)nhash15(callsign, std::strlen(callsign), 146u), callsign, nullptr);
            else noprint = 1;
        }
    } else if (ntype < 0) {                                        // type 3: <hash> GRID6 dBm
        const int dbm = -(ntype + 1);
        char grid6[7];
        std::memset(grid6, 0, sizeof grid6);
        std::snprintf(grid6, sizeof grid6, "%c%.*s", callsign[5], 5, callsign);
        if (!legal_power(dbm) || !std::isalpha(static_cast<unsigned char>(grid6[0])) ||
            !std::isalpha(static_cast<unsigned char>(grid6[1])) ||
            !std::isdigit(static_cast<unsigned char>(grid6[2])) ||
            !std::isdigit(static_cast<unsigned char>(grid6[3])))
            noprint = 1;
        const int slot = (n2 - ntype - 64) / 128;
        const char* known = tab.call_at(slot);
        if (known[0] != '\0')
            std::snprintf(callsign, kHashWidth, "<%s>", known);
        else
            std::snprintf(callsign, kHashWidth, "<...>");
        std::snprintf(dbm_txt, sizeof dbm_txt, "%2d", dbm);
        std::snprintf(call_loc_pow, 23, "%s %s %s", callsign, grid6, dbm_txt);
        std::snprintf(call, kHashWidth, "%s", callsign);
        std::snprintf(loc, 7, "%s", grid6);
        std::snprintf(pwr, 3, "%s", dbm_txt);
        if (ntype == -64) noprint = 1;
    }
    return noprint;
}

// ---------------------------------------------------------------- channel symbols
namespace {
// encode() + interleave() + "2 * bit + sync" (wsprsim_utils.c:302-309) of the 11 packed bytes.  Both the convolutional
// code and the interleaver are linear over GF(2): the 162 interleaved code bits are the XOR of one fixed 162-bit pattern
// per set data bit (the patterns are made once by the plain routines above), and eight bits at a time become eight
// symbols through a 256-entry table.  Same 162 bytes as the three loops, a fifth of their time -- every decoded message
// is encoded again for the subtraction.
struct SymbolCoder {
    uint64_t pattern[88][3];          // data bit i (byte i / 8, MSB first) -> interleaved code bits, bit k of word k / 64
    uint64_t spread[256];             // 8 code bits -> 8 bytes of value 2 * bit
    unsigned char sync_bytes[168];    // sync vector, padded to whole words
    SymbolCoder() {
        for (int i = 0; i < 88; ++i) {
            unsigned char data[11] = {0}, bits[176];
            data[i >> 3] = static_cast<unsigned char>(0x80u >> (i & 7));
            std::memset(bits, 0, sizeof bits);
            conv_encode(bits, data, 11);
            interleave162(bits);
            pattern[i][0] = pattern[i][1] = pattern[i][2] = 0;
            for (int k = 0; k < kNSym; ++k) pattern[i][k >> 6] |= static_cast<uint64_t>(bits[k] & 1u) << (k & 63);
        }
        for (int b = 0; b < 256; ++b) {
            uint64_t w = 0;
            for (int j = 0; j < 8; ++j) w |= static_cast<uint64_t>(2u * ((b >> j) & 1u)) << (8 * j);
            spread[b] = w;
        }
        std::memset(sync_bytes, 0, sizeof sync_bytes);
        std::memcpy(sync_bytes, sync_vector(), kNSym);
    }
};
void symbols_of_packed(const unsigned char* data, unsigned char* symbols) {
    static const SymbolCoder coder;
    uint64_t w[3] = {0, 0, 0};
    for (int b = 0; b < 11; ++b) {
        unsigned v = data[b];
        while (v) {
            const int hi = 31 - __builtin_clz(v);                    // bit position within the byte, 7 = MSB = first sent
            const uint64_t* pt = coder.pattern[8 * b + (7 - hi)];
            w[0] ^= pt[0]; w[1] ^= pt[1]; w[2] ^= pt[2];
            v &= ~(1u << hi);
        }
    }
    unsigned char out[168];
    for (int q = 0; q < 21; ++q) {
        uint64_t word = coder.spread[(w[q >> 3] >> (8 * (q & 7))) & 0xffu], sv;
        std::memcpy(&sv, coder.sync_bytes + 8 * q, 8);
        word += sv;                                                   // bytes 0..3: no carry between them
        std::memcpy(out + 8 * q, &word, 8);
    }
    std::memcpy(symbols, out, kNSym);
}
}  // namespace

// wsprsim_utils.c:163-316: text -> 50 bits -> 162 convolutionally coded,
// interleaved bits -> 4-FSK symbol = 2*bit + sync.
int channel_symbols(const char* text, HashTable& tab, unsigned char* symbols) {
    char buf[24];
    std::memset(buf, 0, sizeof buf);
    std::strncpy(buf, text, 22);
    const size_t len = std::strlen(buf);
    const size_t sp = std::strcspn(buf, " "), sl = std::strcspn(buf, "/");
    const size_t lt = std::strcspn(buf, "<"), gt = std::strcspn(buf, ">");
    static const int round_to_legal[10] = {0, -1, 1, 0, -1, 2, 1, 0, -1, 1};
    auto legalise = [&](int p) {
        p = p < 0 ? 0 : (p > 60 ? 60 : p);
        return p + round_to_legal[p % 10];
    };
    unsigned long n = 0;
    int m = 0;
    Splitter tok(buf);

    if (sp > 3 && sp < 7 && sl == len && lt == len) {              // CALL GRID dBm
        const char* cs = tok.next(" ");
        const char* gr = tok.next(" ");
        const char* pw = tok.next(" ");
        if (!cs || !gr || !pw) return 0;
        n = pack_callsign(cs);
        char codes[4];
        for (int i = 0; i < 4; ++i) codes[i] = locator_code(gr[i]);
        m = static_cast<int>(pack_grid_power(codes, std::atoi(pw)));
    } else if (lt == 0 && gt < len) {                              // <CALL> GRID6 dBm
        const char* cs = tok.next("<> ");
        const char* gr = tok.next(" ");
        const char* pw = tok.next(" ");
        if (!cs || !gr || !pw) return 0;
        const int ntype = -(legalise(std::atoi(pw)) + 1);
        m = 128 * static_cast<int>(nhash15(cs, std::strlen(cs), 146u)) + ntype + 64;
        char rot[7];
        std::memset(rot, 0, sizeof rot);
        const int gl = static_cast<int>(std::strlen(gr));
        for (int i = 0; i < gl - 1 && i < 6; ++i) rot[i] = gr[i + 1];
        rot[5] = gr[0];
        n = pack_callsign(rot);
    } else if (sl < len) {                                         // PFX/CALL dBm, CALL/S dBm
        char* cs = tok.next(" ");
        if (!cs || sl == 0 || sl > std::strlen(cs)) return 0;
        const char* pw = tok.next(" ");
        if (!pw) return 0;
        const int power = legalise(std::atoi(pw));
        int32_t n1, ng, nadd;
        pack_compound(cs, &n1, &ng, &nadd);
        m = 128 * ng + (power + 1 + nadd) + 64;
        n = static_cast<unsigned long>(static_cast<long>(n1));
    } else {
        return 0;
    }

    unsigned char data[11] = {0};
    data[0] = static_cast<unsigned char>(n >> 20);
    data[1] = static_cast<unsigned char>(n >> 12);
    data[2] = static_cast<unsigned char>(n >> 4);
    data[3] = static_cast<unsigned char>(((n & 0x0f) << 4) + ((m >> 18) & 0x0f));
    data[4] = static_cast<unsigned char>(m >> 10);
    data[5] = static_cast<unsigned char>(m >> 2);
    data[6] = static_cast<unsigned char>((m & 0x03) << 6);

    {   // the reference re-unpacks its own packing (wsprsim_utils.c:280-300); the only
        // lasting effect is the hash-table entry, which type-3 decodes depend on
        signed char chk[11];
        std::memcpy(chk, data, sizeof chk);
        char a[23], b[13], c[13], d[7], e[3];
        QuietView quiet(tab);
        unpack_message(chk, quiet, a, b, d, e, c);
    }

    symbols_of_packed(data, symbols);
    return 1;
}

}  // namespace wspr

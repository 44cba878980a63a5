// Host-side message layer of the MI355X WSPR decoder: K=32 r=1/2 convolutional
// code + Fano sequential decoder, interleaver, 50-bit source coding (pack/unpack),
// callsign hash.  The north star keeps this integer/string work on the host CPU;
// the HIP pipeline hands it soft symbols and receives channel symbols back.
//
// Reference interfaces mirrored (same names are exported with C linkage from
// wspr_capi.cpp): wsprd/fano.h:14-28, wsprd/wsprd_utils.h:32-42,
// wsprd/wsprsim_utils.h:1-9, wsprd/nhash.h:3.
#pragma once
#include <cstddef>
#include <cstdint>

namespace wspr {

constexpr int kNSym = 162;
constexpr int kNBits = 81;
constexpr int kHashSlots = 32768;
constexpr int kHashWidth = 13;
constexpr int kLocWidth = 5;

// snprintf(dst, cap, "%s", src) without the format machinery: at most cap - 1 characters and the terminating NUL, nothing
// beyond it touched (the per-decode bookkeeping copies a dozen short texts; 70 ns each through snprintf)
inline void copy_text(char* dst, size_t cap, const char* src) {
    if (cap == 0) return;
    size_t i = 0;
    for (; i + 1 < cap && src[i] != '\0'; ++i) dst[i] = src[i];
    dst[i] = '\0';
}

// sync vector (wsprd/wsprd.c:84-93) as bytes 0/1
const unsigned char* sync_vector();

uint32_t nhash15(const void* key, size_t len, uint32_t seed);

char callsign_code(char ch);
char locator_code(char ch);
unsigned long pack_callsign(const char* call);
unsigned long pack_grid_power(const char* grid_codes, int power);
void pack_compound(char* call, int32_t* n, int32_t* m, int32_t* nadd);

void interleave162(unsigned char* sym);
void deinterleave162(unsigned char* sym);

int conv_encode(unsigned char* out, const unsigned char* data, unsigned nbytes);

// Branch-metric table [sent bit][received soft symbol], wsprd/wsprd.c:467-473
struct FanoMetrics {
    int tab[2][256];
    FanoMetrics();
};
const FanoMetrics& default_metrics();

int fano_decode(unsigned* metric, unsigned* cycles, unsigned* maxnp, unsigned char* data,
                const unsigned char* symbols, unsigned nbits, const int mettab[2][256],
                int delta, unsigned maxcycles);

void unpack_50bits(const signed char* dat, int32_t* n1, int32_t* n2);
int unpack_callsign(int32_t ncall, char* call);
int unpack_grid(int32_t ngrid, char* grid);
int unpack_prefix(int32_t nprefix, char* call);
// What unpack_message() / channel_symbols() need of the callsign hash memory.  The reference keeps it in two flat arrays
// local to wspr_decode (wsprd.c:478-479); a batch decoded with usehashtable shares one memory across segments in
// index order and sees it through a view that logs what each segment looked up and stored (wspr_pipeline.h, HashBatch).
struct HashTable {
    virtual ~HashTable() {}
    // callsign stored at `slot`, "" if none: the type-3 look-up of wsprd_utils.c:296-300
    virtual const char* call_at(int slot) = 0;
    // the same for a look-up whose answer is thrown away (the re-unpack inside channel_symbols, wsprsim_utils.c:280-300)
    virtual const char* peek(int slot) = 0;
    // type 1 (grid != nullptr): call and locator stored at `slot`; type 2 (grid == nullptr): the call alone
    virtual void put(int slot, const char* call, const char* grid) = 0;
};
// the reference's own form: hashtab[32768][13], loctab[32768][5]; dirty (optional) collects the slots written
struct FlatHashTable : HashTable {
    char* hashtab;
    char* loctab;
    void* dirty_vec;            // std::vector<int>* or nullptr (kept opaque: this header stays free of <vector>)
    FlatHashTable(char* h, char* l, void* dirty = nullptr) : hashtab(h), loctab(l), dirty_vec(dirty) {}
    const char* call_at(int slot) override { return hashtab + (size_t)slot * kHashWidth; }
    const char* peek(int slot) override { return hashtab + (size_t)slot * kHashWidth; }
    void put(int slot, const char* call, const char* grid) override;
};

int unpack_message(const signed char* msg, HashTable& tab, char* call_loc_pow,
                   char* call, char* loc, char* pwr, char* callsign);
int unpack_message(const signed char* msg, char* hashtab, char* loctab, char* call_loc_pow,
                   char* call, char* loc, char* pwr, char* callsign);
int channel_symbols(const char* text, HashTable& tab, unsigned char* symbols);
int channel_symbols(const char* text, char* hashtab, char* loctab, unsigned char* symbols);

}  // namespace wspr

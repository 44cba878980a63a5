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
int unpack_message(const signed char* msg, char* hashtab, char* loctab, char* call_loc_pow,
                   char* call, char* loc, char* pwr, char* callsign);
int channel_symbols(const char* text, char* hashtab, char* loctab, unsigned char* symbols);

}  // namespace wspr

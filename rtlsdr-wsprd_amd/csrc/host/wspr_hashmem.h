// The callsign hash memory of a batch decoded with usehashtable, and the per-thread cache of what decoded messages unpack
// and re-encode to.  Pure host C++ (no HIP): compiled with g++ like the message layer, unit-tested without a GPU
// (tests/test_hashmem.py through tests/helpers/hashmem_check.cpp).
#pragma once
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "wspr_message.h"

namespace wspr {

// ---- usehashtable on a batch -------------------------------------------------------------------------
// The reference's hash memory (hashtable.txt read before, written after every decode: wsprd.c:481-494, 842-852) orders
// the segments: what a type-3 "<call>" message resolves to (wsprd_utils.c:296-300) depends on what was heard before.
// A batch is nevertheless decoded IN PARALLEL: every segment sees the memory through a view (SegHashView) that
//   * answers a look-up from the segment's own earlier stores, else from the stores of EARLIER segments as currently
//     known (empty in the first round), else from the table loaded from the file -- and logs what it answered;
//   * logs the segment's stores in order.
// Afterwards the logs are checked in index order: a segment whose logged look-ups still get the same answers from its
// predecessors' (now known) stores is exactly what the serial walk would have produced -- its decode is a function of
// its samples and those answers alone; the others are decoded again against the updated memory, round by round, until
// none is left (segment k is final after round k at the latest; in practice after one or two).
struct HashOp {
    int32_t seg;        // global segment index
    int32_t slot;       // 0 .. 32767
    int32_t kind;       // 1 = type-1 store (call + locator), 2 = type-2 store (call only), 3 = look-up answered by the base
    char call[13];      // stored call, or the answer the look-up got ("" = none)
    char grid[5];
    char pad[2];
};
static_assert(sizeof(HashOp) == 32, "HashOp is exchanged between ranks as raw bytes");

struct HashBatch {
    int seg0 = 0;                                   // global index of this call's first segment
    std::vector<char> base_call, base_grid;         // the file as loaded: [32768][13], [32768][5]
    std::vector<HashOp> prior;                      // stores of segments outside this call (other shards), ascending seg
    std::vector<std::vector<HashOp>> log;           // per segment of this call: its stores and base look-ups, in order
    struct Ver { int32_t seg; char call[13]; };
    std::vector<std::vector<Ver>> ver;              // per slot: last store of each storing segment, ascending seg
    std::vector<int> touched;                       // slots with versions
    int rounds = 0, redecoded = 0;
    // what a later WSPR_HASH_REVISIT must find unchanged: the call completed (valid), over this many slots and samples
    bool valid = false;
    int nslots = 0, samples = 0;

    HashBatch();
    void load_file();                               // hashtable.txt of the working directory (wsprd.c:481-494)
    void resize(int nseg) { log.assign((size_t)nseg, {}); }
    void rebuild();                                 // ver := prior + log
    const char* lookup(int slot, int gseg) const;   // what segment gseg finds at slot from its predecessors / the file
    std::vector<int> invalid() const;               // local indices of segments with a look-up that would now differ
    std::vector<HashOp> stores() const;             // this call's stores in segment order
    static void commit_file(const std::vector<char>& call0, const std::vector<char>& grid0, const HashOp* w, size_t n);
    void commit_file() const;                       // file := base + prior + this call's stores (wsprd.c:842-852)
};

// One segment's window on the batch's hash memory (see HashBatch): own stores first, then the predecessors', then the file.
struct SegHashView : HashTable {
    HashBatch& hb;
    const int s;                                    // index within the call
    char tmp[13];                                   // an own store's text, copied: the log may grow under the caller
    SegHashView(HashBatch* hb_, int s_) : hb(*hb_), s(s_) {}
    const char* own(int slot);
    const char* peek(int slot) override;
    const char* call_at(int slot) override;
    void put(int slot, const char* call, const char* grid) override;
};

// What unpacking a decoded 50-bit message and re-encoding its text yield is a pure function of the bits as long as no
// hash look-up is involved (types 1 and 2; a type 3 asks the table): texts, the "noprint" flag, the stores into the
// hash memory (unpk_'s and those of the re-unpack inside get_wspr_channel_symbols) and the 162 channel symbols.  A
// receiver hears the same stations slot after slot, a batch holds thousands of copies of a few hundred messages, and
// this host work (a dozen snprintf, the convolutional encoder, the interleaver: ~1.2 us) is most of what a rank with
// few CPUs spends per decode.  Per host thread: the first occurrence is computed through a recording view of the
// segment's table, later ones replay the stores into THEIR segment's table and copy the rest.
class MessageCache {
public:
    struct Put { int slot; bool has_grid; char call[13]; char grid[5]; };
    struct Entry {
        int noprint = 0;
        char clp[23], call[13], loc[7], pwr[3], callsign[13];
        std::vector<Put> unpack_puts, chan_puts;
        int chan_state = 0;                          // 0 not asked yet, 1 symbols valid, 2 the text does not encode
        unsigned char sym[kNSym];
    };
    struct Handle { int noprint; Entry* entry; };    // entry == nullptr: not cacheable (the message looked the table up)
    static MessageCache& of_this_thread();
    // unpk_() of reference wsprd_utils.c:228-313 on the decoded bytes (decdata[0..10] as the Fano decoder left them)
    Handle unpack(const unsigned char* decdata, HashTable& tab, char* call_loc_pow, char* call, char* loc, char* pwr,
                  char* callsign);
    // get_wspr_channel_symbols(call_loc_pow, ...) of reference wsprsim_utils.c:163-316
    int symbols(Handle& h, const char* call_loc_pow, HashTable& tab, unsigned char* sym);
    size_t size() const { return map_.size(); }
    unsigned long lookups = 0, hits = 0;             // unpack() calls of this thread and how many the map answered

private:
    std::unordered_map<uint64_t, Entry> map_;
};

}  // namespace wspr

// CPU check of the batch hash memory (rtlsdr-wsprd_amd/csrc/host/wspr_hashmem.{h,cpp}: HashBatch, SegHashView,
// MessageCache) without a GPU: a model "decoder" replays, per segment, a list of decoded 50-bit messages through the
// product's own unpack_message() / channel_symbols() -- and, like the real decoder, behaves differently depending on what
// a type-3 look-up answers (here: an unresolved "<...>" makes the segment decode one more message, which stores).  The
// batch is "decoded" in rounds against HashBatch views (all segments at once, then only those whose logged look-ups no
// longer hold) and must equal ONE flat table walked through all segments in index order (the reference with -H:
// wsprd.c:481-494, 842-852) -- every text, every symbol vector, and the final hashtable.txt; once as one call, once as
// shards that exchange their stores (the protocol of wspr_decode_batch_hashed / dist.hashed_rounds).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../rtlsdr-wsprd_amd/csrc/host/wspr_message.cpp"
#include "../../rtlsdr-wsprd_amd/csrc/host/wspr_hashmem.cpp"

using namespace wspr;

namespace {
struct Msg { std::string text; unsigned char data[11]; };

bool bits_of(const std::string& text, unsigned char* data11) {
    std::vector<char> h((size_t)kHashSlots * kHashWidth, 0), l((size_t)kHashSlots * kLocWidth, 0);
    unsigned char sym[kNSym];
    if (!channel_symbols(text.c_str(), h.data(), l.data(), sym)) return false;
    unsigned char soft[kNSym];
    for (int i = 0; i < kNSym; ++i) soft[i] = (sym[i] >> 1) ? 255 : 0;
    deinterleave162(soft);
    unsigned metric, cycles, maxnp;
    memset(data11, 0, 11);
    return fano_decode(&metric, &cycles, &maxnp, data11, soft, kNBits, default_metrics().tab, 60, 10000) == 0;
}

struct Out {                                   // what the model decoder reports for one decoded message
    int noprint; std::string clp, callsign; int sym_ok; unsigned sym_sum;
    bool operator==(const Out& o) const { return noprint == o.noprint && clp == o.clp && callsign == o.callsign && sym_ok == o.sym_ok && sym_sum == o.sym_sum; }
};

// one segment through a table view; `extra` = the message decoded additionally after every unresolved "<...>"
std::vector<Out> decode_segment(const std::vector<int>& list, const std::vector<Msg>& pool, int extra, HashTable& tab, bool cached) {
    std::vector<Out> outs;
    std::vector<int> work(list.begin(), list.end());
    for (size_t i = 0; i < work.size(); ++i) {
        const Msg& m = pool[(size_t)work[i]];
        char clp[23] = {0}, call[13] = {0}, loc[7] = {0}, pwr[3] = {0}, cs[13] = {0};
        unsigned char sym[kNSym];
        Out o{};
        if (cached) {
            MessageCache& mc = MessageCache::of_this_thread();
            MessageCache::Handle h = mc.unpack(m.data, tab, clp, call, loc, pwr, cs);
            o.noprint = h.noprint;
            o.sym_ok = o.noprint ? -1 : mc.symbols(h, clp, tab, sym);
        } else {
            signed char msg[12] = {0};
            for (int k = 0; k < 11; ++k) msg[k] = (signed char)m.data[k];
            o.noprint = unpack_message(msg, tab, clp, call, loc, pwr, cs);
            o.sym_ok = o.noprint ? -1 : channel_symbols(clp, tab, sym);
        }
        o.clp = clp; o.callsign = cs;
        if (o.sym_ok == 1) for (int k = 0; k < kNSym; ++k) o.sym_sum = o.sym_sum * 31u + sym[k];
        outs.push_back(o);
        if (!strncmp(clp, "<...>", 5) && work.size() < list.size() + 4) work.push_back(extra);
    }
    return outs;
}

std::string dump(const std::vector<char>& call, const std::vector<char>& grid) {
    std::string s;
    char line[64];
    for (int i = 0; i < kHashSlots; ++i)
        if (call[(size_t)i * kHashWidth]) {
            snprintf(line, sizeof line, "%5d %s %s\n", i, call.data() + (size_t)i * kHashWidth, grid.data() + (size_t)i * kLocWidth);
            s += line;
        }
    return s;
}
std::string slurp(const char* path) {
    std::string s;
    if (FILE* f = fopen(path, "r")) { char b[4096]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) s.append(b, n); fclose(f); }
    return s;
}
}  // namespace

// returns 0 if everything agrees; the working directory receives hashtable.txt (run it in a scratch directory)
extern "C" int hashmem_selftest(unsigned seed, int nseg, int per_seg, double frac23, int nshards, int* rounds_out,
                                int* redecoded_out, int* resolved_out, char* err, int errcap) {
    std::mt19937 rng(seed);
    auto fail = [&](const std::string& what) { snprintf(err, (size_t)errcap, "%s", what.c_str()); return 1; };
    // ---- message pool: plain calls (many, so that hash slots collide), compound-call stations in both forms
    std::vector<Msg> pool;
    const char* L = "ABCDEFGHIJKLMNOPQRSTUVWXYZ";
    const int powers[] = {0, 3, 7, 10, 13, 17, 20, 23, 27, 30, 33, 37, 40};
    for (int i = 0; i < 3000; ++i) {
        char t[32];
        snprintf(t, sizeof t, "%c%c%d%c%c%c %c%c%d%d %d", L[rng() % 26], L[rng() % 26], (int)(rng() % 10), L[rng() % 26], L[rng() % 26],
                 L[rng() % 26], L[rng() % 18], L[rng() % 18], (int)(rng() % 10), (int)(rng() % 10), powers[rng() % 13]);
        Msg m; m.text = t;
        if (bits_of(m.text, m.data)) pool.push_back(m);
    }
    const int nplain = (int)pool.size();
    if (nplain < 2000) return fail("message pool too small");
    struct Station { int type2, type3; };
    std::vector<Station> stations;
    for (int i = 0; i < 40; ++i) {
        char c2[32], c3[32];
        char call[16];
        snprintf(call, sizeof call, "%c%c%d/%c%d%c%c", L[rng() % 26], L[rng() % 26], (int)(rng() % 10), L[rng() % 26], (int)(rng() % 10), L[rng() % 26], L[rng() % 26]);
        const int pw = powers[rng() % 13];
        snprintf(c2, sizeof c2, "%s %d", call, pw);
        snprintf(c3, sizeof c3, "<%s> %c%c%d%d%c%c %d", call, L[rng() % 18], L[rng() % 18], (int)(rng() % 10), (int)(rng() % 10), L[rng() % 24], L[rng() % 24], pw);
        Msg a, b; a.text = c2; b.text = c3;
        if (!bits_of(a.text, a.data) || !bits_of(b.text, b.data)) continue;
        stations.push_back({(int)pool.size(), (int)pool.size() + 1});
        pool.push_back(a); pool.push_back(b);
    }
    if (stations.size() < 20) return fail("too few compound-call stations encode");
    const int extra = 7;                                   // a plain message: decoded after every unresolved "<...>"
    // ---- the job
    std::vector<std::vector<int>> job((size_t)nseg);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (int s = 0; s < nseg; ++s)
        for (int k = 0; k < per_seg; ++k) {
            if (U(rng) < frac23) {
                const int st = (int)(rng() % stations.size());
                job[(size_t)s].push_back(((s + st) & 1) ? stations[(size_t)st].type3 : stations[(size_t)st].type2);
            } else {
                job[(size_t)s].push_back((int)(rng() % (unsigned)nplain));
            }
        }
    // ---- the serial walk: ONE flat table through all segments in order
    std::vector<char> fh((size_t)kHashSlots * kHashWidth, 0), fl((size_t)kHashSlots * kLocWidth, 0);
    std::vector<std::vector<Out>> want((size_t)nseg);
    int resolved = 0;
    for (int s = 0; s < nseg; ++s) {
        FlatHashTable flat(fh.data(), fl.data());
        want[(size_t)s] = decode_segment(job[(size_t)s], pool, extra, flat, false);
        for (const Out& o : want[(size_t)s]) resolved += o.clp[0] == '<' && o.clp.compare(0, 5, "<...>") != 0;
    }
    const std::string want_file = dump(fh, fl);
    if (resolved_out) *resolved_out = resolved;
    // ---- the batch, in shards (1 = one call), to the fixed point
    remove("hashtable.txt");
    if (nshards < 1) nshards = 1;
    struct Shard { int lo, hi; HashBatch hb; std::vector<HashOp> stores; std::string seen; bool started = false; };
    std::vector<Shard> sh((size_t)nshards);
    std::vector<std::vector<Out>> got((size_t)nseg);
    for (int r = 0; r < nshards; ++r) {
        const int base = nseg / nshards, rem = nseg % nshards;
        sh[(size_t)r].lo = r * base + std::min(r, rem);
        sh[(size_t)r].hi = sh[(size_t)r].lo + base + (r < rem ? 1 : 0);
    }
    int rounds = 0, redecoded = 0;
    auto run_shard = [&](Shard& S, const std::vector<HashOp>& prior) {
        HashBatch& hb = S.hb;
        auto decode = [&](int local) {
            hb.log[(size_t)local].clear();
            SegHashView v(&hb, local);
            got[(size_t)(S.lo + local)] = decode_segment(job[(size_t)(S.lo + local)], pool, extra, v, true);
        };
        hb.prior = prior;
        if (!S.started) {
            hb.load_file();                                // empty: no file yet
            hb.seg0 = S.lo;
            hb.resize(S.hi - S.lo);
            for (int i = 0; i < S.hi - S.lo; ++i) decode(i);
            S.started = true;
        }
        for (;;) {
            hb.rebuild();
            const std::vector<int> todo = hb.invalid();
            if (todo.empty()) break;
            ++rounds; redecoded += (int)todo.size();
            for (int i : todo) decode(i);
            if (rounds > nseg + 4) return false;
        }
        S.stores = hb.stores();
        return true;
    };
    for (int pass = 0;; ++pass) {
        bool changed = false;
        for (int r = 0; r < nshards; ++r) {
            std::vector<HashOp> prior;
            for (int q = 0; q < r; ++q) prior.insert(prior.end(), sh[(size_t)q].stores.begin(), sh[(size_t)q].stores.end());
            const std::string sig(reinterpret_cast<const char*>(prior.data()), prior.size() * sizeof(HashOp));
            if (sh[(size_t)r].started && sig == sh[(size_t)r].seen) continue;
            const std::vector<HashOp> before = sh[(size_t)r].stores;
            if (!run_shard(sh[(size_t)r], prior)) return fail("no fixed point");
            sh[(size_t)r].seen = sig;
            if (before.size() != sh[(size_t)r].stores.size() ||
                memcmp(before.data(), sh[(size_t)r].stores.data(), before.size() * sizeof(HashOp)) != 0) changed = true;
        }
        if (!changed) break;
        if (pass > nshards + 2) return fail("the shards' exchange does not settle");
    }
    std::vector<HashOp> all;
    for (auto& S : sh) all.insert(all.end(), S.stores.begin(), S.stores.end());
    HashBatch empty;
    HashBatch::commit_file(empty.base_call, empty.base_grid, all.data(), all.size());
    if (rounds_out) *rounds_out = rounds;
    if (redecoded_out) *redecoded_out = redecoded;
    for (int s = 0; s < nseg; ++s)
        if (!(got[(size_t)s] == want[(size_t)s])) {
            char b[160];
            snprintf(b, sizeof b, "segment %d differs from the serial walk (%zu vs %zu decodes)", s, got[(size_t)s].size(), want[(size_t)s].size());
            return fail(b);
        }
    if (slurp("hashtable.txt") != want_file) return fail("hashtable.txt differs from the serial walk's table");
    return 0;
}

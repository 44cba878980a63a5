// Batch hash memory (usehashtable on a batch) and the per-thread message cache: see wspr_hashmem.h.
#include "wspr_hashmem.h"

#include <algorithm>
#include <cstdio>

namespace wspr {

HashBatch::HashBatch()
    : base_call((size_t)kHashSlots * kHashWidth, 0), base_grid((size_t)kHashSlots * kLocWidth, 0), ver((size_t)kHashSlots) {}

void HashBatch::load_file() {                                     // wsprd.c:481-494
    std::fill(base_call.begin(), base_call.end(), 0);
    std::fill(base_grid.begin(), base_grid.end(), 0);
    if (FILE* fh = fopen("hashtable.txt", "r+")) {
        char line[80], hcall[13], hgrid[5];
        int nh;
        while (fgets(line, sizeof line, fh) != nullptr) {
            hgrid[0] = hcall[0] = '\0';
            if (sscanf(line, "%d %12s %4s", &nh, hcall, hgrid) < 2) continue;
            if (nh >= 0 && nh < kHashSlots) {
                snprintf(base_call.data() + (size_t)nh * kHashWidth, kHashWidth, "%s", hcall);
                if (strlen(hgrid) > 0) snprintf(base_grid.data() + (size_t)nh * kLocWidth, kLocWidth, "%s", hgrid);
            }
        }
        fclose(fh);
    }
}

void HashBatch::rebuild() {
    for (int slot : touched) ver[(size_t)slot].clear();
    touched.clear();
    auto add = [&](const HashOp& op) {
        if (op.kind != 1 && op.kind != 2) return;
        if (op.slot < 0 || op.slot >= kHashSlots) return;
        auto& v = ver[(size_t)op.slot];
        if (v.empty()) touched.push_back(op.slot);
        if (v.empty() || v.back().seg != op.seg) v.emplace_back();
        v.back().seg = op.seg;
        memcpy(v.back().call, op.call, sizeof op.call);
    };
    // ascending segment order: the other shards' stores that precede this call, this call's, the ones that follow
    size_t p = 0;
    for (; p < prior.size() && prior[p].seg < seg0; ++p) add(prior[p]);
    for (const auto& l : log) for (const HashOp& op : l) add(op);
    for (; p < prior.size(); ++p) add(prior[p]);
}

const char* HashBatch::lookup(int slot, int gseg) const {
    const auto& v = ver[(size_t)slot];
    for (size_t i = v.size(); i-- > 0;)
        if (v[i].seg < gseg) return v[i].call;
    return base_call.data() + (size_t)slot * kHashWidth;
}

std::vector<int> HashBatch::invalid() const {
    std::vector<int> out;
    for (size_t s = 0; s < log.size(); ++s)
        for (const HashOp& op : log[s])
            if (op.kind == 3 && strcmp(lookup(op.slot, seg0 + (int)s), op.call) != 0) { out.push_back((int)s); break; }
    return out;
}

std::vector<HashOp> HashBatch::stores() const {
    std::vector<HashOp> out;
    for (const auto& l : log) for (const HashOp& op : l) if (op.kind == 1 || op.kind == 2) out.push_back(op);
    return out;
}

void HashBatch::commit_file(const std::vector<char>& call0, const std::vector<char>& grid0, const HashOp* w, size_t n) {
    std::vector<char> call = call0, grid = grid0;
    for (size_t i = 0; i < n; ++i) {
        const HashOp& op = w[i];
        if ((op.kind != 1 && op.kind != 2) || op.slot < 0 || op.slot >= kHashSlots) continue;
        snprintf(call.data() + (size_t)op.slot * kHashWidth, kHashWidth, "%s", op.call);
        if (op.kind == 1) snprintf(grid.data() + (size_t)op.slot * kLocWidth, kLocWidth, "%s", op.grid);
    }
    if (FILE* fh = fopen("hashtable.txt", "w")) {                 // wsprd.c:842-852
        for (int i = 0; i < kHashSlots; ++i)
            if (call[(size_t)i * kHashWidth] != '\0')
                fprintf(fh, "%5d %s %s\n", i, call.data() + (size_t)i * kHashWidth, grid.data() + (size_t)i * kLocWidth);
        fclose(fh);
    }
}

void HashBatch::commit_file() const {
    std::vector<HashOp> all;
    size_t p = 0;
    for (; p < prior.size() && prior[p].seg < seg0; ++p) all.push_back(prior[p]);
    for (const HashOp& op : stores()) all.push_back(op);
    for (; p < prior.size(); ++p) all.push_back(prior[p]);
    commit_file(base_call, base_grid, all.data(), all.size());
}

const char* SegHashView::own(int slot) {
    const auto& l = hb.log[(size_t)s];
    for (size_t i = l.size(); i-- > 0;)
        if (l[i].slot == slot && (l[i].kind == 1 || l[i].kind == 2)) { memcpy(tmp, l[i].call, sizeof tmp); return tmp; }
    return nullptr;
}
const char* SegHashView::peek(int slot) {
    if (const char* c = own(slot)) return c;
    return hb.lookup(slot, hb.seg0 + s);
}
const char* SegHashView::call_at(int slot) {
    if (const char* c = own(slot)) return c;
    const char* c = hb.lookup(slot, hb.seg0 + s);
    HashOp op{};
    op.seg = hb.seg0 + s; op.slot = slot; op.kind = 3;
    copy_text(op.call, sizeof op.call, c);
    hb.log[(size_t)s].push_back(op);
    return c;                                   // base / version storage: unchanged for the whole round
}
void SegHashView::put(int slot, const char* call, const char* grid) {
    HashOp op{};
    op.seg = hb.seg0 + s; op.slot = slot; op.kind = grid ? 1 : 2;
    copy_text(op.call, sizeof op.call, call);
    if (grid) copy_text(op.grid, sizeof op.grid, grid);
    hb.log[(size_t)s].push_back(op);
}

namespace {
struct RecordingTable : HashTable {
    HashTable& t;
    std::vector<MessageCache::Put>& puts;
    bool looked_up = false;
    RecordingTable(HashTable& t_, std::vector<MessageCache::Put>& p) : t(t_), puts(p) {}
    const char* call_at(int slot) override { looked_up = true; return t.call_at(slot); }
    const char* peek(int slot) override { return t.peek(slot); }
    void put(int slot, const char* call, const char* grid) override {
        MessageCache::Put p{};
        p.slot = slot; p.has_grid = grid != nullptr;
        copy_text(p.call, sizeof p.call, call);
        if (grid) copy_text(p.grid, sizeof p.grid, grid);
        puts.push_back(p);
        t.put(slot, call, grid);
    }
};
}  // namespace

MessageCache& MessageCache::of_this_thread() {
    static thread_local MessageCache c;
    return c;
}

MessageCache::Handle MessageCache::unpack(const unsigned char* decdata, HashTable& tab, char* call_loc_pow, char* call,
                                          char* loc, char* pwr, char* callsign) {
    uint64_t key = 0;
    for (int k = 0; k < 7; ++k) key = (key << 8) | decdata[k];   // the 50 message bits live in bytes 0..6 (unpack50)
    if (map_.size() > 20000) map_.clear();                       // a few MB per host thread at most
    ++lookups;
    auto it = map_.find(key);
    if (it != map_.end()) {
        ++hits;
        Entry& e = it->second;
        memcpy(call_loc_pow, e.clp, sizeof e.clp); memcpy(call, e.call, sizeof e.call); memcpy(loc, e.loc, sizeof e.loc);
        memcpy(pwr, e.pwr, sizeof e.pwr); memcpy(callsign, e.callsign, sizeof e.callsign);
        for (const Put& p : e.unpack_puts) tab.put(p.slot, p.call, p.has_grid ? p.grid : nullptr);
        return Handle{e.noprint, &e};
    }
    signed char message[12] = {0};
    for (int k = 0; k < 11; ++k) message[k] = (signed char)decdata[k];
    Entry e;
    RecordingTable rec(tab, e.unpack_puts);
    const int noprint = unpack_message(message, rec, call_loc_pow, call, loc, pwr, callsign);
    if (rec.looked_up) return Handle{noprint, nullptr};
    e.noprint = noprint;
    memcpy(e.clp, call_loc_pow, sizeof e.clp); memcpy(e.call, call, sizeof e.call); memcpy(e.loc, loc, sizeof e.loc);
    memcpy(e.pwr, pwr, sizeof e.pwr); memcpy(e.callsign, callsign, sizeof e.callsign);
    Entry& stored = map_.emplace(key, std::move(e)).first->second;     // references to elements survive rehashing
    return Handle{noprint, &stored};
}

int MessageCache::symbols(Handle& h, const char* call_loc_pow, HashTable& tab, unsigned char* sym) {
    if (!h.entry) return channel_symbols(call_loc_pow, tab, sym);
    Entry& e = *h.entry;
    if (e.chan_state == 0) {
        RecordingTable rec(tab, e.chan_puts);
        e.chan_state = channel_symbols(call_loc_pow, rec, e.sym) ? 1 : 2;
    } else {
        for (const Put& p : e.chan_puts) tab.put(p.slot, p.call, p.has_grid ? p.grid : nullptr);
    }
    if (e.chan_state == 1) memcpy(sym, e.sym, kNSym);
    return e.chan_state == 1;
}

}  // namespace wspr

// Sanitizer driver (SURVEY section 5 "race detection / sanitizers"; the reference's Makefile:5-7 suggests ASan): a plain C++
// program -- no Python interpreter in the process -- linked against a build of the LAB library whose HOST code is
// instrumented (tools/sanitize.sh: AddressSanitizer + UndefinedBehaviorSanitizer, or ThreadSanitizer).  It walks the
// threaded host paths: slots of one call, lanes of several calls, the pinned chunk ring of the host-buffer entry, the
// batch hash memory's rounds, a receiver session whose feed / roll-over / decode run on three threads, three receivers fed by
// three RX threads and decoded together, three threads sharing lane 0, the node-level
// call folded onto lanes, buffer release.  "host" = the part that needs no GPU (message layer, file formats, hash file).
// Every check is a plain comparison; the sanitizers report on their own.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

#include "../include/wspr_mi355x_bench.h"

static int g_fail = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { fprintf(stderr, "CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } \
    } while (0)

static const int NS = 45000;

// one synthetic segment: the reference's self-test signal model (rtlsdr_wsprd.c:743-760) in noise, normalised to 0.5
static void make_segment(const char* msg, double f0, double t0, double snr_db, unsigned seed, float* I, float* Q) {
    static thread_local std::vector<char> hashtab(HASHTAB_SIZE * HASHTAB_ENTRY_LEN), loctab(HASHTAB_SIZE * LOCTAB_ENTRY_LEN);
    unsigned char sym[162];
    char text[32];
    snprintf(text, sizeof text, "%s", msg);
    std::fill(hashtab.begin(), hashtab.end(), 0);
    std::fill(loctab.begin(), loctab.end(), 0);
    CHECK(get_wspr_channel_symbols(text, hashtab.data(), loctab.data(), sym) == 1);
    std::mt19937 rng(seed);
    std::normal_distribution<double> nz(0.0, std::sqrt((375.0 / 2500.0) / 2.0));
    std::vector<double> di(NS), dq(NS);
    for (int i = 0; i < NS; ++i) { di[i] = nz(rng); dq[i] = nz(rng); }
    const double amp = std::pow(10.0, snr_db / 20.0), dt = 1.0 / 375.0, df = 375.0 / 256.0;
    double phi = 0.0;
    const int start = (int)std::lround(t0 / dt);
    for (int s = 0; s < 162; ++s) {
        const double dphi = 2.0 * M_PI * dt * (f0 + (sym[s] - 1.5) * df);
        for (int k = 0; k < 256; ++k) {
            const int i = start + s * 256 + k;
            if (i >= 0 && i < NS) { di[i] += amp * std::cos(phi); dq[i] += amp * std::sin(phi); }
            phi += dphi;
        }
    }
    double peak = 1e-24;
    for (int i = 0; i < NS; ++i) { peak = std::max(peak, std::fabs(di[i])); peak = std::max(peak, std::fabs(dq[i])); }
    const float scale = (float)(0.5 / peak);
    for (int i = 0; i < NS; ++i) { I[i] = (float)di[i] * scale; Q[i] = (float)dq[i] * scale; }
}

static decoder_options options(int usehash = 0) {
    decoder_options o;
    memset(&o, 0, sizeof o);
    o.freq = 144489000; o.npasses = 2; o.subtraction = 1; o.usehashtable = usehash;
    return o;
}

static void host_part() {
    // message layer round trips (the reference's own harness covers values; here: memory and UB under instrumentation)
    std::vector<char> hashtab(HASHTAB_SIZE * HASHTAB_ENTRY_LEN), loctab(HASHTAB_SIZE * LOCTAB_ENTRY_LEN);
    const char* msgs[] = {"K1JT FN20 20", "PJ4/K1ABC 37", "<PJ4/K1ABC> FK52UD 37", "K1ABC/7 30", "W1AW FN31 10", "VA2GKA FN35 23"};
    int mettab[2][256];
    wspr_fano_metric_table(mettab);
    for (const char* m : msgs) {
        char text[32];
        snprintf(text, sizeof text, "%s", m);
        unsigned char sym[162];
        CHECK(get_wspr_channel_symbols(text, hashtab.data(), loctab.data(), sym) == 1);
        unsigned char soft[162];
        for (int i = 0; i < 162; ++i) soft[i] = (sym[i] >> 1) ? 200 : 56;
        deinterleave(soft);
        unsigned metric, cycles, maxnp;
        unsigned char data[11] = {0};
        CHECK(fano(&metric, &cycles, &maxnp, data, soft, 81, mettab, 60, 10000) == 0);
        signed char msg11[12] = {0};
        for (int i = 0; i < 11; ++i) msg11[i] = (signed char)data[i];
        char clp[23] = {0}, call[13] = {0}, loc[7] = {0}, pwr[3] = {0}, cs[13] = {0};
        unpk_(msg11, hashtab.data(), loctab.data(), clp, call, loc, pwr, cs);
        CHECK(strcmp(clp, m) == 0);
    }
    // file formats
    std::vector<float> I(NS), Q(NS), I2(NS), Q2(NS);
    make_segment("K1JT FN20 20", 10.0, 2.0, -5.0, 7, I.data(), Q.data());
    char path[64];
    snprintf(path, sizeof path, "/tmp/wspr_san_%d.iq", (int)getpid());
    CHECK(wspr_write_iq_file(path, I.data(), Q.data()) == NS);
    CHECK(wspr_read_iq_file(path, I2.data(), Q2.data()) == NS);
    unlink(path);
    decoder_results r;
    memset(&r, 0, sizeof r);
    r.freq = 144.490550; snprintf(r.call, sizeof r.call, "K1JT"); snprintf(r.loc, sizeof r.loc, "FN20"); snprintf(r.pwr, sizeof r.pwr, "20");
    char line[160];
    decoder_options o = options();
    snprintf(o.rcall, sizeof o.rcall, "VA2GKA"); snprintf(o.rloc, sizeof o.rloc, "FN35");
    CHECK(wspr_format_spot(&r, line, sizeof line) > 0);
    CHECK(wspr_format_spot_timestamped(&r, 2026, 9, 29, 1, 2, line, sizeof line) > 0);
    char url[512];
    CHECK(wspr_format_wsprnet_url(&r, &o, 144489000.0, 2026, 9, 29, 1, 2, "0.5.6", url, sizeof url) > 0);
    CHECK(wspr_format_wsprnet_url(nullptr, &o, 144489000.0, 2026, 9, 29, 1, 2, "0.5.6", url, sizeof url) > 0);
    // hash file
    wspr_hash_op ops[2];
    memset(ops, 0, sizeof ops);
    ops[0].seg = 0; ops[0].slot = 17; ops[0].kind = 1; snprintf(ops[0].call, 13, "K1JT"); snprintf(ops[0].grid, 5, "FN20");
    ops[1].seg = 1; ops[1].slot = 99; ops[1].kind = 2; snprintf(ops[1].call, 13, "PJ4/K1ABC");
    CHECK(wspr_hash_commit(ops, 2) == 0);
    int lo, hi;
    wspr_shard_range(65536, 3, 8, &lo, &hi);
    CHECK(lo == 24576 && hi == 32768);
}

static int decode_batch(std::vector<float>& I, std::vector<float>& Q, int nseg, decoder_options o, std::vector<int>& nres,
                        std::vector<decoder_results>& out) {
    out.assign((size_t)nseg * 8, decoder_results());
    nres.assign(nseg, 0);
    return wspr_decode_batch(I.data(), Q.data(), nseg, NS, NS, o, out.data(), 8, nres.data(), 0);
}

static void gpu_part(int nseg) {
    CHECK(wspr_device_ready() == 1);
    if (g_fail) return;
    std::vector<float> I((size_t)nseg * NS), Q((size_t)nseg * NS);
    const char* calls[] = {"K1JT FN20 20", "W1AW FN31 10", "VA2GKA FN35 23", "G4ABC IO91 27"};
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 8; ++t)
            th.emplace_back([&, t] {
                for (int s = t; s < nseg; s += 8)
                    make_segment(calls[s % 4], -90.0 + 180.0 * (s % 17) / 16.0, 2.0 + 0.01 * (s % 50), -14.0, 100 + s,
                                 I.data() + (size_t)s * NS, Q.data() + (size_t)s * NS);
            });
        for (auto& t : th) t.join();
    }
    // 1. one call: its slots (three threads inside the library), pageable rows through the pinned chunk ring
    std::vector<int> n0;
    std::vector<decoder_results> r0;
    CHECK(decode_batch(I, Q, nseg, options(), n0, r0) == 0);
    int decoded = 0;
    for (int s = 0; s < nseg; ++s) decoded += (n0[s] >= 1 && strncmp(r0[(size_t)s * 8].message, calls[s % 4], 22) == 0);
    printf("one call, %d segments: %d decode their message\n", nseg, decoded);
    CHECK(decoded >= nseg * 9 / 10);
    // 2. three lanes at once, one of them on pinned rows; results must be the first call's
    CHECK(wspr_pin_host_buffer(I.data(), I.size() * 4) == 0 && wspr_pin_host_buffer(Q.data(), Q.size() * 4) == 0);
    {
        std::vector<std::vector<int>> n(3);
        std::vector<std::vector<decoder_results>> r(3);
        std::vector<std::thread> th;
        for (int k = 0; k < 3; ++k)
            th.emplace_back([&, k] {
                CHECK(wspr_bind_thread_lane(1 + k) == 1 + k);
                wspr_set_thread_slots(k == 2 ? 1 : 0);
                for (int rep = 0; rep < 2; ++rep) CHECK(decode_batch(I, Q, nseg, options(), n[k], r[k]) == 0);
            });
        for (auto& t : th) t.join();
        for (int k = 0; k < 3; ++k) {
            CHECK(n[k] == n0);
            CHECK(memcmp(r[k].data(), r0.data(), r0.size() * sizeof(decoder_results)) == 0);
        }
    }
    CHECK(wspr_unpin_host_buffer(I.data()) == 0 && wspr_unpin_host_buffer(Q.data()) == 0);
    // 3. usehashtable on a batch: the shared hash memory's rounds (segments alternate type 2 / type 3 of one station)
    {
        const int nh = 48;
        std::vector<float> HI((size_t)nh * NS), HQ((size_t)nh * NS);
        for (int s = 0; s < nh; ++s)
            make_segment(s % 2 == 0 ? "PJ4/K1ABC 37" : "<PJ4/K1ABC> FK52UD 37", 20.0, 2.0, -8.0, 900 + s, HI.data() + (size_t)s * NS,
                         HQ.data() + (size_t)s * NS);
        unlink("hashtable.txt");
        std::vector<int> nh_res;
        std::vector<decoder_results> rh;
        CHECK(decode_batch(HI, HQ, nh, options(1), nh_res, rh) == 0);
        int resolved = 0;
        for (int s = 1; s < nh; s += 2) resolved += nh_res[s] >= 1 && strncmp(rh[(size_t)s * 8].message, "<PJ4/K1ABC> FK52UD 37", 22) == 0;
        printf("usehashtable batch: %d of %d hashed calls resolved through earlier segments\n", resolved, nh / 2);
        CHECK(resolved >= nh / 2 - 2);
        unlink("hashtable.txt");
    }
    // 4. a receiver session: feed (RX thread), roll-over (main loop) and decode (decoder thread) at once
    {
        wspr_session* ss = wspr_session_create(options());
        CHECK(ss != nullptr);
        std::atomic<bool> stop{false};
        std::vector<uint8_t> raw(65536);
        std::mt19937 rng(5);
        for (auto& b : raw) b = (uint8_t)(120 + rng() % 16);
        std::thread rx([&] { while (!stop.load()) CHECK(wspr_session_feed(ss, raw.data(), (uint32_t)raw.size()) >= 0); });
        std::thread dec([&] {
            decoder_results d[50];
            int n = 0;
            for (int k = 0; k < 6; ++k) { usleep(20000); CHECK(wspr_session_decode(ss, k & 1, d, &n) >= 0); }
        });
        for (int k = 0; k < 12; ++k) { usleep(10000); CHECK(wspr_session_rollover(ss) >= 0); }
        dec.join();
        stop.store(true);
        rx.join();
        wspr_session_destroy(ss);
    }
    // 4b. three receivers, one RX thread each (their front ends share the library's reserved lane: turns), and their
    //     completed buffers decoded together
    {
        wspr_session* rx[3];
        std::vector<std::thread> th;
        for (int k = 0; k < 3; ++k) { rx[k] = wspr_session_create(options()); CHECK(rx[k] != nullptr); }
        for (int k = 0; k < 3; ++k)
            th.emplace_back([&, k] {
                std::vector<uint8_t> raw(65536);
                std::mt19937 rng(50 + k);
                for (auto& b : raw) b = (uint8_t)(120 + rng() % 16);
                for (int c = 0; c < 150; ++c) CHECK(wspr_session_feed(rx[k], raw.data(), (uint32_t)raw.size()) >= 0);
            });
        for (auto& t : th) t.join();
        {                                           // ... and ten callbacks of all three at once
            std::vector<uint8_t> raw(3 * 65536);
            std::mt19937 rng(77);
            for (auto& b : raw) b = (uint8_t)(120 + rng() % 16);
            const uint8_t* ptrs[3] = {raw.data(), raw.data() + 65536, raw.data() + 2 * 65536};
            int fills[3];
            for (int c = 0; c < 10; ++c) CHECK(wspr_session_feed_many(rx, ptrs, 65536, 3, fills) == 0);
            CHECK(fills[0] == fills[1] && fills[1] == fills[2]);
        }
        int bufs[3], nres[3], flags[3];
        for (int k = 0; k < 3; ++k) bufs[k] = wspr_session_rollover(rx[k]);
        std::vector<decoder_results> d(3 * 8);
        CHECK(wspr_session_decode_many(rx, bufs, 3, d.data(), 8, nres, flags) == 0);      // 150 callbacks: too short, all three
        CHECK(wspr_session_fill(rx[0], bufs[0]) == wspr_session_fill(rx[1], bufs[1]) && wspr_session_fill(rx[0], bufs[0]) > 700);
        for (int k = 0; k < 3; ++k) wspr_session_destroy(rx[k]);
    }
    // 4c. threads that never bound a lane: all on lane 0, their calls take turns; results must be the first call's
    {
        const int part = std::min(nseg, 64);
        std::vector<std::vector<int>> n(3, std::vector<int>(part));
        std::vector<std::vector<decoder_results>> r(3, std::vector<decoder_results>((size_t)part * 8));
        std::vector<std::thread> th;
        for (int k = 0; k < 3; ++k)
            th.emplace_back([&, k] {
                for (int rep = 0; rep < 2; ++rep)
                    CHECK(wspr_decode_batch(I.data(), Q.data(), part, NS, NS, options(), r[k].data(), 8, n[k].data(), 0) == 0);
            });
        for (auto& t : th) t.join();
        for (int k = 0; k < 3; ++k) {
            CHECK(std::equal(n[k].begin(), n[k].end(), n0.begin()));
            CHECK(memcmp(r[k].data(), r0.data(), (size_t)part * 8 * sizeof(decoder_results)) == 0);
        }
    }
    // 5. the node-level call folded onto lanes of this one device (lab hook)
    {
        setenv("WSPR_NODE_VIRTUAL", "1", 1);
        std::vector<int> nn(nseg);
        std::vector<decoder_results> rn((size_t)nseg * 8);
        CHECK(wspr_bind_thread_lane(0) == 0);
        CHECK(wspr_decode_batch_node(I.data(), Q.data(), nseg, NS, NS, options(), rn.data(), 8, nn.data(), 3) == 0);
        CHECK(nn == n0 && memcmp(rn.data(), r0.data(), r0.size() * sizeof(decoder_results)) == 0);
    }
    // 6. (round 6) a -H shard whose store buffer is too small: -3, nothing committed; completed by a revisit; a revisit whose
    //    rows are gone (buffers released) must be refused, not write through a dangling pointer
    {
        const int nh = 32;
        std::vector<float> HI((size_t)nh * NS), HQ((size_t)nh * NS);
        for (int s = 0; s < nh; ++s)
            make_segment(s % 2 == 1 ? "PJ4/K1ABC 37" : "<PJ4/K1ABC> FK52UD 37", 20.0, 2.0, -8.0, 1900 + s, HI.data() + (size_t)s * NS,
                         HQ.data() + (size_t)s * NS);
        unlink("hashtable.txt");
        std::vector<decoder_results> rh((size_t)nh * 8);
        std::vector<int> nr(nh);
        std::vector<wspr_hash_op> st(4 * nh + 64);
        int n_st = 0, n_re = 0;
        CHECK(wspr_decode_batch_hashed(HI.data(), HQ.data(), nh, NS, NS, options(1), rh.data(), 8, nr.data(), 0, 10, nullptr, 0,
                                       WSPR_HASH_KEEP_FILE, st.data(), 1, &n_st, &n_re) == -3);
        CHECK(n_st > 1 && access("hashtable.txt", F_OK) != 0);
        CHECK(wspr_decode_batch_hashed(HI.data(), HQ.data(), nh, NS, NS, options(1), rh.data(), 8, nr.data(), 0, 10, nullptr, 0,
                                       WSPR_HASH_KEEP_FILE | WSPR_HASH_REVISIT, st.data(), (int)st.size(), &n_st, &n_re) == 0);
        wspr_hash_op fake;
        memset(&fake, 0, sizeof fake);
        fake.seg = 5; fake.slot = (int)(nhash("PJ4/K1ABC", 9, 146)); fake.kind = 2; snprintf(fake.call, sizeof fake.call, "PJ4/K1ABC");   // segment 10 (a hashed call, unresolved so far) now resolves: it must be decoded again
        wspr_release_buffers();
        CHECK(wspr_decode_batch_hashed(HI.data(), HQ.data(), nh, NS, NS, options(1), rh.data(), 8, nr.data(), 0, 10, &fake, 1,
                                       WSPR_HASH_KEEP_FILE | WSPR_HASH_REVISIT, st.data(), (int)st.size(), &n_st, &n_re) < 0);
        unlink("hashtable.txt");
    }
    // 7. (round 6) the node-level call on resident rows: peer copies made to fail (the staged copy takes over), one shard
    //    made to fail (the whole call fails and reports nothing)
    {
        float *dI = nullptr, *dQ = nullptr;
        const size_t bytes = (size_t)nseg * NS * 4;
        CHECK(hipMalloc((void**)&dI, bytes) == hipSuccess && hipMalloc((void**)&dQ, bytes) == hipSuccess);
        CHECK(hipMemcpy(dI, I.data(), bytes, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dQ, Q.data(), bytes, hipMemcpyHostToDevice) == hipSuccess);
        std::vector<int> nn(nseg);
        std::vector<decoder_results> rn((size_t)nseg * 8);
        setenv("WSPR_NODE_FAIL_PEER", "1", 1);
        CHECK(wspr_decode_batch_node_device(dI, dQ, 0, nseg, NS, NS, options(), rn.data(), 8, nn.data(), 3) == 0);
        CHECK(nn == n0 && memcmp(rn.data(), r0.data(), r0.size() * sizeof(decoder_results)) == 0);
        unsetenv("WSPR_NODE_FAIL_PEER");
        setenv("WSPR_NODE_FAIL_SHARD", "2", 1);
        CHECK(wspr_decode_batch_node_device(dI, dQ, 0, nseg, NS, NS, options(), rn.data(), 8, nn.data(), 3) < 0);
        CHECK(std::all_of(nn.begin(), nn.end(), [](int v) { return v == 0; }));
        unsetenv("WSPR_NODE_FAIL_SHARD");
        (void)hipFree(dI); (void)hipFree(dQ);
    }
    printf("released %zu bytes\n", wspr_release_buffers());
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "host";
    host_part();
    if (mode == "gpu") gpu_part(argc > 2 ? atoi(argv[2]) : 384);
    printf("%s: %d failed checks\n", g_fail ? "SANITIZE DRIVER FAILED" : "SANITIZE DRIVER OK", g_fail);
    return g_fail ? 1 : 0;
}

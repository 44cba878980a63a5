// Test helper: compares wspr::glibc_sinf/cosf (the device routine, compiled here for
// the host) with the host libm over float bit patterns [lo, hi) with a stride.
// Returns the number of mismatching inputs. Built by tests/test_sincosf.py with g++.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include "../../rtlsdr-wsprd_amd/csrc/kernels/glibc_sincosf.h"

extern "C" long sincosf_mismatches(uint32_t lo, uint32_t hi, uint32_t stride, int nthreads, uint32_t* first_bad) {
    std::atomic<long> bad{0};
    std::atomic<uint32_t> first{0xffffffffu};
    auto work = [&](int t) {
        long local = 0;
        for (uint64_t b = (uint64_t)lo + (uint64_t)t * stride; b < hi; b += (uint64_t)stride * nthreads) {
            uint32_t u = (uint32_t)b;
            float x; std::memcpy(&x, &u, 4);
            for (int sgn = 0; sgn < 2; sgn++) {
                float v = sgn ? -x : x;
                float a = sinf(v), c = cosf(v);
                float a2 = wspr::glibc_sinf(v), c2 = wspr::glibc_cosf(v);
                float a3, c3;
                wspr::glibc_sincosf_pair(v, &a3, &c3);
                if (std::memcmp(&a2, &a3, 4) || std::memcmp(&c2, &c3, 4)) {
                    if (!(a2 != a2 && a3 != a3)) local++;
                }
                if (std::memcmp(&a, &a2, 4) || std::memcmp(&c, &c2, 4)) {
                    if (!(a != a && a2 != a2 && c != c && c2 != c2)) {   // both-NaN is fine
                        local++;
                        uint32_t cur = first.load();
                        while (u < cur && !first.compare_exchange_weak(cur, u)) {}
                    }
                }
            }
        }
        bad += local;
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
    for (auto& t : th) t.join();
    if (first_bad) *first_bad = first.load();
    return bad.load();
}

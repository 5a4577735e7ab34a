#define private public
#include "../../../leon_amd/csrc/host_rc.h"
#include <cstdio>
#include <random>
using namespace leon;
// the chain alone on records produced beforehand (in DRAM, or a 2 MB set reused: argv[2] = 1)
int main(int argc, char** argv) {
    const uint32_t k = 31;
    const size_t n = argc > 1 ? atol(argv[1]) : 4000000;
    std::mt19937_64 rng(1);
    std::vector<uint64_t> km(n);
    for (auto& x : km) x = rng() >> 2;
    std::vector<ChainRec16> recs(n * k);
    uint64_t c[4] = {1, 1, 1, 1}, t = 0;
    ChainRec16* out = recs.data();
    for (size_t a = 0; a < n; a++) for (uint32_t i = 0; i < k; i++, t++, out++) {
        const uint32_t sy = (uint32_t)(km[a] >> (2 * (k - 1 - i))) & 3u;
        const uint64_t c01 = c[0] + c[1], lo = sy == 0 ? 0 : sy == 1 ? c[0] : sy == 2 ? c01 : c01 + c[2], fr = c[sy];
        uint64_t q, r; const uint64_t d = 5 + t + 1;
        asm("divq %[d]" : "=a"(q), "=d"(r) : "a"(0ull), "d"(fr), [d] "r"(d) : "cc");
        out->c = q; out->lo = (uint32_t)lo; out->fr = (uint32_t)fr; c[sy]++;
    }
    for (int rep = 0; rep < 3; rep++) {
        AnchorDictCoder cd;
        const size_t plain = 20;
        cd.encode_kmers(km.data(), plain, k);
        auto t0 = std::chrono::steady_clock::now();
        cd.encode_records(recs.data() + plain * k, (n - plain) * k, plain * k);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        cd.flush();
        printf("NEW consumer alone, records from DRAM: %.3f ns/symbol, %zu bytes\n", dt / ((n - plain) * (double)k) * 1e9, cd.size());
    }
    {   // a cache-resident stretch of records, over and over (the state keeps evolving; output discarded by rewinding)
        AnchorDictCoder cd;
        cd.encode_kmers(km.data(), 20, k);
        const size_t m = 60000, start = (n > 1100000 ? 1000000 : n / 2) * (size_t)k;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 300; i++) { cd.w_ = 0; cd.encode_records(recs.data() + start, m, start); }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("NEW consumer alone, records in cache: %.3f ns/symbol\n", dt / (300.0 * m) * 1e9);
    }
}

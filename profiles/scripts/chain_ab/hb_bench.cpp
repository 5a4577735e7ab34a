// The host chains of small launches (leon_amd/csrc/host_blocks.h) alone: ns per symbol of HostBlockCoder::code on records made here
// from a synthetic block (the DNA stream's model mix).  No GPU involved.
#include "../../../leon_amd/csrc/host_blocks.h"
#include <chrono>
#include <cstdio>
#include <random>
using namespace leon;
int main(int argc, char** argv) {
    const size_t n = argc > 1 ? atol(argv[1]) : 1160000;
    std::mt19937_64 rng(3);
    const uint32_t sizes_small[8] = {2, 5, 5, 2, 3, 3, 3, 2};
    std::vector<std::vector<uint32_t>> freq(80);
    for (uint32_t m = 0; m < 80; m++) freq[m].assign(m < 8 ? sizes_small[m] : 256, 1);
    std::vector<uint64_t> recs(n);
    for (size_t i = 0; i < n; i++) {
        // ~35 % small models, the rest numeric byte-count / low-byte models, values skewed
        uint32_t m = (rng() % 100) < 35 ? (uint32_t)(rng() % 8) : 8 + 9 * (uint32_t)(rng() % 8) + (uint32_t)((rng() % 10) < 5 ? 0 : 1 + rng() % 3);
        const uint32_t sz = (uint32_t)freq[m].size();
        uint32_t v = (rng() % 10) < 7 ? (uint32_t)(rng() % (sz < 4 ? sz : 4)) : (uint32_t)(rng() % sz);
        uint64_t lo = 0;
        for (uint32_t x = 0; x < v; x++) lo += freq[m][x];
        recs[i] = lo | ((uint64_t)freq[m][v] << 22) | ((uint64_t)m << 44);
        freq[m][v]++;
    }
    (void)hb_recip_table();
    HostBlockCoder c;
    for (int rep = 0; rep < 5; rep++) {
        c.start(0x23332552u, 8);
        auto t0 = std::chrono::steady_clock::now();
        c.code(recs.data(), n);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        c.flush();
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < c.size(); i++) { h ^= c.data()[i]; h *= 1099511628211ull; }
        printf("%.3f ns per symbol, %zu bytes, fnv %016llx\n", dt / n * 1e9, c.size(), (unsigned long long)h);
    }
}

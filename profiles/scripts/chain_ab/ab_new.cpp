#include "../../../leon_amd/csrc/host_rc.h"
#include <cstdio>
#include <random>
using namespace leon;
int main(int argc, char** argv) {
    const uint32_t k = argc > 2 ? atoi(argv[2]) : 31;
    const size_t n = argc > 1 ? atol(argv[1]) : 4000000;
    const uint32_t W = k >= 32 ? 2 : 1;
    std::mt19937_64 rng(1);
    std::vector<uint64_t> km(n * W);
    for (size_t a = 0; a < n; a++) { if (W == 1) km[a] = rng() >> (64 - 2 * k); else { km[2 * a] = rng(); km[2 * a + 1] = rng() >> (128 - 2 * k); } }
    for (int rep = 0; rep < 3; rep++) {
        AnchorDictWorker w(k);
        auto t0 = std::chrono::steady_clock::now();
        const size_t step = n / 16;
        for (size_t i = 0; i < n; i += step) { size_t m = std::min(step, n - i); w.push(std::vector<uint64_t>(km.begin() + i * W, km.begin() + (i + m) * W)); }
        w.drain();
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        w.coder().flush();
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < w.coder().size(); i++) { h ^= w.coder().data()[i]; h *= 1099511628211ull; }
        printf("NEW chain k=%u: %.3f ns/symbol wall, %.3f ns/symbol busy, %zu bytes, fnv %016llx\n", k, dt / (n * (double)k) * 1e9, w.busy_ms() * 1e6 / (n * (double)k), w.coder().size(), (unsigned long long)h);
    }
}

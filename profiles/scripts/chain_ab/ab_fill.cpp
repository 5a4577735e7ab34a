// one helper's work (ChainFeed::fill): k-mers -> 16-byte records {C = floor(fr * 2^64 / d), lo, fr}; the form of the first
// round-3 build (hardware division, counts in memory) against a reshaped one (double-precision estimate + exact fix-up,
// cumulative counts in registers)
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
struct Rec { uint64_t c; uint32_t lo, fr; };
static inline uint64_t scaled_div(uint64_t fr, uint64_t d) { uint64_t q, r; asm("divq %[d]" : "=a"(q), "=d"(r) : "a"(0ull), "d"(fr), [d] "r"(d) : "cc"); return q; }
static inline uint64_t scaled_fp(uint64_t fr, uint64_t d) {
    const double inv = 1.0 / (double)d;
    uint64_t q0 = (uint64_t)((double)fr * 18446744073709551616.0 * inv * (1.0 - 0x1p-50));   // at most 2^15 below the quotient, never above
    uint64_t r = 0 - q0 * d;                                    // fr * 2^64 - q0 * d, exact: it is below 2^15 * d
    const uint64_t q1 = (uint64_t)((double)r * inv);
    q0 += q1; r -= q1 * d;
    if ((int64_t)r < 0) { q0--; r += d; }
    if (r >= d) { q0++; r -= d; }
    if (r >= d) q0 = scaled_div(fr, d);
    return q0;
}
static void fill_old(Rec* out, const uint64_t* km, size_t n, uint32_t k, uint64_t t0, uint64_t* c) {
    uint64_t d = 5 + t0 + 1;
    for (size_t a = 0; a < n; a++) {
        const uint64_t* w = km + a;
        for (uint32_t i = 0; i < k; i++, d++, out++) {
            const uint32_t bit = 2 * (k - 1 - i);
            const uint32_t sy = (uint32_t)(w[bit >> 6] >> (bit & 63)) & 3u;
            const uint64_t c01 = c[0] + c[1];
            const uint64_t lo = sy == 0 ? 0 : sy == 1 ? c[0] : sy == 2 ? c01 : c01 + c[2];
            const uint64_t fr = c[sy];
            out->c = scaled_div(fr, d); out->lo = (uint32_t)lo; out->fr = (uint32_t)fr;
            c[sy]++;
        }
    }
}
template <bool FP> static void fill_new(Rec* out, const uint64_t* km, size_t n, uint32_t k, uint64_t t0, uint64_t* c) {
    uint64_t d = 5 + t0 + 1;
    uint64_t cum[5] = {0, c[0], c[0] + c[1], c[0] + c[1] + c[2], c[0] + c[1] + c[2] + c[3]};   // cum[j] = counts of the symbols below j
    for (size_t a = 0; a < n; a++) {
        uint64_t x = km[a] << (64 - 2 * k);                    // first base in the two highest bits
        for (uint32_t i = 0; i < k; i++, d++, out++, x <<= 2) {
            const uint32_t sy = (uint32_t)(x >> 62);
            const uint64_t lo = cum[sy], fr = cum[sy + 1] - lo;
            out->c = FP ? scaled_fp(fr, d) : scaled_div(fr, d); out->lo = (uint32_t)lo; out->fr = (uint32_t)fr;
            cum[1] += sy < 1; cum[2] += sy < 2; cum[3] += sy < 3; cum[4]++;
        }
    }
    c[0] = cum[1]; c[1] = cum[2] - cum[1]; c[2] = cum[3] - cum[2]; c[3] = cum[4] - cum[3];
}
int main() {
    const uint32_t k = 31;
    const size_t n = 1500000;
    std::mt19937_64 rng(2);
    std::vector<uint64_t> km(n);
    for (auto& x : km) x = rng() >> 2;
    std::vector<Rec> r0(n * k), r1(n * k);
    for (int variant = 0; variant < 3; variant++) {
        uint64_t c[4] = {1, 1, 1, 1};
        auto t0 = std::chrono::steady_clock::now();
        if (variant == 0) fill_old(r0.data(), km.data(), n, k, 600, c);
        else if (variant == 1) fill_new<false>(r1.data(), km.data(), n, k, 600, c);
        else fill_new<true>(r1.data(), km.data(), n, k, 600, c);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        bool same = true;
        if (variant) for (size_t i = 0; i < n * k; i++) if (r0[i].c != r1[i].c || r0[i].lo != r1[i].lo || r0[i].fr != r1[i].fr) { same = false; break; }
        printf("%s: %.2f ns per symbol%s\n", variant == 0 ? "division, counts in memory (first build)" : variant == 1 ? "division, cumulative counts in registers" : "double estimate + exact fix-up, cumulative counts in registers",
               dt / (n * k) * 1e9, variant ? (same ? "  == first build" : "  DIFFERS") : "");
    }
    std::mt19937_64 r2(5);                                       // exactness of the estimate on operands up to 2^34
    uint64_t bad = 0;
    for (int i = 0; i < 30000000; i++) {
        uint64_t d = (r2() >> (30 + r2() % 24)); if (d < 513) d = 513 + (d & 63);
        uint64_t fr = 1 + r2() % (d - 1);
        if (i % 3 == 0) fr = d - 1 - (r2() % 3 % (d - 1));
        if (i % 7 == 0) fr = 1 + r2() % 3;
        if (scaled_fp(fr, d) != scaled_div(fr, d)) bad++;
    }
    printf("mismatches in 30 M random (fr, d), d up to 2^34: %llu\n", (unsigned long long)bad);
}

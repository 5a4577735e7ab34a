// host_blocks.h -- read blocks' range-coder chains on HOST cores, from the records the device's modeler waves make
// (rc_kernels.hip k_rc_records: cumLow | freq << 22 | model << 44 per symbol).  A launch of a few hundred blocks leaves the GPU
// waiting for ONE block's serial chain per CU (~280 cycles per symbol on a lone wave); a host core runs the same chain
// (RangeEncoder::encode, gatb RangeCoder.cpp [RECALLED]: range /= total; low += cumLow * range; range *= freq; renormalise)
// in ~10 cycles per symbol.  The models' totals are not in the records: a total is the model's alphabet size plus the
// symbols coded on it so far, which the chain counts itself.
#pragma once
#include <stdint.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace leon {

constexpr uint32_t HB_COUNT_BITS = 22;                        // a record's counts: blocks of up to 2^22 - 512 symbols
constexpr uint64_t HB_COUNT_MASK = (1ull << HB_COUNT_BITS) - 1;

// floor((2^72 - 1) / d) for 256 < d < 2^22 (0 below): with h = mulhi(x, m), floor(x / d) = h >> 8 or one more (x < 2^64)
inline const uint64_t* hb_recip_table() {
    static uint64_t* table = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        uint64_t* t = static_cast<uint64_t*>(malloc(sizeof(uint64_t) << HB_COUNT_BITS));
        for (uint64_t d = 0; d < (1ull << HB_COUNT_BITS); d++)
            t[d] = d <= 256 ? 0 : (uint64_t)((((((unsigned __int128)255) << 64) | ~0ull)) / d);
        table = t;
    });
    return table;
}

class HostBlockCoder {
public:
    // AbstractDnaCoder::startBlock: every model back to Order0Model::clear (total = alphabet size)
    void start(uint32_t small_sizes, uint32_t n_small) {
        low_ = 0; range_ = ~0ull; w_ = 0;
        for (uint32_t m = 0; m < 128; m++) tot_[m] = m < n_small ? ((small_sizes >> (4 * m)) & 15u) : 256u;
    }
    // RangeEncoder::encode for one record
    static inline __attribute__((always_inline)) void step(uint64_t w, uint32_t* tot, const uint64_t* T, uint64_t& low, uint64_t& range, uint8_t*& p,
                                                           const uint64_t kTop, const uint64_t kBottom) {
        const uint64_t lo = w & HB_COUNT_MASK, fr = (w >> HB_COUNT_BITS) & HB_COUNT_MASK;
        const uint32_t m = (uint32_t)(w >> (2 * HB_COUNT_BITS)) & 127u;
        const uint64_t t = tot[m]++;
        uint64_t q;
        if (__builtin_expect(t > 256, 1)) {
            q = (uint64_t)(((unsigned __int128)range * T[t]) >> 64) >> 8;
            uint64_t rem = range - q * t;
            while (__builtin_expect(rem >= t, 0)) { q++; rem -= t; }     // (one short with probability < 2^-8)
        } else q = range / t;
        // both ends' products start together (the flag's path: multiply, add, xor, compare, select); the range is their difference
        const uint64_t Lu = low + q * lo, Tu = low + q * (lo + fr);
        const uint64_t Ru = Tu - Lu, x = Lu ^ Tu;
        if (__builtin_expect(Ru >= kBottom, 1)) {          // (x < 2^48 implies Ru < 2^48: adding Ru flips a bit of Lu at or above Ru's highest one)
            // no byte, or exactly one: selected without a branch (a byte leaves after every fourth symbol or so)
            *p = (uint8_t)(Lu >> 56);
            low = Lu; range = Ru;
#if defined(__x86_64__)
            asm("cmpq %[top], %[x]\n\tcmovbq %[L8], %[low]\n\tcmovbq %[R8], %[range]\n\tadcq $0, %[p]"
                : [low] "+r"(low), [range] "+r"(range), [p] "+r"(p)
                : [x] "r"(x), [top] "r"(kTop), [L8] "r"(Lu << 8), [R8] "r"(Ru << 8)
                : "cc");
#else
            const uint64_t one = x < kTop ? 1 : 0, sh = one << 3;
            p += one; low <<= sh; range <<= sh;
#endif
        } else {
            low = Lu; range = Ru;
            while ((low ^ (low + range)) < kTop || (range < kBottom && ((range = (0 - low) & (kBottom - 1)), true))) {
                *p++ = (uint8_t)(low >> 56); range <<= 8; low <<= 8;
            }
        }
    }
    // the next n symbols of the block
    void code(const uint64_t* rec, uint64_t n) {
        static constexpr uint64_t kTop = 1ull << 56, kBottom = 1ull << 48;
        const uint64_t* const T = hb_recip_table();
        uint64_t low = low_, range = range_;
        for (uint64_t i0 = 0; i0 < n; i0 += 4096) {
            const uint64_t i1 = i0 + 4096 < n ? i0 + 4096 : n;
            if (out_.size() < w_ + 8 * (i1 - i0) + 64) out_.resize(out_.size() * 2 + 8 * (i1 - i0) + 4096);
            uint8_t* p = out_.data() + w_;
            for (uint64_t i = i0; i < i1; i++) step(rec[i], tot_, T, low, range, p, kTop, kBottom);
            w_ = (size_t)(p - out_.data());
        }
        low_ = low; range_ = range;
    }
    // Two blocks' next symbols, one of each per iteration: a chain is a dozen DEPENDENT cycles per symbol (multiply-high by the reciprocal,
    // the ends' products, the select) in which the core has room for a second one -- two chains side by side take about as long as one.
    // Same bytes as code() on each: the steps are code()'s own.
    static void code2(HostBlockCoder& A, const uint64_t* ra, uint64_t na, HostBlockCoder& B, const uint64_t* rb, uint64_t nb) {
        static constexpr uint64_t kTop = 1ull << 56, kBottom = 1ull << 48;
        const uint64_t* const T = hb_recip_table();
        const uint64_t n = na < nb ? na : nb;
        uint64_t lowA = A.low_, rangeA = A.range_, lowB = B.low_, rangeB = B.range_;
        for (uint64_t i0 = 0; i0 < n; i0 += 4096) {
            const uint64_t i1 = i0 + 4096 < n ? i0 + 4096 : n;
            if (A.out_.size() < A.w_ + 8 * (i1 - i0) + 64) A.out_.resize(A.out_.size() * 2 + 8 * (i1 - i0) + 4096);
            if (B.out_.size() < B.w_ + 8 * (i1 - i0) + 64) B.out_.resize(B.out_.size() * 2 + 8 * (i1 - i0) + 4096);
            uint8_t* pA = A.out_.data() + A.w_;
            uint8_t* pB = B.out_.data() + B.w_;
            for (uint64_t i = i0; i < i1; i++) {
                step(ra[i], A.tot_, T, lowA, rangeA, pA, kTop, kBottom);
                step(rb[i], B.tot_, T, lowB, rangeB, pB, kTop, kBottom);
            }
            A.w_ = (size_t)(pA - A.out_.data());
            B.w_ = (size_t)(pB - B.out_.data());
        }
        A.low_ = lowA; A.range_ = rangeA; B.low_ = lowB; B.range_ = rangeB;
        if (na > n) A.code(ra + n, na - n);
        if (nb > n) B.code(rb + n, nb - n);
    }
    void flush() {                                             // RangeEncoder::flush
        if (out_.size() < w_ + 8) out_.resize(w_ + 64);
        for (int i = 0; i < 8; i++) { out_[w_++] = (uint8_t)(low_ >> 56); low_ <<= 8; }
    }
    const uint8_t* data() const { return out_.data(); }
    size_t size() const { return w_; }
private:
    uint64_t low_ = 0, range_ = ~0ull;
    uint32_t tot_[128];
    std::vector<uint8_t> out_;
    size_t w_ = 0;
};

}  // namespace leon

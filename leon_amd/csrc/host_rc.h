// host_rc.h -- host-side order-0 range coder used for the ONE stream of the format that is a single serial
// chain over the whole file: the anchor dictionary (Leon::encodeInsertedAnchor -> _anchorRangeEncoder with
// _anchorDictModel(5) [RECALLED]).  ~31 symbols per inserted anchor, adaptive, no block structure, so it has
// no data parallelism to give a GPU; it runs on a host thread overlapped with the device stages.
// (The read blocks -- the bulk of the symbols -- are coded on the device: rc_kernels.hip.)
#pragma once
#include <stdint.h>
#include <vector>

namespace leon {

class HostOrder0Model {
public:
    explicit HostOrder0Model(uint32_t n) : n_(n), r_(n + 1) { clear(); }
    void clear() { for (uint32_t i = 0; i <= n_; i++) r_[i] = i; }
    uint64_t low(uint32_t c) const { return r_[c]; }
    uint64_t high(uint32_t c) const { return r_[c + 1]; }
    uint64_t total() const { return r_[n_]; }
    uint32_t size() const { return n_; }
    void update(uint32_t c) {
        for (uint32_t i = c + 1; i <= n_; i++) r_[i] += 1;
        if (r_[n_] >= kMaxRange) rescale();
    }
private:
    static constexpr uint64_t kMaxRange = 1ull << 48;
    void rescale() {
        for (uint32_t i = 1; i <= n_; i++) {
            r_[i] /= 2;
            if (r_[i] <= r_[i - 1]) r_[i] = r_[i - 1] + 1;
        }
    }
    uint32_t n_;
    std::vector<uint64_t> r_;
};

class HostRangeEncoder {
public:
    void clear() { low_ = 0; range_ = ~0ull; buf_.clear(); }
    void encode(HostOrder0Model& m, uint32_t c) {
        range_ /= m.total();
        low_ += m.low(c) * range_;
        range_ *= m.high(c) - m.low(c);
        while ((low_ ^ (low_ + range_)) < kTop || (range_ < kBottom && ((range_ = (0 - low_) & (kBottom - 1)), true))) {
            buf_.push_back((uint8_t)(low_ >> 56));
            range_ <<= 8;
            low_ <<= 8;
        }
        m.update(c);
    }
    void flush() {
        for (int i = 0; i < 8; i++) { buf_.push_back((uint8_t)(low_ >> 56)); low_ <<= 8; }
    }
    const std::vector<uint8_t>& bytes() const { return buf_; }
private:
    static constexpr uint64_t kTop = 1ull << 56, kBottom = 1ull << 48;
    uint64_t low_ = 0, range_ = ~0ull;
    std::vector<uint8_t> buf_;
};

}  // namespace leon

// host_rc.h -- host-side coder of the ONE stream of the format that is a single serial chain over the whole
// file: the anchor dictionary (Leon::encodeInsertedAnchor -> _anchorRangeEncoder with _anchorDictModel(5)
// [RECALLED]): k symbols per inserted anchor, adaptive, never reset, so it has no data parallelism to give a
// GPU (a lone wave runs such a chain ~50x slower than a CPU core).  It runs on a host thread, fed window by
// window while the device resolves, walks and codes the read blocks (rc_kernels.hip codes those).
#pragma once
#include <stdint.h>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace leon {

// Order-0 adaptive model over {A,C,T,G,N} + carry-less 64-bit range coder, specialised for the dictionary
// stream (5 symbols, cumulative counts updated branch-free, output written through a raw cursor): ~8 ns/symbol.
class AnchorDictCoder {
public:
    AnchorDictCoder() { clear(); }
    void clear() {
        low_ = 0; range_ = ~0ull; n_ = 0; buf_.clear(); w_ = 0;
        for (int i = 0; i <= 5; i++) cum_[i] = i;              // Order0Model::clear: _charRanges[i] = i
    }
    // k-mer in the 2-bit code (w[0] low word, w[1] high word when k >= 32), first base in the highest bits
    // (LargeInt::toString order)
    inline void encode_kmer(const uint64_t* w, uint32_t k) {
        if (buf_.size() < w_ + 8 * (size_t)k + 16) buf_.resize(buf_.size() * 2 + 8 * (size_t)k + 4096);
        for (uint32_t i = 0; i < k; i++) {
            const uint32_t bit = 2 * (k - 1 - i);
            encode((uint32_t)(w[bit >> 6] >> (bit & 63)) & 3u);
        }
    }
    void flush() {                                             // RangeEncoder::flush
        if (buf_.size() < w_ + 8) buf_.resize(w_ + 8);
        for (int i = 0; i < 8; i++) { buf_[w_++] = (uint8_t)(low_ >> 56); low_ <<= 8; }
    }
    const uint8_t* data() const { return buf_.data(); }
    size_t size() const { return w_; }
    uint64_t symbols() const { return n_; }
private:
    // one symbol; the caller guarantees 8 bytes of room (a symbol emits at most 8)
    inline void encode(uint32_t c) {
        const uint64_t lo = cum_[c], fr = cum_[c + 1] - cum_[c];
        range_ /= cum_[5];
        low_ += lo * range_;
        range_ *= fr;
        uint8_t* p = buf_.data() + w_;
        while ((low_ ^ (low_ + range_)) < kTop || (range_ < kBottom && ((range_ = (0 - low_) & (kBottom - 1)), true))) {
            *p++ = (uint8_t)(low_ >> 56);
            range_ <<= 8;
            low_ <<= 8;
        }
        w_ = (size_t)(p - buf_.data());
        for (uint32_t i = 1; i <= 5; i++) cum_[i] += (uint64_t)(i > c);    // Order0Model::update, branch-free
        n_++;                                                  // (its rescale needs 2^48 symbols: unreachable)
    }
    static constexpr uint64_t kTop = 1ull << 56, kBottom = 1ull << 48;
    uint64_t low_, range_, n_, cum_[6];
    std::vector<uint8_t> buf_;
    size_t w_;
};

// the worker that owns the coder: batches of anchor k-mers are queued in address order
class AnchorDictWorker {
public:
    explicit AnchorDictWorker(uint32_t k) : k_(k) {}
    ~AnchorDictWorker() { stop(); }
    void push(std::vector<uint64_t>&& kmers) {
        {
            std::lock_guard<std::mutex> g(mu_);
            if (!running_) { running_ = true; quit_ = false; th_ = std::thread([this] { run(); }); }
            q_.emplace_back(std::move(kmers));
            pending_++;
        }
        cv_.notify_all();
    }
    void drain() {                                             // wait until everything queued so far is coded
        std::unique_lock<std::mutex> g(mu_);
        cv_done_.wait(g, [this] { return pending_ == 0; });
    }
    void stop() {
        {
            std::lock_guard<std::mutex> g(mu_);
            if (!running_) return;
            quit_ = true;
        }
        cv_.notify_all();
        th_.join();
        running_ = false;
    }
    void reset() { drain(); coder_.clear(); }
    AnchorDictCoder& coder() { return coder_; }                // only after drain()
private:
    void run() {
        for (;;) {
            std::vector<uint64_t> batch;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return quit_ || !q_.empty(); });
                if (q_.empty()) return;
                batch = std::move(q_.front());
                q_.pop_front();
            }
            const uint32_t W = k_ >= 32 ? 2u : 1u;
            for (size_t a = 0; a + W <= batch.size(); a += W) coder_.encode_kmer(batch.data() + a, k_);
            {
                std::lock_guard<std::mutex> g(mu_);
                pending_--;
            }
            cv_done_.notify_all();
        }
    }
    uint32_t k_;
    AnchorDictCoder coder_;
    std::mutex mu_;
    std::condition_variable cv_, cv_done_;
    std::deque<std::vector<uint64_t>> q_;
    uint64_t pending_ = 0;
    bool running_ = false, quit_ = false;
    std::thread th_;
};

}  // namespace leon

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

// The dictionary stream's model total is 5 + t at symbol t, whatever the data: a helper thread runs ahead and hands
// the chain floor((2^64-1)/(5+t)) in chunks, so the chain divides with one 64x64->128 multiply and a one-step fix-up
// instead of a hardware division on its critical path (-15 % per symbol on an EPYC 9575F).
class ReciprocalStream {
public:
    static constexpr uint64_t kChunk = 1ull << 18;
    static constexpr uint32_t kBufs = 8;
    ReciprocalStream() { for (auto& b : buf_) b.resize(kChunk); }
    ~ReciprocalStream() { stop(); }
    void restart() {                                           // a new stream: t starts at 0 again
        stop();
        produced_ = consumed_ = 0; quit_ = false;
        th_ = std::thread([this] { run(); });
        running_ = true;
    }
    void stop() {
        if (!running_) return;
        { std::lock_guard<std::mutex> g(mu_); quit_ = true; }
        cv_.notify_all();
        th_.join();
        running_ = false;
    }
    // reciprocals of chunk c (symbols c*kChunk ...); chunks are taken in increasing order, taking c releases c-1
    const uint64_t* take(uint64_t c) {
        std::unique_lock<std::mutex> g(mu_);
        if (c > consumed_) { consumed_ = c; cv_.notify_all(); }
        cv_.wait(g, [&] { return produced_ > c; });
        return buf_[c % kBufs].data();
    }
private:
    void run() {
        for (;;) {
            uint64_t c;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return quit_ || produced_ - consumed_ < kBufs; });
                if (quit_) return;
                c = produced_;
            }
            uint64_t* out = buf_[c % kBufs].data();
            const uint64_t t0 = c * kChunk;
            for (uint64_t i = 0; i < kChunk; i++) out[i] = ~0ull / (5 + t0 + i);
            { std::lock_guard<std::mutex> g(mu_); produced_ = c + 1; }
            cv_.notify_all();
        }
    }
    std::vector<uint64_t> buf_[kBufs];
    std::mutex mu_;
    std::condition_variable cv_;
    uint64_t produced_ = 0, consumed_ = 0;
    bool quit_ = false, running_ = false;
    std::thread th_;
};

// Order-0 adaptive model over {A,C,T,G,N} + carry-less 64-bit range coder, specialised for the dictionary
// stream (5 symbols, cumulative counts updated branch-free, output written through a raw cursor): ~4-5 ns/symbol.
class AnchorDictCoder {
public:
    AnchorDictCoder() { clear(); }
    void clear() {
        low_ = 0; range_ = ~0ull; n_ = 0; buf_.clear(); w_ = 0; inv_ = nullptr; inv_chunk_ = ~0ull;
        for (int i = 0; i <= 5; i++) cum_[i] = i;              // Order0Model::clear: _charRanges[i] = i
    }
    void use_reciprocals(ReciprocalStream* r) { recips_ = r; }
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
        const uint64_t lo = cum_[c], fr = cum_[c + 1] - cum_[c], tot = cum_[5];
        uint64_t q;
        if (recips_) {                                         // floor(range / tot) = mulhi(range, floor((2^64-1)/tot)) or one more
            const uint64_t ch = n_ / ReciprocalStream::kChunk;
            if (ch != inv_chunk_) { inv_ = recips_->take(ch); inv_chunk_ = ch; }
            q = (uint64_t)(((unsigned __int128)range_ * inv_[n_ % ReciprocalStream::kChunk]) >> 64);
            q += (range_ - q * tot) >= tot;
        } else q = range_ / tot;
        low_ += lo * q;
        range_ = q * fr;
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
    ReciprocalStream* recips_ = nullptr;
    const uint64_t* inv_ = nullptr;
    uint64_t inv_chunk_ = ~0ull;
};

// the worker that owns the coder: batches of anchor k-mers are queued in address order
class AnchorDictWorker {
public:
    explicit AnchorDictWorker(uint32_t k) : k_(k) { coder_.use_reciprocals(&recips_); recips_.restart(); }
    ~AnchorDictWorker() { stop(); recips_.stop(); }
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
    void reset() { drain(); coder_.clear(); recips_.restart(); }
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
    ReciprocalStream recips_;
    AnchorDictCoder coder_;
    std::mutex mu_;
    std::condition_variable cv_, cv_done_;
    std::deque<std::vector<uint64_t>> q_;
    uint64_t pending_ = 0;
    bool running_ = false, quit_ = false;
    std::thread th_;
};

}  // namespace leon

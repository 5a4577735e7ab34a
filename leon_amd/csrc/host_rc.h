// host_rc.h -- host-side coder of the ONE stream of the format that is a single serial chain over the whole
// file: the anchor dictionary (Leon::encodeInsertedAnchor -> _anchorRangeEncoder with _anchorDictModel(5)
// [RECALLED]): k symbols per inserted anchor, adaptive, never reset, so it has no data parallelism to give a
// GPU (a lone wave runs such a chain ~50x slower than a CPU core).  It runs on a host thread, fed window by
// window while the device resolves, walks and codes the read blocks (rc_kernels.hip codes those).
#pragma once
#include <stdint.h>
#if defined(__x86_64__)
#include <emmintrin.h>
#define LEON_HOST_CHAIN_X86 1
#endif
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <new>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace leon {

// The dictionary stream's model total is 5 + t at symbol t, whatever the data: helper threads run ahead and hand the
// chain, in chunks, the scaled reciprocal m(t) = floor((2^72 - 1) / d), d = 5 + t > 256.  With h = mulhi(x, m):
// floor(256 x / d) = h or h + 1 when x < 2^56, and floor(x / d) = (h >> 8) or one more, "one more" having probability
// < 2^-8 in both cases.  The chain so divides with one multiply (and an immediate shift) and checks the quotient in a
// branch that is rarely taken, instead of waiting for a hardware division (4.1 -> ~2 ns per symbol on an EPYC 9575F
// together with the branch-free renormalisation below).
class ReciprocalStream {
public:
    static constexpr uint64_t kChunk = 1ull << 18;
    static constexpr uint32_t kBufs = 12, kThreads = 3;         // one 128/64-bit division per symbol: ~3 threads keep ahead
    ReciprocalStream() { for (auto& b : buf_) b.resize(kChunk); }
    ~ReciprocalStream() { stop(); }
    void restart() {                                           // a new stream: t starts at 0 again
        stop();
        consumed_ = 0; quit_ = false;
        for (auto& h : holds_) h = 0;
        for (uint32_t j = 0; j < kThreads; j++) th_[j] = std::thread([this, j] { run(j); });
        running_ = true;
    }
    void stop() {
        if (!running_) return;
        { std::lock_guard<std::mutex> g(mu_); quit_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
        running_ = false;
    }
    // reciprocals of chunk c (symbols c*kChunk ...); chunks are taken in increasing order, taking c releases those before it
    const uint64_t* take(uint64_t c) {
        std::unique_lock<std::mutex> g(mu_);
        if (c > consumed_) { consumed_ = c; cv_.notify_all(); }
        cv_.wait(g, [&] { return holds_[c % kBufs] == c + 1; });
        return buf_[c % kBufs].data();
    }
    static inline uint64_t reciprocal(uint64_t d) {            // floor((2^72 - 1) / d); 0 where it would not fit (d <= 256)
        if (d <= 256) return 0;
#ifdef LEON_HOST_CHAIN_X86
        uint64_t q, r;
        asm("divq %[d]" : "=a"(q), "=d"(r) : "a"(~0ull), "d"(255ull), [d] "r"(d) : "cc");
        return q;
#else
        return (uint64_t)(((((unsigned __int128)255) << 64) | ~0ull) / d);
#endif
    }
private:
    void run(uint32_t j) {
        for (uint64_t c = j;; c += kThreads) {
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return quit_ || c < consumed_ + kBufs; });   // the buffer's previous chunk is released
                if (quit_) return;
            }
            uint64_t* out = buf_[c % kBufs].data();
            const uint64_t t0 = c * kChunk;
            for (uint64_t i = 0; i < kChunk; i++) out[i] = reciprocal(5 + t0 + i);
            { std::lock_guard<std::mutex> g(mu_); holds_[c % kBufs] = c + 1; }
            cv_.notify_all();
        }
    }
    std::vector<uint64_t> buf_[kBufs];
    uint64_t holds_[kBufs] = {};                               // chunk index + 1 a buffer currently holds
    std::mutex mu_;
    std::condition_variable cv_;
    uint64_t consumed_ = 0;
    bool quit_ = false, running_ = false;
    std::thread th_[kThreads];
};

// Order-0 adaptive model over {A,C,T,G,N} + carry-less 64-bit range coder, specialised for the dictionary
// stream (5 symbols, cumulative counts updated branch-free, output written through a raw cursor): ~4-5 ns/symbol.
// the coder's output: grows by doubling, is never zero-filled, keeps its capacity from one stream to the next (a
// 100 M-read file's stream is 110 MB: std::vector's resize would clear all of it again for every file)
class ByteBuf {
public:
    ~ByteBuf() { free(p_); }
    uint8_t* data() { return p_; }
    const uint8_t* data() const { return p_; }
    size_t size() const { return cap_; }
    void clear() {}
    void resize(size_t n) {                                    // grow only
        if (n <= cap_) return;
        uint8_t* q = (uint8_t*)realloc(p_, n);
        if (!q) throw std::bad_alloc();
        p_ = q; cap_ = n;
    }
    uint8_t& operator[](size_t i) { return p_[i]; }
private:
    uint8_t* p_ = nullptr;
    size_t cap_ = 0;
};

class AnchorDictCoder {
public:
    AnchorDictCoder() { clear(); }
    void clear() {
        low_ = 0; range_ = ~0ull; n_ = 0; buf_.clear(); w_ = 0; inv_ = nullptr;
        for (int i = 0; i <= 5; i++) cum_[i] = i;              // Order0Model::clear: _charRanges[i] = i
    }
    void use_reciprocals(ReciprocalStream* r) { recips_ = r; }
    // `count` k-mers in the 2-bit code (W = 1 word per k-mer below k = 32, else low word then high word), first base in
    // the highest bits (LargeInt::toString order)
    void encode_kmers(const uint64_t* w, size_t count, uint32_t k) {
        const uint32_t W = k >= 32 ? 2u : 1u;
#ifdef LEON_HOST_CHAIN_X86
        while (count && (!recips_ || n_ < 256)) { encode_kmer_plain(w, k); w += W; count--; }   // (the scaled reciprocal needs a total above 256)
        if (!count) return;
        if (W == 2) encode_chain<true>(w, count, k); else encode_chain<false>(w, count, k);
#else
        for (; count; count--, w += W) encode_kmer_plain(w, k);
#endif
    }
    inline void encode_kmer(const uint64_t* w, uint32_t k) { encode_kmers(w, 1, k); }
    inline void encode_kmer_plain(const uint64_t* w, uint32_t k) {
        if (buf_.size() < w_ + 8 * (size_t)k + 16) buf_.resize(buf_.size() * 2 + 8 * (size_t)k + 4096);
        for (uint32_t i = 0; i < k; i++) {
            const uint32_t bit = 2 * (k - 1 - i);
            encode((uint32_t)(w[bit >> 6] >> (bit & 63)) & 3u);
        }
    }
    void flush() {                                             // RangeEncoder::flush
        if (buf_.size() < w_ + 24) buf_.resize(w_ + 24);
        settle();
        for (int i = 0; i < 8; i++) { buf_[w_++] = (uint8_t)(low_ >> 56); low_ <<= 8; }
    }
    const uint8_t* data() const { return buf_.data(); }
    size_t size() const { return w_; }
    uint64_t symbols() const { return n_; }
private:
#ifdef LEON_HOST_CHAIN_X86
    // `count` k-mers through the latency-shaped chain; kTwo: k >= 32 (two words per k-mer)
    template <bool kTwo> void encode_chain(const uint64_t* w, size_t count, uint32_t k) {
        // The state carried from symbol to symbol is (low, range) BEFORE the renormalisation the
        // previous symbol owes.  Its usual outcomes -- no byte, one byte -- are both formed and selected by conditional
        // moves: with h = mulhi(range, m) the quotient is h >> 8 without the byte and h itself with it (range < 2^56
        // there), so ONE multiply serves both.  The rare outcomes (two bytes at once, the carry-less coder's
        // range < BOTTOM reset, a quotient one too small) leave through branches that are almost never taken.
        // Loop-carried path: multiply-high, shift, select, multiply = ~9 cycles and no data-dependent branch, where
        // the plain form has a division, a multiply and a ~25 %-taken renormalisation branch.  The model's cumulative
        // counts live in two vector registers (one add each per symbol) and are read back through a small table.
        alignas(16) static const uint64_t kInc[4][4] = {{1, 1, 1, 1}, {0, 1, 1, 1}, {0, 0, 1, 1}, {0, 0, 0, 1}};
        alignas(16) uint64_t tbl[6] = {0, 0, cum_[1], cum_[2], cum_[3], cum_[4]};       // tbl[1 + c] = cum_[c]
        __m128i ca = _mm_load_si128((const __m128i*)(tbl + 2)), cb = _mm_load_si128((const __m128i*)(tbl + 4));
        uint64_t L = low_, R = range_, tot = 5 + n_;          // the model's total IS the symbol index + 5
        uint8_t* p = buf_.data() + w_;
        for (size_t a = 0; a < count; a++, w += kTwo ? 2 : 1) {
            if ((size_t)(buf_.data() + buf_.size() - p) < 8 * (size_t)k + 16) {
                const size_t used = (size_t)(p - buf_.data());
                buf_.resize(buf_.size() * 2 + 8 * (size_t)k + 4096);
                p = buf_.data() + used;
            }
            uint64_t x = w[0], xh = kTwo ? w[1] : 0;          // first base in the two highest bits of the (xh:)x register(s)
            if (kTwo) { const uint32_t s = 128 - 2 * k; xh = s ? (xh << s) | (x >> (64 - s)) : xh; x <<= s; }
            else x <<= 64 - 2 * k;
            for (uint64_t tot_k = tot + k; tot < tot_k;) {
                const uint64_t n = tot - 5, off = n % ReciprocalStream::kChunk;
                if (off == 0 || !inv_) inv_ = recips_->take(n / ReciprocalStream::kChunk);
                const uint64_t* ipb = inv_ + off - tot;        // ipb[tot] = the reciprocal of symbol n
                const uint64_t left = ReciprocalStream::kChunk - off;
                const uint64_t tot_e = left < tot_k - tot ? tot + left : tot_k;
                for (; tot < tot_e; tot++) {
                    uint32_t c;
                    if (kTwo) { c = (uint32_t)(xh >> 62); xh = (xh << 2) | (x >> 62); x <<= 2; }
                    else { c = (uint32_t)(x >> 62); x <<= 2; }
                    const uint64_t lo = tbl[1 + c], fr = tbl[2 + c] - lo, m = ipb[tot];
                    ca = _mm_add_epi64(ca, _mm_load_si128((const __m128i*)kInc[c]));       // Order0Model::update
                    cb = _mm_add_epi64(cb, _mm_load_si128((const __m128i*)kInc[c] + 1));
                    _mm_store_si128((__m128i*)(tbl + 2), ca);
                    _mm_store_si128((__m128i*)(tbl + 4), cb);
                    const uint64_t xr = L ^ (L + R);           // xr < TOP: the previous symbol owes (at least) one byte
                    uint64_t q;
                    // the branch-free form is valid for "no byte" and "exactly one byte" (the shifted test is xr << 8)
                    if (__builtin_expect((xr >> 48) == 0 || R < kBottom, 0)) {
                        while ((L ^ (L + R)) < kTop || (R < kBottom && ((R = (0 - L) & (kBottom - 1)), true))) {
                            *p++ = (uint8_t)(L >> 56);
                            R <<= 8;
                            L <<= 8;
                        }
                        q = (uint64_t)(((unsigned __int128)R * m) >> 64) >> 8;
                    } else {
                        const uint64_t h = (uint64_t)(((unsigned __int128)R * m) >> 64);
                        q = h >> 8;
                        *p = (uint8_t)(L >> 56);
                        // (q, R, L, p) = one byte ? (h, R << 8, L << 8, p + 1) : unchanged -- as conditional moves (a compiler
                        // turns the selects back into the unpredictable branch this loop exists to avoid)
                        asm("cmpq %[top], %[xr]\n\tcmovbq %[h], %[q]\n\tcmovbq %[R1], %[R]\n\tcmovbq %[L1], %[L]\n\tadcq $0, %[p]"
                            : [q] "+r"(q), [R] "+r"(R), [L] "+r"(L), [p] "+r"(p)
                            : [xr] "r"(xr), [top] "r"(kTop), [h] "r"(h), [R1] "r"(R << 8), [L1] "r"(L << 8)
                            : "cc");
                    }
                    if (__builtin_expect(R - q * tot >= tot, 0)) {  // quotient one too small: probability < 2^-8
                        asm volatile("" : "+r"(q));            // (keeps this a branch: as a select it would sit on the chain)
                        q++;
                    }
                    R = q * fr;
                    L += q * lo;
                }
            }
        }
        w_ = (size_t)(p - buf_.data());
        low_ = L; range_ = R; n_ = tot - 5;
        cum_[1] = tbl[2]; cum_[2] = tbl[3]; cum_[3] = tbl[4]; cum_[4] = tbl[5]; cum_[5] = tot;
    }
#endif
    // the renormalisation the last symbol owes (the state is kept un-normalised between symbols, see encode_kmer)
    inline void settle() {
        uint8_t* p = buf_.data() + w_;
        while ((low_ ^ (low_ + range_)) < kTop || (range_ < kBottom && ((range_ = (0 - low_) & (kBottom - 1)), true))) {
            *p++ = (uint8_t)(low_ >> 56);
            range_ <<= 8;
            low_ <<= 8;
        }
        w_ = (size_t)(p - buf_.data());
    }
    // one symbol, plain form (no reciprocal stream); the caller guarantees 16 bytes of room
    inline void encode(uint32_t c) {
        settle();
        const uint64_t lo = cum_[c], fr = cum_[c + 1] - cum_[c], tot = cum_[5];
        const uint64_t q = range_ / tot;
        low_ += lo * q;
        range_ = q * fr;
        for (uint32_t i = 1; i <= 5; i++) cum_[i] += (uint64_t)(i > c);    // Order0Model::update, branch-free
        n_++;                                                  // (its rescale needs 2^48 symbols: unreachable)
    }
    static constexpr uint64_t kTop = 1ull << 56, kBottom = 1ull << 48;
    uint64_t low_, range_, n_, cum_[6];
    ByteBuf buf_;
    size_t w_;
    ReciprocalStream* recips_ = nullptr;
    const uint64_t* inv_ = nullptr;
};

// The inverse of AnchorDictCoder (Leon::decodeAnchorDict / RangeDecoder on _anchorDictModel(5) [RECALLED]): one serial
// chain again, on a host core.  k symbols per anchor, first base in the highest bits.  Returns false on a corrupt stream.
// range / total uses the encoder's reciprocal stream (the total is 5 + t here too); the decoder's second division,
// (code - low) / range, is replaced by four multiplies: the symbol is the number of cumulative counts c with
// c * range <= code - low.
inline bool decode_anchor_dict(const uint8_t* p, uint64_t n, uint64_t n_anchors, uint32_t k, uint64_t* out) {
    constexpr uint64_t kTop = 1ull << 56, kBottom = 1ull << 48;
    const uint32_t W = k >= 32 ? 2u : 1u;
    uint64_t low = 0, range = ~0ull, code = 0, i = 0, t = 0, cum[6] = {0, 1, 2, 3, 4, 5};
    for (int b = 0; b < 8; b++) code = (code << 8) | (i < n ? p[i] : 0), i++;
    ReciprocalStream recips;
    const uint64_t* inv = nullptr;
    if (n_anchors * (uint64_t)k > (1u << 16)) recips.restart();   // (not worth three threads for a handful of symbols)
    const bool use_inv = n_anchors * (uint64_t)k > (1u << 16);
    for (uint64_t a = 0; a < n_anchors; a++) {
        unsigned __int128 km = 0;
        for (uint32_t j = 0; j < k; j++, t++) {
            const uint64_t tot = 5 + t;
            uint64_t r;
            if (use_inv && t >= 256) {
                const uint64_t off = t % ReciprocalStream::kChunk;
                if (off == 0 || !inv) inv = recips.take(t / ReciprocalStream::kChunk);
                r = (uint64_t)(((unsigned __int128)range * inv[off]) >> 64) >> 8;     // floor(range / tot) or one less
                if (range - r * tot >= tot) r++;
            } else r = range / tot;
            if (r == 0) return false;
            // the symbol is the number of cumulative counts c with c * r <= code - low; the five products also are the
            // new low and range (no second multiply, no division).  Selected with conditional moves: the symbol is as good
            // as random, a branch on it would be mispredicted three times out of four
            const uint64_t d = code - low;
            const uint64_t p1 = r * cum[1], p2 = r * cum[2], p3 = r * cum[3], p4 = r * cum[4], p5 = r * tot;
            const uint64_t g1 = d >= p1, g2 = d >= p2, g3 = d >= p3, g4 = d >= p4;
            const uint32_t c = (uint32_t)(g1 + g2 + g3 + g4);
            uint64_t lo = g1 ? p1 : 0, hi = g1 ? p2 : p1;
            lo = g2 ? p2 : lo; hi = g2 ? p3 : hi;
            lo = g3 ? p3 : lo; hi = g3 ? p4 : hi;
            lo = g4 ? p4 : lo; hi = g4 ? p5 : hi;
            low += lo;
            range = hi - lo;
            // renormalisation: nothing or one byte in the usual case, both formed and selected without a branch (a byte leaves
            // after every fourth symbol or so: as a branch that is a misprediction per byte); the rare cases take the loop
            {
                const uint64_t x = low ^ (low + range);
                const bool one = x < kTop;                       // the top byte is settled
                const uint64_t low1 = low << 8, range1 = range << 8, code1 = (code << 8) | (i < n ? p[i] : 0);
                const uint64_t x1 = low1 ^ (low1 + range1);
                if (__builtin_expect((one && (x1 < kTop || range1 < kBottom)) || (!one && range < kBottom), 0)) {
                    while ((low ^ (low + range)) < kTop || (range < kBottom && ((range = (0 - low) & (kBottom - 1)), true))) {
                        code = (code << 8) | (i < n ? p[i] : 0); i++;
                        range <<= 8;
                        low <<= 8;
                        if (i > n + 16) return false;              // far past the end: not a stream this coder wrote (a range of 0 would spin here)
                    }
                } else {
                    low = one ? low1 : low; range = one ? range1 : range; code = one ? code1 : code; i += one;
                }
            }
            cum[1] += (uint64_t)(c < 1); cum[2] += (uint64_t)(c < 2); cum[3] += (uint64_t)(c < 3); cum[4] += (uint64_t)(c < 4);   // Order0Model::update, branch-free
            if (c > 3) return false;                           // an N inside an anchor: not a stream this coder wrote
            km = (km << 2) | c;
        }
        out[a * W] = (uint64_t)km;
        if (W == 2) out[a * W + 1] = (uint64_t)(km >> 64);
    }
    return true;
}

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
    void reset() { drain(); coder_.clear(); recips_.restart(); busy_ms_ = 0; }
    double busy_ms() const { return busy_ms_; }              // time spent coding since the last reset (read after drain())
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
            const auto t0 = std::chrono::steady_clock::now();
            coder_.encode_kmers(batch.data(), batch.size() / W, k_);
            busy_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
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
    double busy_ms_ = 0;
    bool running_ = false, quit_ = false;
    std::thread th_;
};

}  // namespace leon

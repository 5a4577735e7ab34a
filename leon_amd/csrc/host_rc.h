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
#include <pthread.h>
#include <sched.h>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <new>
#include <algorithm>
#include <deque>
#include <memory>
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


// ---- the dictionary chain's feed -------------------------------------------------------------------------------
// Everything the chain needs about symbol t EXCEPT the coder state is known before the chain gets there: the model is
// order-0 with +1 updates, so a symbol's cumulative count lo and frequency fr are prefix counts of the symbols before it,
// and the total is 5 + t.  Helper threads run ahead of the chain and turn the k-mers into one record per symbol:
//     lo, fr and C = floor(fr * 2^64 / d), d = 5 + t + 1 the total of the NEXT symbol (one hardware division, fr < d).
// With those the chain carries the QUOTIENT q(t) = floor(range / total) instead of the range.  The new range is q * fr, so
//     floor(q * fr / d) = Phi           and   floor(q * fr * 2^8 / d) = (Phi << 8) | (Plo >> 56),     Phi:Plo = q * C,
// exactly, unless the product's fraction is within q of the next integer (probability ~ q / 2^56: then, rarely, a division).
// The next quotient so needs ONE multiply of the carried quotient and does not wait for the new range: the loop-carried
// path is multiply, add, xor, compare, select (~7 cycles; the renormalisation test low ^ (low + range) < TOP is what bounds
// it), with three multiplies per symbol, where carrying the range costs multiply-high, shift, select, multiply (~9) and four.
// LEON_CHAIN_PIN=1 (measurement: DESIGN.md 4.4): the chain and its helpers run on the CPUs that share one L3 -- the one the thread
// that starts them happens to be on -- so that the helpers' records reach the chain through that cache instead of across the fabric.
struct ChainCpus {
    bool use = false;
    cpu_set_t set;
    void pick() {
        const char* e = getenv("LEON_CHAIN_PIN");
        if (!e || e[0] != '1') return;
        const int cpu = sched_getcpu();
        if (cpu < 0) return;
        char path[128];
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
        FILE* f = fopen(path, "r");
        if (!f) return;
        char line[1024];
        const bool ok = fgets(line, sizeof line, f) != nullptr;
        fclose(f);
        if (!ok) return;
        cpu_set_t l3, allowed;
        CPU_ZERO(&l3);
        for (char* p = line; *p && *p != '\n';) {               // "0-7,128-135"
            char* q = nullptr;
            const long a = strtol(p, &q, 10);
            if (q == p) break;
            long b = a;
            if (*q == '-') { p = q + 1; b = strtol(p, &q, 10); }
            for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (c >= 0) CPU_SET((int)c, &l3);
            p = *q == ',' ? q + 1 : q;
            if (*q != ',') break;
        }
        if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
        CPU_AND(&set, &l3, &allowed);
        use = CPU_COUNT(&set) >= 7;                              // room for the chain and its helpers
    }
    void apply() const { if (use) (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set); }
};

struct ChainRec16 { uint64_t c; uint32_t lo, fr; };          // streams below 2^32 symbols (every count fits 32 bits)
struct ChainRec24 { uint64_t c, lo, fr; };

class ChainFeed {
public:
    static constexpr uint32_t kMaxBufs = 1024;                  // the ring's size is n_bufs_ (LEON_CHAIN_RING, default 512), at most this
    static constexpr uint64_t kPlainBelow = 512;               // symbols coded in the plain form at the start of a stream (m needs a total above 256)
    struct View {
        const void* recs = nullptr; uint64_t n_syms = 0;       // produced records (null for a plain segment): ChainRec16, or ChainRec24 when `wide`
        bool wide = false;
        const uint64_t* kmers = nullptr; uint32_t n_kmers = 0;
        uint64_t t0 = 0;                                       // index of the segment's first symbol in the stream
        uint64_t c_end[4] = {};                                // the model's counts (1 + occurrences of A, C, T, G) after the segment
    };
    explicit ChainFeed(uint32_t k) : k_(k), W_(k >= 32 ? 2u : 1u), seg_kmers_(std::max<uint32_t>(1, (1u << 15) / k)) {
        // ~4.8 ns per record and helper inside the library against the chain's ~1.8 per symbol: three helpers keep ahead of it, and
        // every further busy core lowers the clock the chain's own gets (with the deep ring: 786-808 ms per step with 3 helpers,
        // 795-817 with 4, 821-848 with 5, 840-859 with 7; profiles/r3_chain_helpers_ab.txt).  So three work all the time and
        // three SPARES only while the look-ahead is thin (the start of a stream, a host whose other tenants slow the helpers down).
        uint32_t n = 3, spares = 3;
        if (const char* e = getenv("LEON_CHAIN_HELPERS")) { const int v = atoi(e); if (v >= 1 && v <= 32) n = (uint32_t)v; }
        if (const char* e = getenv("LEON_CHAIN_SPARES")) { const int v = atoi(e); if (v >= 0 && v <= 32) spares = (uint32_t)v; }
        n_threads_ = n; n_spares_ = spares;
        // The ring: 512 segments of ~32 k symbols = ~30 ms of the chain's work ahead of it (12 MB of records per millisecond, touched
        // only as far as the helpers get ahead).  With 16 segments -- one millisecond -- a helper that lost its core for a time slice to
        // another tenant of the host stalled the chain, which takes its segments in order: 76-98 ms of a loaded host's 870-888 ms steps
        // were the chain waiting for records (LEON_TRACE_STEP=1 prints it), 1.3-1.8 ms of a quiet host's 822.
        if (const char* e = getenv("LEON_CHAIN_RING")) { const int v = atoi(e); if (v >= 4 && v <= (int)kMaxBufs) n_bufs_ = (uint32_t)v; }
    }
    ~ChainFeed() { stop(); }
    ChainCpus cpus_;                                           // (set by the owner before start())
    void start() {
        if (running_) return;
        quit_ = false;
        for (uint32_t b = 0; b < n_bufs_; b++) if (!buf_[b]) buf_[b].reset(new ChainRec24[(size_t)seg_kmers_ * k_]);      // (0.8 MB each, untouched until used; only for contexts that code a dictionary)
        for (uint32_t j = 0; j < n_threads_ + n_spares_; j++) th_.emplace_back([this, j] { run(j >= n_threads_); });
        running_ = true;
    }
    void stop() {
        if (!running_) return;
        { std::lock_guard<std::mutex> g(mu_); quit_ = true; }
        cv_work_.notify_all(); cv_spare_.notify_all(); cv_ready_.notify_all();
        for (auto& t : th_) t.join();
        th_.clear();
        running_ = false;
    }
    // a new stream (every segment pushed so far must have been taken and released)
    void reset_stream() {
        std::lock_guard<std::mutex> g(mu_);
        segs_.clear(); base_ = 0; next_claim_ = 0; next_take_ = 0; released_ = 0; prefix_upto_ = 0; t_pushed_ = 0;
        for (auto& c : prefix_c_) c = 1;
        for (auto& r : ready_) r = 0;
    }
    // `count` k-mers (W words each) that follow the ones pushed before; they must stay where they are until their segments
    // have been released.  Returns the number of segments they were cut into.
    uint32_t push(const uint64_t* kmers, size_t count) {
        uint32_t n = 0;
        {
            std::lock_guard<std::mutex> g(mu_);
            while (count) {
                // the first symbols of a stream go in a short segment of their own, coded in the plain form
                const bool plain = t_pushed_ < kPlainBelow;
                const size_t take = plain ? std::min<size_t>(count, (kPlainBelow - t_pushed_ + k_ - 1) / k_) : std::min<size_t>(count, seg_kmers_);
                Seg sg; sg.kmers = kmers; sg.n_kmers = (uint32_t)take; sg.plain = plain; sg.t0 = t_pushed_;
                sg.wide = t_pushed_ + take * k_ + 8 >= (1ull << 32);
                segs_.push_back(sg);
                kmers += take * W_; count -= take; t_pushed_ += take * k_; n++;
            }
        }
        cv_work_.notify_all(); cv_spare_.notify_all();
        return n;
    }
    // the next segment, in order; blocks until its records are there
    void take(View& v) {
        std::unique_lock<std::mutex> g(mu_);
        const uint64_t s = next_take_++;
        if (n_spares_ && (int64_t)(next_claim_ - next_take_) < (int64_t)low_water()) cv_spare_.notify_all();        // thin look-ahead: the spares join in
        cv_ready_.wait(g, [&] { return ready_[s % n_bufs_] == s + 1; });
        const Seg& sg = segs_[s - base_];
        v.recs = sg.plain ? nullptr : buf_[s % n_bufs_].get();
        v.wide = sg.wide;
        v.n_syms = (uint64_t)sg.n_kmers * k_; v.kmers = sg.kmers; v.n_kmers = sg.n_kmers; v.t0 = sg.t0;
        for (int i = 0; i < 4; i++) v.c_end[i] = sg.c_end[i];
    }
    void release() {                                           // the segment taken last is done: its buffer may be refilled
        { std::lock_guard<std::mutex> g(mu_); released_++; while (base_ < released_ && !segs_.empty()) { segs_.pop_front(); base_++; } }
        cv_work_.notify_all();
    }
private:
    struct Seg { const uint64_t* kmers; uint32_t n_kmers; bool plain, wide; uint64_t t0; uint64_t c0[4], c_end[4]; };
    // occurrences of the codes 1, 2, 3 in a k-mer's 2-bit groups (the unused high bits are zero: code 0 is the rest)
    static inline void count_codes(uint64_t x, uint64_t* n) {
        const uint64_t lo = x & 0x5555555555555555ull, hi = (x >> 1) & 0x5555555555555555ull;
        n[1] += (uint64_t)__builtin_popcountll(lo & ~hi); n[2] += (uint64_t)__builtin_popcountll(hi & ~lo); n[3] += (uint64_t)__builtin_popcountll(lo & hi);
    }
    uint64_t low_water() const { return n_bufs_ / 4; }          // segments of look-ahead below which the spare helpers work
    void run(bool spare) {
        cpus_.apply();
        for (;;) {
            uint64_t s; Seg sg;
            {
                std::unique_lock<std::mutex> g(mu_);
                auto work = [&] { return next_claim_ < base_ + segs_.size() && next_claim_ < released_ + n_bufs_; };
                if (spare) cv_spare_.wait(g, [&] { return quit_ || (work() && (int64_t)(next_claim_ - next_take_) < (int64_t)low_water()); });
                else cv_work_.wait(g, [&] { return quit_ || work(); });
                if (quit_) return;
                s = next_claim_++;
                sg = segs_[s - base_];
            }
            uint64_t h[4] = {0, 0, 0, 0};                       // the segment's histogram: what the segments after it need first
            for (uint32_t a = 0; a < sg.n_kmers; a++) for (uint32_t w = 0; w < W_; w++) count_codes(sg.kmers[(size_t)a * W_ + w], h);
            h[0] = (uint64_t)sg.n_kmers * k_ - h[1] - h[2] - h[3];
            uint64_t c[4];
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_prefix_.wait(g, [&] { return quit_ || prefix_upto_ == s; });
                if (quit_) return;
                for (int i = 0; i < 4; i++) { c[i] = prefix_c_[i]; prefix_c_[i] += h[i]; }
                prefix_upto_ = s + 1;
                Seg& d = segs_[s - base_];
                for (int i = 0; i < 4; i++) { d.c0[i] = c[i]; d.c_end[i] = prefix_c_[i]; }
            }
            cv_prefix_.notify_all();
            if (!sg.plain) { if (sg.wide) fill(buf_[s % n_bufs_].get(), sg, c); else fill(reinterpret_cast<ChainRec16*>(buf_[s % n_bufs_].get()), sg, c); }
            { std::lock_guard<std::mutex> g(mu_); ready_[s % n_bufs_] = s + 1; }
            cv_ready_.notify_all();
        }
    }
    static inline uint64_t scaled(uint64_t fr, uint64_t d) {   // floor(fr * 2^64 / d), fr < d
#ifdef LEON_HOST_CHAIN_X86
        uint64_t q, r;
        asm("divq %[d]" : "=a"(q), "=d"(r) : "a"(0ull), "d"(fr), [d] "r"(d) : "cc");
        return q;
#else
        return (uint64_t)((((unsigned __int128)fr) << 64) / d);
#endif
    }
    // (the cumulative counts stay in registers and the k-mer is shifted out two bits at a time: 3.0 ns per record on the EPYC 9575F
    // against 5.3-6.5 with the counts in an array and a variable shift per base -- profiles/scripts/chain_ab/ab_fill.cpp; what is
    // left is the division.  A double-precision estimate with an exact fix-up instead of it: 2.5 ns under g++, 5.2 under clang.)
    template <typename Rec> void fill(Rec* out, const Seg& sg, uint64_t* c) const {
        uint64_t d = 5 + sg.t0 + 1;                              // total of the symbol AFTER the one being recorded
        uint64_t cum[5] = {0, c[0], c[0] + c[1], c[0] + c[1] + c[2], c[0] + c[1] + c[2] + c[3]};   // cum[j] = counts of the symbols below j
        for (uint32_t a = 0; a < sg.n_kmers; a++) {
            const uint64_t* w = sg.kmers + (size_t)a * W_;
            unsigned __int128 x = W_ == 2 ? (((unsigned __int128)w[1] << 64) | w[0]) << (128 - 2 * k_) : (unsigned __int128)w[0] << (128 - 2 * k_);
            for (uint32_t i = 0; i < k_; i++, d++, out++, x <<= 2) {    // first base in the two highest bits
                const uint32_t sy = (uint32_t)(x >> 126);
                const uint64_t lo = cum[sy], fr = cum[sy + 1] - lo;
                out->c = scaled(fr, d);
                out->lo = (decltype(out->lo))lo; out->fr = (decltype(out->fr))fr;
                cum[1] += sy < 1; cum[2] += sy < 2; cum[3] += sy < 3; cum[4]++;
            }
        }
        c[0] = cum[1]; c[1] = cum[2] - cum[1]; c[2] = cum[3] - cum[2]; c[3] = cum[4] - cum[3];
    }
    const uint32_t k_, W_, seg_kmers_;
    uint32_t n_threads_ = 5;
    std::unique_ptr<ChainRec24[]> buf_[kMaxBufs];          // (holds either record type)
    uint64_t ready_[kMaxBufs] = {};                            // segment index + 1 a buffer currently holds
    uint32_t n_bufs_ = 512, n_spares_ = 0;
    std::condition_variable cv_spare_;
    std::deque<Seg> segs_;                                     // segments pushed and not yet released; segs_[0] is segment base_
    uint64_t base_ = 0, next_claim_ = 0, next_take_ = 0, released_ = 0, prefix_upto_ = 0, t_pushed_ = 0;
    uint64_t prefix_c_[4] = {1, 1, 1, 1};                      // Order0Model::clear: every symbol starts with a count of 1
    std::mutex mu_;
    std::condition_variable cv_work_, cv_prefix_, cv_ready_;
    bool quit_ = false, running_ = false;
    std::vector<std::thread> th_;
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
        low_ = 0; range_ = ~0ull; n_ = 0; buf_.clear(); w_ = 0;
        for (int i = 0; i <= 5; i++) cum_[i] = i;              // Order0Model::clear: _charRanges[i] = i
    }
    // `count` k-mers in the 2-bit code (W = 1 word per k-mer below k = 32, else low word then high word), first base in
    // the highest bits (LargeInt::toString order), in the plain form: a division per symbol, the model kept here
    void encode_kmers(const uint64_t* w, size_t count, uint32_t k) {
        const uint32_t W = k >= 32 ? 2u : 1u;
        for (; count; count--, w += W) encode_kmer_plain(w, k);
    }
    inline void encode_kmer(const uint64_t* w, uint32_t k) { encode_kmers(w, 1, k); }
    inline void encode_kmer_plain(const uint64_t* w, uint32_t k) {
        if (buf_.size() < w_ + 8 * (size_t)k + 16) buf_.resize(buf_.size() * 2 + 8 * (size_t)k + 4096);
        for (uint32_t i = 0; i < k; i++) {
            const uint32_t bit = 2 * (k - 1 - i);
            encode((uint32_t)(w[bit >> 6] >> (bit & 63)) & 3u);
        }
    }
    // one segment of the feed: its records when the helpers produced them, else (the first symbols of a stream; no x86) its k-mers
    void encode_segment(const ChainFeed::View& v, uint32_t k) {
#ifdef LEON_HOST_CHAIN_X86
        if (v.recs) {
            if (v.wide) encode_records(static_cast<const ChainRec24*>(v.recs), v.n_syms, v.t0);
            else encode_records(static_cast<const ChainRec16*>(v.recs), v.n_syms, v.t0);
            for (int i = 0; i < 4; i++) cum_[i + 1] = cum_[i] + v.c_end[i];     // the model as the plain form would have left it
            cum_[5] = cum_[4] + 1;
            return;
        }
#endif
        encode_kmers(v.kmers, v.n_kmers, k);
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
    // n symbols from their records (ChainFeed), the first one being symbol t0 of the stream.  The state carried from symbol
    // to symbol is (q, L): the quotient floor(range / total) the NEXT symbol will use and the normalised low.  Per symbol:
    //     Lu = L + q * lo, Tu = L + q * (lo + fr) (= Lu + the new range), x = Lu ^ Tu;   Phi:Plo = q * C
    // the next quotient is Phi when no byte leaves and Phi8 = (Phi << 8) | (Plo >> 56) when exactly one does (x < TOP) -- both formed,
    // selected by conditional moves.  The product's fraction within q of the next integer (see ChainFeed; rare), two bytes at
    // once or the carry-less coder's range < BOTTOM reset take a division.
    template <typename Rec> void encode_records(const Rec* r, uint64_t n, uint64_t t0) {
        // Room for the worst case up front -- 8 bytes per symbol: the loop below shifts 64 bits out at most -- so that the fast path's
        // unchecked store can never run past the buffer, whatever mix of rare-path symbols (several bytes each) came before it.
        // (It was n + 64 on the argument that a genomic dictionary codes at ~2 bits per symbol: an unstated property of the input.)
        if (buf_.size() < w_ + 8 * n + 64) buf_.resize(buf_.size() * 2 + 8 * n + 4096);
        settle();
        uint8_t* p = buf_.data() + w_;
        uint8_t* p_end = buf_.data() + buf_.size() - 32;
        uint64_t L = low_, R = range_, tot1 = 5 + t0;          // tot1: the total the carried quotient was formed with ...
        uint64_t q = R / tot1;
        const Rec* const e = r + n;
        for (; r < e; r++) {
            tot1++;                                            // ... from here on, the total of the NEXT symbol
            __builtin_prefetch(reinterpret_cast<const char*>(r) + 2048);      // (the records come from other cores' caches or from memory)
            const uint64_t lo = r->lo, lf = lo + r->fr;
            const uint64_t Lu = L + q * lo;
            uint64_t Tu = L + q * lf;
            asm("" : "+r"(Tu));                                // (the new range is Tu - Lu: a subtraction, not a fourth multiply)
            const uint64_t Ru = Tu - Lu, x = Lu ^ Tu;
            // (a second multiply, (q << 8) * C, instead of the double-word shift measured the same on Zen 5 -- 1.73 against 1.75 ns
            // per symbol -- and costs Intel's single multiplier a slot)
            const uint64_t q8 = q << 8;
            const unsigned __int128 P = (unsigned __int128)q * r->c;
            const uint64_t Phi = (uint64_t)(P >> 64), Plo = (uint64_t)P, Phi8 = (Phi << 8) | (Plo >> 56), Plo8 = Plo << 8;
            if (__builtin_expect((x >> 48) == 0 || Ru < kBottom, 0)) {
                L = Lu; R = Ru;
                while ((L ^ (L + R)) < kTop || (R < kBottom && ((R = (0 - L) & (kBottom - 1)), true))) {
                    if (p >= p_end) { const size_t used = (size_t)(p - buf_.data()); buf_.resize(buf_.size() * 2 + 4096); p = buf_.data() + used; p_end = buf_.data() + buf_.size() - 32; }
                    *p++ = (uint8_t)(L >> 56);
                    R <<= 8;
                    L <<= 8;
                }
                q = R / tot1;
                continue;
            }
            uint64_t qn = Phi, Ln = Lu;
            R = Ru;
            *p = (uint8_t)(Lu >> 56);
            // (qn, R, Ln, p) = one byte ? (Phi:Plo << 8, Ru << 8, Lu << 8, p + 1) : unchanged -- as conditional moves (a compiler
            // turns the selects back into the unpredictable branch this loop exists to avoid)
            asm("cmpq %[top], %[x]\n\tcmovbq %[A], %[qn]\n\tcmovbq %[R1], %[R]\n\tcmovbq %[L1], %[Ln]\n\tadcq $0, %[p]"
                : [qn] "+r"(qn), [R] "+r"(R), [Ln] "+r"(Ln), [p] "+r"(p)
                : [x] "r"(x), [top] "r"(kTop), [A] "r"(Phi8), [R1] "r"(Ru << 8), [L1] "r"(Lu << 8)
                : "cc");
            // in doubt (conservatively, for either outcome): the fraction of q * C / 2^64 (of q * C / 2^56) is within q of 1
            if (__builtin_expect((Plo + q < q) | (Plo8 + q8 < q8), 0)) qn = R / tot1;
            q = qn; L = Ln;
        }
        w_ = (size_t)(p - buf_.data());
        low_ = L; range_ = R; n_ += n;
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
};

// The inverse of AnchorDictCoder (Leon::decodeAnchorDict / RangeDecoder on _anchorDictModel(5) [RECALLED]): one serial
// chain again, on a host core.  k symbols per anchor, first base in the highest bits.  Returns false on a corrupt stream.
// Here the symbol is not known ahead, so nothing but the reciprocal of the total (5 + t, whatever the data) can be prepared by
// helpers; but a step has only four outcomes.  With r = floor(range / total) carried from the step before:
//     p_j = r * cum[j] (five products); the symbol is the number of j with p_j <= code - low (no second division);
//     its new range is one of R_s = p_(s+1) - p_s, and the NEXT quotient floor(R_s / total') is formed for ALL FOUR beside the
//     comparisons (multiply-high by the next total's scaled reciprocal: h >> 8 when no byte leaves, h when one does),
// so that once the symbol is known only selections remain: symbol -> (low, range, h), then the renormalisation test -> (r, low,
// code).  The carried path is multiply, compare, select, add, xor, compare, select (~12 cycles); computing the quotient after
// the symbol (multiply-high, shift, multiply, compare, four selects, subtract, add, xor, compare, select) was ~19.
inline bool decode_anchor_dict(const uint8_t* p, uint64_t n, uint64_t n_anchors, uint32_t k, uint64_t* out) {
    constexpr uint64_t kTop = 1ull << 56, kBottom = 1ull << 48;
    const uint32_t W = k >= 32 ? 2u : 1u;
    uint64_t low = 0, range = ~0ull, code = 0, i = 0, t = 0, cum[6] = {0, 1, 2, 3, 4, 5};
    for (int b = 0; b < 8; b++) code = (code << 8) | (i < n ? p[i] : 0), i++;
    ReciprocalStream recips;
    const bool use_inv = n_anchors * (uint64_t)k > (1u << 16);     // (not worth three threads for a handful of symbols)
    if (use_inv) recips.restart();
    const uint64_t* inv = nullptr;
    uint64_t inv_chunk = ~0ull;
    auto recip_of_symbol = [&](uint64_t tt) -> uint64_t {            // floor((2^72 - 1) / (5 + tt)); symbols are asked for in increasing order
        const uint64_t c = tt / ReciprocalStream::kChunk;
        if (c != inv_chunk) { inv = recips.take(c); inv_chunk = c; }
        return inv[tt % ReciprocalStream::kChunk];
    };
    auto next_byte = [&]() -> uint64_t { const uint64_t v = i < n ? p[i] : 0; i++; return v; };
    uint64_t r = 0;                                             // floor(range / total) of the symbol about to be decoded, when r_valid
    bool r_valid = false;
    for (uint64_t a = 0; a < n_anchors; a++) {
        unsigned __int128 km = 0;
        for (uint32_t j = 0; j < k; j++, t++) {
            const uint64_t tot = 5 + t;
            if (!r_valid) r = range / tot;
            if (r == 0) return false;
            // the symbol is the number of cumulative counts c with c * r <= code - low; the products also are the new low and
            // the candidates of the new range
            const uint64_t d = code - low;
            const uint64_t p1 = r * cum[1], p2 = r * cum[2], p3 = r * cum[3], p4 = r * cum[4];
            const bool g1 = d >= p1, g2 = d >= p2, g3 = d >= p3, g4 = d >= p4;
            if (g4) return false;                              // an N inside an anchor: not a stream this coder wrote
            const uint64_t R0 = p1, R1 = p2 - p1, R2 = p3 - p2, R3 = p4 - p3;
            const uint32_t c = (uint32_t)g1 + (uint32_t)g2 + (uint32_t)g3;
            uint64_t lo, R, h = 0;
            const bool fast = use_inv && t >= 256;
            if (fast) {
                const uint64_t m1 = recip_of_symbol(t + 1);
                uint64_t h0 = (uint64_t)(((unsigned __int128)R0 * m1) >> 64), h1 = (uint64_t)(((unsigned __int128)R1 * m1) >> 64);
                uint64_t h2 = (uint64_t)(((unsigned __int128)R2 * m1) >> 64), h3 = (uint64_t)(((unsigned __int128)R3 * m1) >> 64);
                // (clang selects the range first and multiplies once; forcing the four products apart, as written, is slower: 5.4 against
                //  4.7 ns per symbol on the EPYC 9575F -- the step is bound by its ~45 instructions, not by this dependency)
                // selected by the symbol, as a tree of conditional moves: the symbol is as good as random, a branch on it would be
                // mispredicted three times out of four
                const uint64_t lo_a = g1 ? p1 : 0, R_a = g1 ? R1 : R0, h_a = g1 ? h1 : h0;      // symbol 0 or 1
                const uint64_t lo_b = g3 ? p3 : p2, R_b = g3 ? R3 : R2, h_b = g3 ? h3 : h2;     // symbol 2 or 3
                lo = g2 ? lo_b : lo_a; R = g2 ? R_b : R_a; h = g2 ? h_b : h_a;
            } else {
                lo = g3 ? p3 : g2 ? p2 : g1 ? p1 : 0;
                R = g3 ? R3 : g2 ? R2 : g1 ? R1 : R0;
            }
            low += lo;
            range = R;
            // renormalisation: nothing or one byte in the usual case, both formed and selected without a branch (a byte leaves
            // after every fourth symbol or so: as a branch that is a misprediction per byte); the rare cases take the loop
            const uint64_t x = low ^ (low + range);
            if (__builtin_expect(!fast || (x >> 48) == 0 || range < kBottom, 0)) {
                while ((low ^ (low + range)) < kTop || (range < kBottom && ((range = (0 - low) & (kBottom - 1)), true))) {
                    code = (code << 8) | next_byte();
                    range <<= 8;
                    low <<= 8;
                    if (i > n + 16) return false;                  // far past the end: not a stream this coder wrote (a range of 0 would spin here)
                }
                r_valid = false;
            } else {
                // x < TOP: the top byte is settled, exactly one byte leaves (x >= 2^48 here)
                const uint64_t byte = i < n ? p[i] : 0;
                const uint64_t tot1 = tot + 1;
                uint64_t rn = h >> 8;
#ifdef LEON_HOST_CHAIN_X86
                // (rn, range, low, code, i) = one byte ? (h, range << 8, low << 8, code << 8 | byte, i + 1) : unchanged -- as conditional
                // moves (the compiler makes a branch of the selects, mispredicted at every byte: 4.7 against 5.5 ns per symbol for nothing)
                asm("cmpq %[top], %[x]\n\tcmovbq %[h], %[rn]\n\tcmovbq %[R8], %[rg]\n\tcmovbq %[L8], %[lw]\n\tcmovbq %[C8], %[cd]\n\tadcq $0, %[i]"
                    : [rn] "+r"(rn), [rg] "+r"(range), [lw] "+r"(low), [cd] "+r"(code), [i] "+r"(i)
                    : [x] "r"(x), [top] "r"(kTop), [h] "r"(h), [R8] "r"(range << 8), [L8] "r"(low << 8), [C8] "r"((code << 8) | byte)
                    : "cc");
#else
                const bool one = x < kTop;
                rn = one ? h : rn; range = one ? range << 8 : range; low = one ? low << 8 : low; code = one ? (code << 8) | byte : code; i += one;
#endif
                if (__builtin_expect(range - rn * tot1 >= tot1, 0)) rn++;          // the scaled reciprocal's quotient is one short with probability < 2^-8
                r = rn; r_valid = true;
            }
            cum[1] += (uint64_t)(c < 1); cum[2] += (uint64_t)(c < 2); cum[3] += (uint64_t)(c < 3); cum[4] += 1; cum[5] += 1;   // Order0Model::update, branch-free
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
    explicit AnchorDictWorker(uint32_t k) : k_(k), feed_(k) {}
    ~AnchorDictWorker() { stop(); feed_.stop(); }
    void push(std::vector<uint64_t>&& kmers) {
        const uint32_t W = k_ >= 32 ? 2u : 1u;
        {
            std::lock_guard<std::mutex> g(mu_);
            if (!running_) { running_ = true; quit_ = false; feed_.cpus_.pick(); feed_.start(); th_ = std::thread([this] { feed_.cpus_.apply(); run(); }); }   // (threads only for contexts that code a dictionary)
            q_.emplace_back(std::move(kmers), 0u);
            // the helpers start on the batch's records at once (the vector's storage does not move with the deque's entry)
            q_.back().second = feed_.push(q_.back().first.data(), q_.back().first.size() / W);
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
    void reset() { drain(); coder_.clear(); feed_.reset_stream(); busy_ms_ = 0; starved_ms_ = 0; }
    double busy_ms() const { return busy_ms_; }              // time spent coding since the last reset (read after drain())
    double starved_ms() const { return starved_ms_; }        // ... of which the chain waited for its helpers' records
    AnchorDictCoder& coder() { return coder_; }                // only after drain()
private:
    void run() {
        for (;;) {
            std::pair<std::vector<uint64_t>, uint32_t> batch;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return quit_ || !q_.empty(); });
                if (q_.empty()) return;
                batch = std::move(q_.front());
                q_.pop_front();
            }
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t sgm = 0; sgm < batch.second; sgm++) {
                ChainFeed::View v;
                const auto tw = std::chrono::steady_clock::now();
                feed_.take(v);
                starved_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count();
                coder_.encode_segment(v, k_);
                feed_.release();
            }
            busy_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            {
                std::lock_guard<std::mutex> g(mu_);
                pending_--;
            }
            cv_done_.notify_all();
        }
    }
    uint32_t k_;
    ChainFeed feed_;
    AnchorDictCoder coder_;
    std::mutex mu_;
    std::condition_variable cv_, cv_done_;
    std::deque<std::pair<std::vector<uint64_t>, uint32_t>> q_;
    uint64_t pending_ = 0;
    double busy_ms_ = 0, starved_ms_ = 0;
    bool running_ = false, quit_ = false;
    std::thread th_;
};

}  // namespace leon

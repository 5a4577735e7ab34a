// decode_kernels.hip -- DnaDecoder::execute per read block (gatb DnaCoder.cpp DnaDecoder, RangeCoder.cpp RangeDecoder
// [RECALLED]): the inverse of the encode path.  Read blocks are independent (models and
// previous values reset per block), inside a block everything is one serial chain (every decoded symbol decides what is
// read next, every extension step needs the k-mer the previous step produced), so the mapping is ONE WAVE PER BLOCK:
// the wave-uniform chain runs on all lanes at once, and the lanes are used where the work is wide -- the search of a
// symbol in the model's cumulative counts (two ballots over the two-level table of rc_model.h), the model update,
// the refill of the payload window (256 bytes, 4 per lane).  2 000 blocks keep 2 000 waves in flight, which is what
// hides the memory latency of the bloom probes (one dependent probe set per decoded base).
#include "kernels.h"
#include "rc_model.h"
#include <cstdlib>

namespace leon {

constexpr uint32_t DC_NSLOT = 14;                          // numeric models in LDS (17 KB per block: every block resident)
constexpr uint32_t DC_LIST_CAP = 8192;                     // N / error positions of ONE read in the block's own scratch; longer
                                                           // lists (a 70 kb read full of errors) come from a shared bump pool

constexpr uint32_t DC_LIST_LDS = 64;                       // the first entries of both lists are mirrored in LDS (a cursor step is then no memory round trip)

size_t decode_scratch_bytes(uint64_t n_blocks) {
    return (size_t)n_blocks * ((RC_NNUM - DC_NSLOT) * RC_STRIDE + 2 * DC_LIST_CAP) * sizeof(uint32_t);
}

namespace {

typedef __attribute__((address_space(3))) uint32_t lds_u32;

struct Dec {                                               // wave-uniform decoder state
    uint64_t low, range, code;
    const uint8_t* p; uint64_t n, i;                       // payload, its size, next byte
    uint32_t win;                                          // this lane's 4 bytes of the current 256-byte window
    lds_u32* lds;                                          // small models + DC_NSLOT numeric slots (typed as LDS: ds_read, not flat loads)
    uint8_t* slotmap;
    uint32_t* gmodels;                                     // overflow numeric models (global)
    uint32_t nused;
    uint32_t lane;
    bool bad;                                              // ran far past the payload's end: not a stream the encoder wrote
};

__device__ inline void win_load(Dec& d) {                   // bytes [i & ~255, +256): the buffer is padded past n
    const uint64_t base = d.i & ~255ull;
    uint32_t w; __builtin_memcpy(&w, d.p + base + 4 * d.lane, 4);
    d.win = w;
}
__device__ inline uint32_t next_byte(Dec& d) {
    uint32_t b = 0;
    if (d.i < d.n) {
        const uint32_t o = (uint32_t)(d.i & 255);
        b = ((uint32_t)__builtin_amdgcn_readlane((int)d.win, (int)(o >> 2)) >> (8 * (o & 3))) & 0xFFu;
    }
    d.i++;
    if ((d.i & 255) == 0 && d.i < d.n) win_load(d);
    return b;
}

// table access: LDS (plain, in order within the wave) or the global overflow area (through L2: other lanes' updates must be seen)
__device__ inline uint32_t tld(const uint32_t* t, uint32_t idx) { return __hip_atomic_load(t + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint32_t tld(const lds_u32* t, uint32_t idx) { return ((const volatile lds_u32*)t)[idx]; }
__device__ inline void tinc(uint32_t* t, uint32_t idx) { (void)__hip_atomic_fetch_add(t + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void tinc(lds_u32* t, uint32_t idx) { (void)__hip_atomic_fetch_add(t + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // ds_add_u32

// floor(x / d) for d < 2^31: two rounds of a double-precision estimate (the generic 64-bit division is ~150 instructions
// on this chain), then an exact fix-up.  The reciprocal is the correctly rounded one (v_rcp_f64 alone is good to ~2^-26
// only: an estimate ABOVE the quotient would wrap the remainder); the fix-up loop is bounded whatever the estimates were.
__device__ inline uint64_t div_u64_u32(uint64_t x, uint32_t dv) {
    const double inv = 1.0 / (double)dv;
    uint64_t q = (uint64_t)((double)x * inv * 0.99999999999);          // never above the quotient (x < 2^64: q fits)
    uint64_t rem = x - q * dv;                                          // < ~2^13 * dv + ...: a second, now exact-ish, round
    const uint64_t q2 = (uint64_t)((double)rem * inv * 0.99999999999);
    q += q2; rem -= q2 * dv;
    for (int i = 0; i < 4 && rem >= dv; i++) { q++; rem -= dv; }        // a step or two
    if (rem >= dv) q = x / dv;                                          // (never taken; keeps the chain finite by construction)
    return q;
}

// RangeDecoder::nextByte on one model: the symbol, with the model updated (Order0Model::update)
template <bool GLB, typename PT> __device__ inline uint32_t decode_on(Dec& d, PT T, bool small, uint32_t size) {
    const uint32_t lane = d.lane;
    const uint32_t tot = small ? tld(T, RC_LW + size) : tld(T, 16);
    const uint64_t r = div_u64_u32(d.range, tot);
    // RangeDecoder: value = (code - low) / r, symbol = the last c with F(c) <= value.  Equivalently the last c with
    // F(c) * r <= code - low: every lane multiplies its own cumulative count, no second division (F(c) * r <= range).
    const uint64_t dist = d.code - d.low;
    // level 1: the 16-block, F(16 j) = H[j]
    uint32_t j = 0, base = 0;
    if (!small) {
        const uint32_t h = lane < 16 ? tld(T, lane) : 0xFFFFFFFFu;
        const unsigned long long m1 = __ballot(lane < 16 && (uint64_t)h * r <= dist);
        j = m1 ? (uint32_t)__popcll(m1) - 1 : 0;             // H[0] = 0 passes
        base = (uint32_t)__builtin_amdgcn_readlane((int)h, (int)j);
    }
    // level 2: inside the block, F(16 j + l) = H[j] + Lw[16 j + l]; a small model's Lw[size] is its total
    const uint32_t lim = small ? size : 16u;
    const uint32_t f = lane <= lim && (small || lane < 16) ? base + tld(T, RC_LW + 16 * j + lane) : 0xFFFFFFFFu;
    const unsigned long long m2 = __ballot(lane < lim && (uint64_t)f * r <= dist);
    const uint32_t l = m2 ? (uint32_t)__popcll(m2) - 1 : 0;
    const uint32_t c = 16 * j + l;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)f, (int)l);
    uint32_t hi;
    if (small || l < 15) hi = (uint32_t)__builtin_amdgcn_readlane((int)f, (int)(l + 1));
    else hi = tld(T, j + 1);                            // next block's start (H[16] = the total)
    d.low += (uint64_t)lo * r;
    d.range = r * (uint64_t)(hi - lo);
    while ((d.low ^ (d.low + d.range)) < (1ull << 56) || (d.range < RC_BOTTOM && ((d.range = (0 - d.low) & (RC_BOTTOM - 1)), true))) {
        d.code = (d.code << 8) | next_byte(d);
        d.range <<= 8;
        d.low <<= 8;
        if (d.i > d.n + 16) { d.bad = true; break; }          // (a crafted payload can drive range to 0, which would spin here for ever)
    }
    // Order0Model::update: F(x) += 1 for x > c
    if (lane > l && lane <= (small ? size : 15u)) tinc(T, RC_LW + 16 * j + lane);
    if (!small && lane >= 16 && lane <= 32 && lane - 16 > j) tinc(T, lane - 16);
    if (GLB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    return c;
}

__device__ inline uint32_t decode_sym(Dec& d, uint32_t m) {   // a numeric model (m >= N_SMALL_MODELS)
    uint32_t slot = d.slotmap[m - N_SMALL_MODELS];
    if (slot == 255) {                                       // first use in this block: Order0Model::clear
        slot = d.nused++;
        if (d.lane == 0) d.slotmap[m - N_SMALL_MODELS] = (uint8_t)slot;
        if (slot < DC_NSLOT) model_init(d.lds + RC_SMALL_WORDS + slot * RC_STRIDE, d.lane, false);
        else {
            uint32_t* g = d.gmodels + (uint64_t)(slot - DC_NSLOT) * RC_STRIDE;
            for (uint32_t x = d.lane; x <= 256; x += 64) __hip_atomic_store(g + RC_LW + x, x < 256 ? (x & 15u) : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d.lane < 17) __hip_atomic_store(g + d.lane, 16 * d.lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (slot < DC_NSLOT) return decode_on<false>(d, d.lds + RC_SMALL_WORDS + slot * RC_STRIDE, false, 256);
    return decode_on<true>(d, d.gmodels + (uint64_t)(slot - DC_NSLOT) * RC_STRIDE, false, 256);
}
__device__ inline uint32_t decode_small(Dec& d, uint32_t m) {   // m < N_SMALL_MODELS
    return decode_on<false>(d, d.lds + m * RC_SSTRIDE, true, small_model_size(m));
}
// CompressionUtils::decodeNumeric: the byte count, then the bytes, low first (one decode site: see the kernel's note on its size)
__device__ inline uint64_t decode_numeric(Dec& d, uint32_t group) {
    uint32_t bc = 0;
    uint64_t v = 0;
#pragma unroll 1
    for (uint32_t i = 0; i <= bc; i++) {
        const uint32_t c = decode_sym(d, numeric_model_id(group, i));
        if (i == 0) bc = c > 8 ? 8 : c; else v |= (uint64_t)c << (8 * (i - 1));
    }
    return v;
}
__device__ inline uint64_t from_delta(uint32_t type, uint64_t prev, uint64_t delta) {
    return type == 0 ? delta : (type == 1 ? prev + delta : prev - delta);
}
// bounds on payload-derived values, written so that nothing wraps: w <= wcap always holds
__host__ __device__ inline bool len_fits(uint64_t len, uint64_t w, uint64_t wcap) { return len < (1ull << 31) && len <= wcap - w; }
__host__ __device__ inline bool anchor_fits(uint64_t apos, uint64_t len, uint32_t k) { return len >= k && apos <= len - k; }
__device__ inline uint8_t bin2nt(uint32_t c) { return (uint8_t)("ACTGN"[c % 5]); }

}  // namespace

// ---- the path cache -------------------------------------------------------------------------------------------------
// What the walk learns from the bloom is a pure function of the k-mer: "from k-mer X, the next m steps each have exactly
// one solid successor, and they are b1..bm".  Every genome locus is walked by every read that covers it (30x), by
// thousands of blocks at once, so the waves share that knowledge through a hash table in HBM (this GPU has 288 GB; the
// table is a few bytes per solid k-mer): a walker that finds its k-mer takes up to 28 positions from ONE 64-byte sector
// instead of 7 probes per position and a memory round trip per three positions.  K-mers are stored ORIENTED -- the k-mer as
// it is extended to the right; a left walk over the read is a right walk over the reverse complement -- so both walks of
// both strands meet in the same entries.  A slot's key is written once (compare-and-swap from EMPTY); its payload
// (count << 56 | path) only grows by atomic max -- two paths from the same k-mer are prefixes of one another -- so a reader
// either sees a complete, correct entry or none: the table changes how fast a base is decoded, never which base.
constexpr uint32_t PC_MAX = 28;                            // bases per entry (56 bits) ...
constexpr uint32_t PC_REG = 30;                            // ... the walker remembers the last 30 (a round decides up to 3 positions)
constexpr uint64_t PC_M60 = (1ull << 60) - 1, PC_M56 = (1ull << 56) - 1;
constexpr uint64_t PC_EMPTY = ~0ull;
template <typename K> struct PCL;
template <> struct PCL<uint64_t> { static constexpr uint32_t SLOTS = 4, WORDS = 2, PAY = 1; };   // {key, payload}
template <> struct PCL<u128> { static constexpr uint32_t SLOTS = 2, WORDS = 4, PAY = 2; };       // {lo, hi, payload, -}

__device__ inline uint64_t pc_ld(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void pc_st(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint64_t pc_cas(uint64_t* p, uint64_t expect, uint64_t v) {
    (void)__hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return expect;
}
__device__ inline void pc_max(uint64_t* p, uint64_t v) { (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint64_t readlane64(uint64_t v, uint32_t l) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
}

// An insertion in flight, one per lane.  Its memory operations ride on the walk's own round trips: the compare-and-swap is
// issued next to a round's probes (pc_issue) and looked at when the probes are back (pc_retire), so publishing costs the
// chain instructions, not latency.  Whatever goes wrong (bucket full, a racing writer) only drops the entry.
template <typename K> struct Pend { uint32_t st = 0, tr = 0; K key = 0; uint64_t pay = 0, ret = 0, bucket = 0; };   // st: 0 idle, 1 to claim, 3 to verify

template <typename K> __device__ inline uint64_t* pc_slot(const PathCache& C, uint64_t bucket, uint32_t i) {
    return C.slots + (bucket * PCL<K>::SLOTS + i) * PCL<K>::WORDS;
}
template <typename K> __device__ inline uint32_t pc_issue(const PathCache& C, Pend<K>& P) {
    const uint32_t st = P.st;
    if (st == 1) P.ret = pc_cas(pc_slot<K>(C, P.bucket, P.tr), PC_EMPTY, (uint64_t)P.key);
    else if (KT<K>::W == 2 && st == 3) P.ret = pc_ld(pc_slot<K>(C, P.bucket, P.tr) + 1);
    return st;
}
template <typename K> __device__ inline void pc_retire(const PathCache& C, Pend<K>& P, uint32_t issued) {
    if (issued == 0) return;
    uint64_t* sp = pc_slot<K>(C, P.bucket, P.tr);
    const uint64_t lo = (uint64_t)P.key, hi = KT<K>::W == 2 ? (uint64_t)(P.key >> (KT<K>::W == 2 ? 64 : 0)) : 0;
    bool next = false;
    if (issued == 1) {
        if (P.ret == PC_EMPTY) { if (KT<K>::W == 2) pc_st(sp + 1, hi); pc_max(sp + PCL<K>::PAY, P.pay); P.st = 0; }     // claimed
        else if (P.ret == lo) { if (KT<K>::W == 1) { pc_max(sp + PCL<K>::PAY, P.pay); P.st = 0; } else P.st = 3; }         // there already (two words: look at the other)
        else next = true;
    } else {
        if (P.ret == hi) { pc_max(sp + PCL<K>::PAY, P.pay); P.st = 0; }
        else if (P.ret == PC_EMPTY) P.st = 0;                                      // its writer is between its two stores: let it be
        else next = true;
    }
    if (next) { P.tr++; P.st = P.tr < PCL<K>::SLOTS ? 1u : 0u; }
}
// Entries for the k-mers a..b steps back (a <= b <= PC_REG), taken by idle lanes.  old:p60 is the oriented sequence walked
// so far (old = the k-mer 30 steps back, p60 = the 30 bases since); the k-mer d steps back is followed by min(d, 28) known bases.
template <typename K> __device__ inline void pc_offer(const PathCache& C, Pend<K>& P, uint32_t lane, uint32_t a, uint32_t b, K old, uint64_t p60, K kmk) {
    if (!C.slots || a > b) return;
    const unsigned long long idle = __ballot(P.st == 0);
    const uint32_t d = a + (uint32_t)__popcll(idle & ((1ull << lane) - 1));
    if (P.st == 0 && d <= b) {
        const uint32_t cnt = d < PC_MAX ? d : PC_MAX;
        const K key = ((old << (2 * (PC_REG - d))) | (K)(p60 >> (2 * d))) & kmk;
        const uint64_t path = (p60 >> (2 * (d - cnt))) & ((1ull << (2 * cnt)) - 1);
        if ((uint64_t)key != PC_EMPTY) {
            P.key = key; P.pay = ((uint64_t)cnt << 56) | path; P.tr = 0; P.st = 1;
            P.bucket = key_hash(key) & C.bucket_mask;
        }
    }
}

// a bucket look-up, split in two so that the answer can be asked for early and read late: every asking lane looks at one slot
template <typename K> __device__ inline uint64_t pc_ask(const PathCache& C, K yq, uint32_t slot, bool ask) {
    uint64_t v = 0;
    if (ask) {
        const uint64_t* sp = pc_slot<K>(C, key_hash(yq) & C.bucket_mask, slot);
        const uint64_t k0 = pc_ld(sp), k1 = KT<K>::W == 2 ? pc_ld(sp + 1) : 0, pv = pc_ld(sp + PCL<K>::PAY);
        if (k0 == (uint64_t)yq && (KT<K>::W == 1 || k1 == (uint64_t)(yq >> (KT<K>::W == 2 ? 64 : 0)))) v = pv;
    }
    return v;
}
__device__ inline uint64_t pc_pick(uint64_t plc, bool mine) {         // the payload some lane found (wave-uniform), 0 if none did
    const unsigned long long hit = __ballot(plc != 0 && mine);
    return hit ? readlane64(plc, (uint32_t)__builtin_ctzll(hit)) : 0ull;
}

__global__ void k_pc_init(uint64_t* slots, uint64_t n_words, uint32_t words_per_slot, uint32_t pay) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t w = (uint32_t)(i % words_per_slot);
        slots[i] = w < pay ? PC_EMPTY : 0ull;
    }
}

// err[0]: 0 ok; otherwise 1 + the first failing block in err[1] (code: 1 address/position out of range, 2 output
// overflow, 3 too many N / error positions in one read)
// NH: the bloom's number of hash functions when the instantiation is for it (Leon's 7), 0 = at run time (leon_device.h bloom_keys)
template <typename K, bool DEEP, uint32_t NH>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_decode_blocks(BloomDev B, PathCache PCc, const uint16_t* rv16g, const uint64_t* anchors, uint64_t n_anchors,
                                                     const uint8_t* payloads, const uint64_t* pay_off, const uint32_t* blk_reads,
                                                     const uint64_t* blk_read0, const uint64_t* blk_out0, uint64_t n_blocks,
                                                     uint8_t* out, uint32_t* out_len, uint32_t* scratch, uint32_t* pool,
                                                     unsigned long long* pool_cursor, uint64_t pool_words, int* err, unsigned long long* stats) {
    __shared__ uint16_t rv16[256];
    __shared__ uint32_t models[RC_SMALL_WORDS + DC_NSLOT * RC_STRIDE];
    __shared__ uint8_t slotmap[RC_NNUM];
    __shared__ uint32_t lstN_[DC_LIST_LDS], lstE_[DC_LIST_LDS];  // the first entries of the read's N / error positions (the usual read has a few)
    load_rv16(rv16, rv16g, NH ? B.block_mask : 0xFFFFu);
    volatile lds_u32* const lstN = (volatile lds_u32*)lstN_;
    volatile lds_u32* const lstE = (volatile lds_u32*)lstE_;
    const uint32_t lane = lane_id(), k = B.k;
    const K kmk = kmask<K>(k);
    constexpr uint32_t W = KT<K>::W;
    const bool cache_on = PCc.slots != nullptr;
    for (uint64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        Dec d;
        d.lane = lane; d.lds = (lds_u32*)models; d.slotmap = slotmap; d.nused = 0; d.bad = false;
        uint32_t* blk_scratch = scratch + b * (uint64_t)((RC_NNUM - DC_NSLOT) * RC_STRIDE + 2 * DC_LIST_CAP);
        d.gmodels = blk_scratch;
        uint32_t* const Nblk = blk_scratch + (RC_NNUM - DC_NSLOT) * RC_STRIDE;
        uint32_t* const Eblk = Nblk + DC_LIST_CAP;
        d.p = payloads + pay_off[b]; d.n = pay_off[b + 1] - pay_off[b]; d.i = 0;
        d.low = 0; d.range = ~0ull; d.code = 0;
        __syncthreads();
        for (uint32_t m = 0; m < N_SMALL_MODELS; m++) model_init(&models[m * RC_SSTRIDE], lane, true);   // AbstractDnaCoder::startBlock
        for (uint32_t i = lane; i < RC_NNUM; i += 64) slotmap[i] = 255;
        __syncthreads();
        win_load(d);
        for (int i = 0; i < 8; i++) d.code = (d.code << 8) | next_byte(d);
        uint64_t prevLen = 0, prevPos = 0, prevAddr = 0;
        uint64_t w = blk_out0[b];
        const uint64_t wcap = blk_out0[b + 1];
        const uint64_t r0 = blk_read0[b];
        int fail = 0;
        Pend<K> P;
        uint32_t n_fast = 0, n_miss = 0, n_slow = 0, n_slowjump = 0, n_jumped = 0, n_hybrid = 0, n_childhit = 0, n_short = 0, n_unclean = 0;
        unsigned long long t_head = 0, t_walk = 0, t_mark = stats ? wall_clock64() : 0;   // rounds by kind (stats != nullptr: measurement)
        auto pool_alloc = [&](uint64_t cnt) -> uint32_t* {           // wave-uniform; never freed (rare, bounded by pool_words)
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(pool_cursor, (unsigned long long)cnt);
            base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
            return base + cnt <= pool_words ? pool + base : nullptr;
        };
        for (uint32_t r = 0; r < blk_reads[b] && !fail; r++) {
            if (d.bad) { fail = 4; break; }
            const uint32_t type = decode_small(d, M_READ_TYPE);
            if (type == 1) {                                  // DnaDecoder::decodeNoAnchorRead
                const uint64_t len = decode_numeric(d, G_NOANCHOR_READSIZE);
                if (!len_fits(len, w, wcap)) { fail = 2; break; }
                for (uint64_t i = 0; i < len && !d.bad; i++) {
                    const uint32_t c = decode_small(d, M_NOANCHOR_READ);
                    if (lane == 0) out[w + i] = bin2nt(c);
                }
                if (lane == 0) out_len[r0 + r] = (uint32_t)len;
                w += len;
                continue;
            }
            // read size, anchor position, anchor address: a delta type and a numeric each (one copy of the code: the
            // kernel's instruction footprint is what a lone wave per SIMD pays for)
            uint64_t len = 0, apos = 0, addr = 0;
#pragma unroll 1
            for (uint32_t f = 0; f < 3; f++) {
                const uint32_t dt = decode_small(d, M_READSIZE_DT + f);
                const uint64_t dv = decode_numeric(d, f == 0 ? G_READSIZE : (f == 1 ? G_ANCHOR_POS : G_ANCHOR_ADDRESS));
                if (f == 0) { len = from_delta(dt, prevLen, dv); prevLen = len; }
                else if (f == 1) { apos = from_delta(dt, prevPos, dv); prevPos = apos; }
                else { addr = from_delta(dt, prevAddr, dv); prevAddr = addr; }
            }
            // values from the payload: compared without sums that could wrap (a crafted delta of type 2 makes them ~2^64)
            if (!len_fits(len, w, wcap)) { fail = 2; break; }
            if (addr >= n_anchors || !anchor_fits(apos, len, k)) { fail = 1; break; }
            K anchor = load_kmer<K>(anchors + addr * W) & kmk;   // (asked for here: it arrives under the next symbol)
            const uint32_t rev = decode_small(d, M_ANCHOR_REVCOMP);
            if (rev) anchor = revcomp(anchor, k);
            // both walks start at the anchor: their first buckets are asked for now and read when the walks begin, after the lists
            uint64_t plc_walk[2] = {0, 0};
            if (cache_on) {
                plc_walk[0] = pc_ask<K>(PCc, revcomp(anchor, k), lane, lane < PCL<K>::SLOTS && apos > 0);
                plc_walk[1] = pc_ask<K>(PCc, anchor, lane, lane < PCL<K>::SLOTS && apos + k < len);
            }
            // N positions, then the positions of the recorded sequencing errors
            uint64_t nN = 0, nErr = 0;
            uint32_t *Npos = Nblk, *Epos = Eblk;
#pragma unroll 1
            for (uint32_t l = 0; l < 2 && !fail; l++) {
                const uint64_t cntv = decode_numeric(d, l == 0 ? G_NUMERIC : G_LEFT_ERROR);
                if (cntv > len) { fail = 3; break; }
                uint32_t* gl = cntv > DC_LIST_CAP ? pool_alloc(cntv) : (l == 0 ? Nblk : Eblk);
                if (!gl) { fail = 3; break; }
                volatile lds_u32* ll = l == 0 ? lstN : lstE;
                uint64_t pv = 0;
                for (uint64_t i = 0; i < cntv; i++) {
                    pv += decode_numeric(d, l == 0 ? G_NPOS : G_ERRPOS);
                    if (lane == 0) { gl[i] = (uint32_t)pv; if (i < DC_LIST_LDS) ll[i] = (uint32_t)pv; }
                }
                if (l == 0) { nN = cntv; Npos = gl; } else { nErr = cntv; Epos = gl; }
            }
            if (fail) break;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();

            if (stats) { const unsigned long long t = wall_clock64(); t_head += t - t_mark; t_mark = t; }
            uint8_t* s = out + w;
            if (lane < k) s[apos + lane] = bin2nt((uint32_t)(uint64_t)(anchor >> (2 * (k - 1 - lane))) & 3u);
            // entry idx of a position list: LDS for the first DC_LIST_LDS, the block's scratch (or the pool) beyond
            auto lget = [&](const uint32_t* gl, const volatile lds_u32* ll, int64_t idx) -> uint32_t {
                return idx < (int64_t)DC_LIST_LDS ? ll[idx] : __hip_atomic_load(gl + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            // DnaDecoder::extendAnchor, left then right.  The position lists are ascending: the left walk consumes them
            // from their ends, the right walk from the first entry beyond the anchor.
#pragma unroll 1
            for (int dir = 0; dir < 2; dir++) {
                K kmer = anchor;
                int64_t pos = dir == 0 ? (int64_t)apos - 1 : (int64_t)(apos + k);
                const int64_t step = dir == 0 ? -1 : 1;
                // cursors: index of the next list entry this walk can meet
                int64_t ni, ei;
                if (dir == 0) {
                    ni = (int64_t)nN - 1; while (ni >= 0 && lget(Npos, lstN, ni) > (uint64_t)(pos < 0 ? 0 : pos)) ni--;
                    if (pos < 0) ni = -1;
                    ei = (int64_t)nErr - 1; while (ei >= 0 && lget(Epos, lstE, ei) > (uint64_t)(pos < 0 ? 0 : pos)) ei--;
                    if (pos < 0) ei = -1;
                } else {
                    ni = 0; while (ni < (int64_t)nN && lget(Npos, lstN, ni) < (uint64_t)pos) ni++;
                    ei = 0; while (ei < (int64_t)nErr && lget(Epos, lstE, ei) < (uint64_t)pos) ei++;
                }
                auto cur = [&](const uint32_t* gl, const volatile lds_u32* ll, int64_t idx, int64_t cnt) -> int64_t {
                    return (idx >= 0 && idx < cnt) ? (int64_t)lget(gl, ll, idx) : -1;
                };
                int64_t nextN = cur(Npos, lstN, ni, (int64_t)nN), nextE = cur(Epos, lstE, ei, (int64_t)nErr);
                // the walked sequence in the walk's own orientation, for the path cache: oldk = the k-mer PC_REG steps
                // back, p60 = the bases since; `run` = how many of the last steps had exactly one solid successor in a
                // row, `fresh` = how many of those this wave probed itself (what it knows and the table may not)
                const K y0 = dir == 1 ? anchor : revcomp(anchor, k);
                K oldk = (K)(y0 >> 60);
                uint64_t p60 = (uint64_t)y0 & PC_M60;
                uint32_t run = 0, fresh = 0;
                // One position: the decoded base goes out, the k-mer moves on by the base the GRAPH follows.
                auto advance = [&](uint32_t res4) -> uint32_t {
                    uint32_t nt_out, nt_seed;
                    bool clean = false;
                    if (pos == nextN) {                       // an N: 'A' in the k-mer, no probe, no symbol
                        ni += step; nextN = cur(Npos, lstN, ni, (int64_t)nN);
                        if (pos == nextE) { ei += step; nextE = cur(Epos, lstE, ei, (int64_t)nErr); }   // (never both, kept in step)
                        nt_out = 4; nt_seed = 0;
                    } else {
                        const uint32_t cnt = (uint32_t)__popc(res4);
                        const uint32_t first = res4 ? (uint32_t)__builtin_ctz(res4) : 0u;
                        const bool is_err = pos == nextE;     // a recorded sequencing error: the true base, the graph's successor
                        clean = cnt == 1;
                        nt_out = first; nt_seed = first;
                        if (is_err || cnt != 1) {
                            const uint32_t m = (!is_err && cnt == 2) ? M_BIFURCATION_BINARY : M_BIFURCATION;
                            const uint32_t c = decode_small(d, m);
                            if (is_err) { ei += step; nextE = cur(Epos, lstE, ei, (int64_t)nErr); nt_out = c; nt_seed = res4 ? first : (c & 3u); }
                            else if (cnt == 2) { nt_out = c == 0 ? first : (uint32_t)__builtin_ctz(res4 & (res4 - 1)); nt_seed = nt_out; }
                            else { nt_out = c; nt_seed = c & 3u; }
                        }
                    }
                    if (lane == 0) s[pos] = bin2nt(nt_out);
                    kmer = dir == 1 ? (((kmer << 2) | (K)nt_seed) & kmk) : ((kmer >> 2) | ((K)nt_seed << (2 * (k - 1))));
                    pos += step;
                    // the same step in the walk's orientation (a base prepended on the left is its complement appended on the right)
                    if (!clean) { if (fresh) pc_offer<K>(PCc, P, lane, 1, run < PC_MAX - 1 ? run : PC_MAX - 1, oldk, p60, kmk); run = 0; fresh = 0; }
                    const uint32_t e = dir == 1 ? nt_seed : (nt_seed ^ 2u);
                    oldk = ((oldk << 2) | (K)((p60 >> 58) & 3u)) & kmk;
                    p60 = ((p60 << 2) | e) & PC_M60;
                    if (clean) { run++; fresh++; }
                    return nt_seed;
                };
                auto succ = [&](K x, uint32_t nt) -> K {
                    return dir == 1 ? (((x << 2) | (K)nt) & kmk) : ((x >> 2) | ((K)nt << (2 * (k - 1))));
                };
                // Rounds.  FAST: the table's bucket for the current k-mer (one sector).  HYBRID, where the graph is expected
                // to branch (an entry ended there, or the table had nothing): the bloom probes of the current k-mer alone,
                // beside the buckets of its four possible successors -- the position is decided and the walk jumps on from
                // the successor in the same round.  SLOW, where the table knows nothing yet: the probes of 1 + 4 (+ 16)
                // k-mers -- lane 0 the current k-mer, lanes 1..4 its four possible successors, DEEP lanes 5..20 the sixteen
                // k-mers two steps ahead (the other lanes repeat lane 0's addresses, which costs no traffic) -- so that
                // when a position is decided, the probe of the k-mer it leads to is already there; what it learns is
                // published.  All memory operations of a round, an insertion's compare-and-swap included, are in flight together.
                enum : uint32_t { FAST = 0, HYBRID = 1, SLOW = 2 };
                uint32_t mode = cache_on ? FAST : SLOW;
                constexpr uint64_t PRE_NONE = ~0ull;
                // the table's answer for the current k-mer when it was asked for ahead of time (0: not there)
                uint64_t pre = cache_on && pos >= 0 && pos < (int64_t)len ? pc_pick(dir == 0 ? plc_walk[0] : plc_walk[1], true) : PRE_NONE;
                while (pos >= 0 && pos < (int64_t)len && !d.bad) {
                    if (pos == nextN) { (void)advance(0); pre = PRE_NONE; continue; }  // nothing to ask the graph
                    const K y = cache_on ? (dir == 1 ? kmer : revcomp(kmer, k)) : (K)0;
                    const bool had_pre = pre != PRE_NONE;
                    uint64_t pl = 0;
                    uint32_t res = 0;
                    if (had_pre) { pl = pre; pre = PRE_NONE; }
                    else {
                        uint64_t plc = 0;                     // per lane: the payload of the slot it looked at, if the key is there
                        uint32_t child = 4;
                        if (cache_on) {
                            if (mode == HYBRID) {             // lanes 4..: successor e = 0..3 of the oriented k-mer, slot by slot
                                child = (lane - 4) / PCL<K>::SLOTS;
                                plc = pc_ask<K>(PCc, ((y << 2) | (K)(child & 3u)) & kmk, (lane - 4) % PCL<K>::SLOTS, lane >= 4 && child < 4);
                            } else plc = pc_ask<K>(PCc, y, lane, lane < PCL<K>::SLOTS);
                        }
                        const uint32_t issued = cache_on ? pc_issue<K>(PCc, P) : 0u;
                        if (mode != FAST) {
                            K km = kmer;
                            if (mode == SLOW) {
                                if (lane >= 1 && lane <= 4) km = succ(kmer, (lane - 1) & 3u);
                                if (DEEP && lane >= 5 && lane <= 20) km = succ(succ(kmer, ((lane - 5) >> 2) & 3u), (lane - 5) & 3u);
                            }
                            res = bloom_contains4<K, NH>(B, rv16, km, revcomp(km, k), dir == 1);
                        }
                        if (cache_on) {
                            if (mode == HYBRID) {
                                n_hybrid++;
                                const uint32_t sd = advance((uint32_t)__builtin_amdgcn_readlane((int)res, 0));
                                const uint32_t e = dir == 1 ? sd : (sd ^ 2u);
                                pre = pc_pick(plc, child == e);
                                if (pre) n_childhit++;
                                if (run == 0) n_unclean++;
                                pc_retire<K>(PCc, P, issued);
                                if (run >= PC_MAX && fresh) pc_offer<K>(PCc, P, lane, PC_MAX, PC_MAX, oldk, p60, kmk);
                                continue;
                            }
                            pl = pc_pick(plc, true);
                            pc_retire<K>(PCc, P, issued);
                        }
                    }
                    const uint64_t remaining = dir == 1 ? (uint64_t)((int64_t)len - pos) : (uint64_t)(pos + 1);
                    uint32_t cnt = (uint32_t)(pl >> 56);
                    const uint64_t path = pl & PC_M56;
                    if (cnt > PC_MAX) cnt = 0;
                    if (cnt != 0 && (mode != SLOW || had_pre || cnt >= 3 || cnt >= remaining)) {
                        if (mode == SLOW && !had_pre) n_slowjump++; else n_fast++;
                        // the table knows the next cnt steps: take as many as the read has, up to its next N
                        if (fresh) pc_offer<K>(PCc, P, lane, 1, run < PC_MAX - 1 ? run : PC_MAX - 1, oldk, p60, kmk);
                        fresh = 0;
                        uint64_t jj = cnt < remaining ? cnt : remaining;
                        if (nextN >= 0) { const uint64_t dN = (uint64_t)(dir == 1 ? nextN - pos : pos - nextN); if (dN < jj) jj = dN; }
                        const uint32_t j = (uint32_t)jj;      // >= 1: pos is not an N position here
                        const uint64_t pj = path >> (2 * (cnt - j));
                        const K yn = ((y << (2 * j)) | (K)pj) & kmk;
                        // a whole entry taken and the read goes on: the next bucket is asked for now and read after this jump's bookkeeping
                        const bool ahead = j == PC_MAX && j < remaining;
                        const uint64_t plc_next = ahead ? pc_ask<K>(PCc, yn, lane, lane < PCL<K>::SLOTS) : 0ull;
                        uint32_t ovr = 0xFFu;                 // a recorded error inside the jump: its base comes from the stream, in walk order
                        while (nextE >= 0 && !d.bad) {
                            const uint64_t tE = (uint64_t)(dir == 1 ? nextE - pos : pos - nextE);
                            if (tE >= j) break;
                            const uint32_t c = decode_small(d, M_BIFURCATION);
                            if (lane == tE) ovr = c;
                            ei += step; nextE = cur(Epos, lstE, ei, (int64_t)nErr);
                        }
                        if (lane < j) {
                            const uint32_t c = (uint32_t)(pj >> (2 * (j - 1 - lane))) & 3u;
                            s[pos + step * (int64_t)lane] = bin2nt(ovr != 0xFFu ? ovr : (dir == 1 ? c : (c ^ 2u)));
                        }
                        kmer = dir == 1 ? yn : revcomp(yn, k);
                        oldk = ((oldk << (2 * j)) | (K)(p60 >> (60 - 2 * j))) & kmk;
                        p60 = ((p60 << (2 * j)) | pj) & PC_M60;
                        run += j; pos += step * (int64_t)j; n_jumped += j;
                        mode = (j == cnt && cnt < PC_MAX) ? HYBRID : FAST;       // the entry ended where the graph stops being a path
                        if (mode == HYBRID) n_short++;
                        if (ahead) pre = pc_pick(plc_next, true);
                        continue;
                    }
                    if (had_pre) { mode = mode == HYBRID ? SLOW : HYBRID; continue; }   // (after a one-k-mer round: the successor is not in the table either -- unknown ground, probe deep)
                    if (mode == FAST) { mode = HYBRID; n_miss++; continue; }
                    n_slow++;
                    uint32_t seed0 = 0, seed1 = 0, done = 0;
#pragma unroll 1
                    for (uint32_t st = 0; st < (DEEP ? 3u : 2u) && pos >= 0 && pos < (int64_t)len; st++) {
                        const uint32_t src = st == 0 ? 0u : (st == 1 ? 1u + seed0 : 5u + 4u * seed0 + seed1);
                        const uint32_t sd = advance((uint32_t)__builtin_amdgcn_readlane((int)res, (int)src));
                        if (st == 0) seed0 = sd; else seed1 = sd;
                        done++;
                    }
                    // k-mers whose 28 following bases became known in this round (none if a step broke the run: run < 3 then)
                    if (run >= PC_MAX) pc_offer<K>(PCc, P, lane, PC_MAX, run < PC_MAX - 1 + done ? run : PC_MAX - 1 + done, oldk, p60, kmk);
                }
                if (fresh) pc_offer<K>(PCc, P, lane, 1, run < PC_MAX - 1 ? run : PC_MAX - 1, oldk, p60, kmk);   // the walk's last steps
            }
            if (lane == 0) for (uint64_t i = 0; i < nN; i++) {                                       // also inside the anchor
                const uint32_t q = i < DC_LIST_LDS ? lstN[i] : __hip_atomic_load(Npos + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (q < len) s[q] = 'N';
            }
            if (lane == 0) out_len[r0 + r] = (uint32_t)len;
            w += len;
            if (stats) { const unsigned long long t = wall_clock64(); t_walk += t - t_mark; t_mark = t; }
        }
        if (!fail && d.bad) fail = 4;
        if (!fail && w != wcap) fail = 2;                    // the block table promised exactly wcap - blk_out0[b] bases
        if (fail && lane == 0) { if (atomicCAS(err, 0, fail) == 0) err[1] = (int)b; }
        if (stats && lane == 0) {
            atomicAdd(stats + 0, n_fast); atomicAdd(stats + 1, n_miss); atomicAdd(stats + 2, n_slow); atomicAdd(stats + 3, n_slowjump);
            atomicAdd(stats + 4, n_jumped); atomicAdd(stats + 5, (unsigned long long)blk_reads[b]); atomicAdd(stats + 6, n_hybrid);
            atomicAdd(stats + 7, t_head); atomicAdd(stats + 8, t_walk);
            atomicAdd(stats + 9, n_childhit); atomicAdd(stats + 10, n_short); atomicAdd(stats + 11, n_unclean);
        }
    }
}

// Before the blocks: what the bloom says around every anchor.  One lane per (anchor, orientation) follows the graph to the
// right for as long as there is exactly one solid successor (at most max_steps) and publishes what a decoding wave would
// have learnt there -- the k-mer 28 steps back with its 28 bases at every step, the last k-mers with the shorter paths that
// lead up to the stop at the end.  Anchors sit every few dozen bases of the genome in both orientations, so the decoding
// waves find most of their ground already known and the probe rounds they are left with are those beyond a branch.
// Massively parallel (25 M lanes at 100 M reads), so its own insertions simply wait for their answers.
template <typename K> __device__ inline void pc_insert_now(const PathCache& C, K key, uint64_t pay) {
    const uint64_t lo = (uint64_t)key, hi = (uint64_t)(key >> (KT<K>::W == 2 ? 64 : 0));
    if (lo == PC_EMPTY) return;
    const uint64_t bucket = key_hash(key) & C.bucket_mask;
    for (uint32_t tr = 0; tr < PCL<K>::SLOTS; tr++) {
        uint64_t* sp = pc_slot<K>(C, bucket, tr);
        const uint64_t old = pc_cas(sp, PC_EMPTY, lo);
        if (old == PC_EMPTY) { if (KT<K>::W == 2) pc_st(sp + 1, hi); pc_max(sp + PCL<K>::PAY, pay); return; }
        if (old == lo) {
            if (KT<K>::W == 1) { pc_max(sp + PCL<K>::PAY, pay); return; }
            const uint64_t h = pc_ld(sp + 1);
            if (h == hi) { pc_max(sp + PCL<K>::PAY, pay); return; }
            if (h == PC_EMPTY) return;                       // its writer is between its two stores
        }
    }
}
template <typename K, uint32_t NH>
__global__ void __launch_bounds__(256) k_pc_prewalk(BloomDev B, PathCache C, const uint16_t* rv16g, const uint64_t* anchors, uint64_t n_anchors, uint32_t max_steps) {
    __shared__ uint16_t rv16[256];
    __shared__ K todo[3][256];                                // k-mers beyond a branch, still to be walked from (per lane)
    load_rv16(rv16, rv16g, NH ? B.block_mask : 0xFFFFu);
    const uint64_t idx = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (idx >= 2 * n_anchors) return;
    const uint32_t k = B.k, t = threadIdx.x;
    const K kmk = kmask<K>(k);
    K y = load_kmer<K>(anchors + (idx >> 1) * KT<K>::W) & kmk;
    if (idx & 1) y = revcomp(y, k);
    uint32_t n_todo = 0, budget = max_steps;
    for (;;) {                                                // one stretch of one-successor steps per turn
        K rc = revcomp(y, k);
        K oldk = (K)(y >> 60);
        uint64_t p60 = (uint64_t)y & PC_M60;
        uint32_t run = 0, res4 = 0;
        while (budget) {
            budget--;
            res4 = bloom_contains4<K, NH>(B, rv16, y, rc, true);
            if (__popc(res4) != 1) break;
            const uint32_t e = (uint32_t)__builtin_ctz(res4);
            y = ((y << 2) | (K)e) & kmk;
            rc = (rc >> 2) | ((K)(e ^ 2u) << (2 * (k - 1)));
            oldk = ((oldk << 2) | (K)((p60 >> 58) & 3u)) & kmk;
            p60 = ((p60 << 2) | e) & PC_M60;
            run++;
            res4 = 0;
            if (run >= PC_MAX) pc_insert_now<K>(C, ((oldk << 4) | (K)(p60 >> 56)) & kmk, ((uint64_t)PC_MAX << 56) | (p60 & PC_M56));
        }
        for (uint32_t d = run < PC_MAX - 1 ? run : PC_MAX - 1; d >= 1; d--)
            pc_insert_now<K>(C, ((oldk << (2 * (PC_REG - d))) | (K)(p60 >> (2 * d))) & kmk, ((uint64_t)d << 56) | (p60 & ((1ull << (2 * d)) - 1)));
        // a branch: the decoder will take ONE of the solid successors, whichever its stream says -- all of them are walked
        // from (a false positive of the bloom ends after a step or two), within the lane's budget of probes
        if (__popc(res4) >= 2 && budget)
            for (uint32_t e = 0; e < 4; e++)
                if (((res4 >> e) & 1u) && n_todo < 3) todo[n_todo++][t] = ((y << 2) | (K)e) & kmk;
        if (!n_todo || !budget) break;
        y = todo[--n_todo][t];
    }
}
void launch_path_cache_prewalk(hipStream_t s, BloomDev B, PathCache C, const uint16_t* rv16, const uint64_t* anchors, uint64_t n_anchors, uint32_t max_steps) {
    if (!C.slots || !n_anchors) return;
    const uint64_t nb = (2 * n_anchors + 255) / 256;
#define PW_LAUNCH(KT, NHV) hipLaunchKernelGGL((k_pc_prewalk<KT, NHV>), dim3((uint32_t)nb), dim3(256), 0, s, B, C, rv16, anchors, n_anchors, max_steps)
    if (B.k >= 32) { if (B.n_hash == 7) PW_LAUNCH(u128, 7); else PW_LAUNCH(u128, 0); }
    else { if (B.n_hash == 7) PW_LAUNCH(uint64_t, 7); else PW_LAUNCH(uint64_t, 0); }
#undef PW_LAUNCH
}

// ---- header blocks: the symbols of the stream, decoded on the device ----------------------------------------------------
// A header block is one serial chain too (HeaderDecoder): which model the next symbol is read with depends on
// the record so far, never on the header TEXT.  So the arithmetic decoding -- the expensive part -- runs here, one wave per
// block, all blocks at once, and hands the symbols on as plain bytes in stream order; the host threads rebuild the text from
// them (host_streams.cpp, the same code that decodes a payload itself, reading bytes instead).
__device__ inline uint32_t hdr_sym(Dec& d, uint32_t m, uint8_t* syms, uint64_t& w, uint64_t cap, bool& over) {
    const uint32_t c = m == HM_TYPE ? decode_on<false>(d, d.lds, true, H_TYPE_COUNT) : decode_sym(d, m);
    if (w < cap) { if (d.lane == 0) syms[w] = (uint8_t)c; } else over = true;
    w++;
    return c;
}
__device__ inline uint64_t hdr_numeric(Dec& d, uint8_t* syms, uint64_t& w, uint64_t cap, bool& over) {
    uint32_t bc = hdr_sym(d, HM_NUMERIC0, syms, w, cap, over);
    if (bc > 8) bc = 8;
    uint64_t v = 0;
#pragma unroll 1
    for (uint32_t i = 0; i < bc; i++) v |= (uint64_t)hdr_sym(d, HM_NUMERIC0 + 1 + i, syms, w, cap, over) << (8 * i);
    return v;
}
__device__ inline uint64_t hdr_count(Dec& d, uint32_t m, uint8_t* syms, uint64_t& w, uint64_t cap, bool& over) {
    const uint64_t x = hdr_sym(d, m, syms, w, cap, over);
    return x < 255 ? x : 255 + hdr_numeric(d, syms, w, cap, over);
}
// err[0]: 0 ok, 1 a block's symbols do not fit its share of `syms` (the caller decodes on the host instead), 2 a payload that
// is not a header stream; err[1]: the block
__global__ void __launch_bounds__(64) k_hdr_decode_symbols(const uint8_t* payloads, const uint64_t* pay_off, const uint32_t* blk_reads, uint64_t n_blocks,
                                                          uint8_t* syms, const uint64_t* sym_begin, unsigned long long* sym_count, int* err) {
    __shared__ uint32_t models[RC_SMALL_WORDS + DC_NSLOT * RC_STRIDE];
    __shared__ uint8_t slotmap[RC_NNUM];
    const uint32_t lane = lane_id();
    for (uint64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        Dec d;
        d.lane = lane; d.lds = (lds_u32*)models; d.slotmap = slotmap; d.nused = 0; d.bad = false; d.gmodels = nullptr;   // 14 models: all in LDS
        d.p = payloads + pay_off[b]; d.n = pay_off[b + 1] - pay_off[b]; d.i = 0;
        d.low = 0; d.range = ~0ull; d.code = 0;
        __syncthreads();
        model_init(&models[0], lane, true);                   // the record type (9 symbols)
        for (uint32_t i = lane; i < RC_NNUM; i += 64) slotmap[i] = 255;
        __syncthreads();
        win_load(d);
        for (int i = 0; i < 8; i++) d.code = (d.code << 8) | next_byte(d);
        uint64_t w = sym_begin[b];
        const uint64_t cap = sym_begin[b + 1];
        bool over = false;
        int fail = 0;
        for (uint32_t r = 0; r < blk_reads[b] && !fail; r++) {
            for (;;) {                                        // one record per turn, at least one symbol each: bounded by the block's share
                if (d.bad) { fail = 2; break; }
                if (over) { fail = 1; break; }
                const uint32_t t = hdr_sym(d, HM_TYPE, syms, w, cap, over);
                if (t == H_END_MATCH) break;
                if (t == H_END) { (void)hdr_count(d, HM_FIELD_INDEX, syms, w, cap, over); break; }
                if (t < H_FIELD_ASCII || t >= H_TYPE_COUNT) { fail = 2; break; }
                (void)hdr_count(d, HM_FIELD_INDEX, syms, w, cap, over);
                if (t == H_FIELD_ASCII) {
                    (void)hdr_count(d, HM_FIELD_COLUMN, syms, w, cap, over);
                    const uint64_t sz = hdr_count(d, HM_MIS_SIZE, syms, w, cap, over);
                    for (uint64_t j = 0; j < sz && !d.bad && !over; j++) (void)hdr_sym(d, HM_ASCII, syms, w, cap, over);
                } else if (t == H_FIELD_DELTA || t == H_FIELD_DELTA_2) (void)hdr_numeric(d, syms, w, cap, over);
                else {
                    if (t != H_FIELD_NUMERIC) (void)hdr_count(d, HM_ZERO, syms, w, cap, over);
                    if (t != H_FIELD_ZERO_ONLY) (void)hdr_numeric(d, syms, w, cap, over);
                    (void)hdr_sym(d, HM_ASCII, syms, w, cap, over);
                }
            }
        }
        if (!fail && d.bad) fail = 2;
        if (!fail && over) fail = 1;
        if (lane == 0) sym_count[b] = fail ? 0ull : (unsigned long long)(w - sym_begin[b]);
        if (fail && lane == 0) { if (atomicCAS(err, 0, fail) == 0) err[1] = (int)b; }
    }
}
void launch_hdr_decode_symbols(hipStream_t s, const uint8_t* payloads, const uint64_t* pay_off, const uint32_t* blk_reads, uint64_t n_blocks,
                               uint8_t* syms, const uint64_t* sym_begin, unsigned long long* sym_count, int* err) {
    if (!n_blocks) return;
    const uint32_t g = (uint32_t)(n_blocks > 256 * 8 ? 256 * 8 : n_blocks);
    hipLaunchKernelGGL(k_hdr_decode_symbols, dim3(g), dim3(64), 0, s, payloads, pay_off, blk_reads, n_blocks, syms, sym_begin, sym_count, err);
}

size_t path_cache_slot_bytes(uint32_t k) { return k >= 32 ? 32 : 16; }
// position-dependent 64-bit sum of the bloom's words (the array is padded past n_bytes, bytes beyond it are never set)
__global__ void k_bloom_fingerprint(const uint8_t* bits, uint64_t n_words, uint64_t* sum) {
    uint64_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t w; __builtin_memcpy(&w, bits + 8 * i, 8);
        acc += mix64(w + 0x9E3779B97F4A7C15ULL * (i + 1));
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd((unsigned long long*)sum, (unsigned long long)acc);
}
void launch_bloom_fingerprint(hipStream_t s, const uint8_t* bits, uint64_t n_bytes, uint64_t* d_sum) {
    hipLaunchKernelGGL(k_bloom_fingerprint, dim3(256 * 8), dim3(256), 0, s, bits, (n_bytes + 7) / 8, d_sum);
}
void launch_path_cache_init(hipStream_t s, PathCache C, uint32_t k) {
    if (!C.slots) return;
    const uint32_t wps = k >= 32 ? 4 : 2, pay = k >= 32 ? 2 : 1;
    const uint64_t n_words = (C.bucket_mask + 1) * 8;       // a bucket is one 64-byte sector
    hipLaunchKernelGGL(k_pc_init, dim3(256 * 32), dim3(256), 0, s, C.slots, n_words, wps, pay);
}

void launch_decode_blocks(hipStream_t s, BloomDev B, PathCache C, const uint16_t* rv16, const uint64_t* anchors, uint64_t n_anchors,
                          const uint8_t* payloads, const uint64_t* pay_off, const uint32_t* blk_reads, const uint64_t* blk_read0,
                          const uint64_t* blk_out0, uint64_t n_blocks, uint8_t* out, uint32_t* out_len, uint32_t* scratch,
                          uint32_t* pool, unsigned long long* pool_cursor, uint64_t pool_words, int* err, unsigned long long* stats) {
    if (!n_blocks) return;
    const uint32_t g = (uint32_t)(n_blocks > 256 * 8 ? 256 * 8 : n_blocks);   // 8 waves per CU: the kernel's registers allow two per SIMD
    static const char* force = getenv("LEON_DC_DEEP");       // measurement override
    const bool deep = force ? force[0] == '1' : true;        // 21 probe sets per round and wave: measured better at 200 and at 2 000 blocks
#define DC_LAUNCH2(KT, D, NHV) hipLaunchKernelGGL((k_decode_blocks<KT, D, NHV>), dim3(g), dim3(64), 0, s, B, C, rv16, anchors, n_anchors, payloads, pay_off, \
                                                  blk_reads, blk_read0, blk_out0, n_blocks, out, out_len, scratch, pool, pool_cursor, pool_words, err, stats)
#define DC_LAUNCH(KT, D) do { if (B.n_hash == 7) DC_LAUNCH2(KT, D, 7); else DC_LAUNCH2(KT, D, 0); } while (0)
    if (B.k >= 32) { if (deep) DC_LAUNCH(u128, true); else DC_LAUNCH(u128, false); }
    else { if (deep) DC_LAUNCH(uint64_t, true); else DC_LAUNCH(uint64_t, false); }
#undef DC_LAUNCH2
#undef DC_LAUNCH
}

}  // namespace leon

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

size_t decode_scratch_bytes(uint64_t n_blocks) {
    return (size_t)n_blocks * ((RC_NNUM - DC_NSLOT) * RC_STRIDE + 2 * DC_LIST_CAP) * sizeof(uint32_t);
}

namespace {

struct Dec {                                               // wave-uniform decoder state
    uint64_t low, range, code;
    const uint8_t* p; uint64_t n, i;                       // payload, its size, next byte
    uint32_t win;                                          // this lane's 4 bytes of the current 256-byte window
    uint32_t* lds;                                         // small models + DC_NSLOT numeric slots
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
template <bool GLB> __device__ inline uint32_t tld(const uint32_t* t, uint32_t idx) {
    if (GLB) return __hip_atomic_load(t + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ((const volatile uint32_t*)t)[idx];
}
template <bool GLB> __device__ inline void tinc(uint32_t* t, uint32_t idx) {
    if (GLB) (void)__hip_atomic_fetch_add(t + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else ((volatile uint32_t*)t)[idx] = ((volatile uint32_t*)t)[idx] + 1;
}

// floor(x / d) for d < 2^31: two rounds of a double-precision estimate (the generic 64-bit division is ~150 instructions
// on this chain), then an exact fix-up
__device__ inline uint64_t div_u64_u32(uint64_t x, uint32_t dv) {
    const double inv = 1.0 / (double)dv;
    uint64_t q = (uint64_t)((double)x * inv * 0.99999999999);          // never above the quotient (x < 2^64: q fits)
    uint64_t rem = x - q * dv;                                          // < ~2^13 * dv + ...: a second, now exact-ish, round
    const uint64_t q2 = (uint64_t)((double)rem * inv * 0.99999999999);
    q += q2; rem -= q2 * dv;
    while (rem >= dv) { q++; rem -= dv; }                               // at most a step or two
    return q;
}

// RangeDecoder::nextByte on one model: the symbol, with the model updated (Order0Model::update)
template <bool GLB> __device__ inline uint32_t decode_on(Dec& d, uint32_t* T, bool small, uint32_t size) {
    const uint32_t lane = d.lane;
    const uint32_t tot = small ? tld<GLB>(T, RC_LW + size) : tld<GLB>(T, 16);
    const uint64_t r = div_u64_u32(d.range, tot);
    // RangeDecoder: value = (code - low) / r, symbol = the last c with F(c) <= value.  Equivalently the last c with
    // F(c) * r <= code - low: every lane multiplies its own cumulative count, no second division (F(c) * r <= range).
    const uint64_t dist = d.code - d.low;
    // level 1: the 16-block, F(16 j) = H[j]
    uint32_t j = 0, base = 0;
    if (!small) {
        const uint32_t h = lane < 16 ? tld<GLB>(T, lane) : 0xFFFFFFFFu;
        const unsigned long long m1 = __ballot(lane < 16 && (uint64_t)h * r <= dist);
        j = m1 ? (uint32_t)__popcll(m1) - 1 : 0;             // H[0] = 0 passes
        base = (uint32_t)__builtin_amdgcn_readlane((int)h, (int)j);
    }
    // level 2: inside the block, F(16 j + l) = H[j] + Lw[16 j + l]; a small model's Lw[size] is its total
    const uint32_t lim = small ? size : 16u;
    const uint32_t f = lane <= lim && (small || lane < 16) ? base + tld<GLB>(T, RC_LW + 16 * j + lane) : 0xFFFFFFFFu;
    const unsigned long long m2 = __ballot(lane < lim && (uint64_t)f * r <= dist);
    const uint32_t l = m2 ? (uint32_t)__popcll(m2) - 1 : 0;
    const uint32_t c = 16 * j + l;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)f, (int)l);
    uint32_t hi;
    if (small || l < 15) hi = (uint32_t)__builtin_amdgcn_readlane((int)f, (int)(l + 1));
    else hi = tld<GLB>(T, j + 1);                            // next block's start (H[16] = the total)
    d.low += (uint64_t)lo * r;
    d.range = r * (uint64_t)(hi - lo);
    while ((d.low ^ (d.low + d.range)) < (1ull << 56) || (d.range < RC_BOTTOM && ((d.range = (0 - d.low) & (RC_BOTTOM - 1)), true))) {
        d.code = (d.code << 8) | next_byte(d);
        d.range <<= 8;
        d.low <<= 8;
        if (d.i > d.n + 16) { d.bad = true; break; }          // (a crafted payload can drive range to 0, which would spin here for ever)
    }
    // Order0Model::update: F(x) += 1 for x > c
    if (lane > l && lane <= (small ? size : 15u)) tinc<GLB>(T, RC_LW + 16 * j + lane);
    if (!small && lane >= 16 && lane <= 32 && lane - 16 > j) tinc<GLB>(T, lane - 16);
    if (GLB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    return c;
}

__device__ inline uint32_t decode_sym(Dec& d, uint32_t m) {
    if (m < N_SMALL_MODELS) return decode_on<false>(d, d.lds + m * RC_SSTRIDE, true, small_model_size(m));
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
// CompressionUtils::decodeNumeric
__device__ inline uint64_t decode_numeric(Dec& d, uint32_t group) {
    uint32_t bc = decode_sym(d, numeric_model_id(group, 0));
    if (bc > 8) bc = 8;
    uint64_t v = 0;
    for (uint32_t i = 0; i < bc; i++) v |= (uint64_t)decode_sym(d, numeric_model_id(group, i + 1)) << (8 * i);
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

// err[0]: 0 ok; otherwise 1 + the first failing block in err[1] (code: 1 address/position out of range, 2 output
// overflow, 3 too many N / error positions in one read)
template <typename K, bool DEEP>
__global__ void __launch_bounds__(64) k_decode_blocks(BloomDev B, const uint16_t* rv16g, const uint64_t* anchors, uint64_t n_anchors,
                                                     const uint8_t* payloads, const uint64_t* pay_off, const uint32_t* blk_reads,
                                                     const uint64_t* blk_read0, const uint64_t* blk_out0, uint64_t n_blocks,
                                                     uint8_t* out, uint32_t* out_len, uint32_t* scratch, uint32_t* pool,
                                                     unsigned long long* pool_cursor, uint64_t pool_words, int* err) {
    __shared__ uint16_t rv16[256];
    __shared__ uint32_t models[RC_SMALL_WORDS + DC_NSLOT * RC_STRIDE];
    __shared__ uint8_t slotmap[RC_NNUM];
    load_rv16(rv16, rv16g);
    const uint32_t lane = lane_id(), k = B.k;
    const K kmk = kmask<K>(k);
    constexpr uint32_t W = KT<K>::W;
    for (uint64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        Dec d;
        d.lane = lane; d.lds = models; d.slotmap = slotmap; d.nused = 0; d.bad = false;
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
        auto pool_alloc = [&](uint64_t cnt) -> uint32_t* {           // wave-uniform; never freed (rare, bounded by pool_words)
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(pool_cursor, (unsigned long long)cnt);
            base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
            return base + cnt <= pool_words ? pool + base : nullptr;
        };
        for (uint32_t r = 0; r < blk_reads[b] && !fail; r++) {
            if (d.bad) { fail = 4; break; }
            const uint32_t type = decode_sym(d, M_READ_TYPE);
            if (type == 1) {                                  // DnaDecoder::decodeNoAnchorRead
                const uint64_t len = decode_numeric(d, G_NOANCHOR_READSIZE);
                if (!len_fits(len, w, wcap)) { fail = 2; break; }
                for (uint64_t i = 0; i < len && !d.bad; i++) {
                    const uint32_t c = decode_sym(d, M_NOANCHOR_READ);
                    if (lane == 0) out[w + i] = bin2nt(c);
                }
                if (lane == 0) out_len[r0 + r] = (uint32_t)len;
                w += len;
                continue;
            }
            uint32_t dt; uint64_t dv;
            dt = decode_sym(d, M_READSIZE_DT); dv = decode_numeric(d, G_READSIZE);
            const uint64_t len = from_delta(dt, prevLen, dv); prevLen = len;
            dt = decode_sym(d, M_ANCHORPOS_DT); dv = decode_numeric(d, G_ANCHOR_POS);
            const uint64_t apos = from_delta(dt, prevPos, dv); prevPos = apos;
            dt = decode_sym(d, M_ANCHORADDR_DT); dv = decode_numeric(d, G_ANCHOR_ADDRESS);
            const uint64_t addr = from_delta(dt, prevAddr, dv); prevAddr = addr;
            const uint32_t rev = decode_sym(d, M_ANCHOR_REVCOMP);
            // values from the payload: compared without sums that could wrap (a crafted delta of type 2 makes them ~2^64)
            if (!len_fits(len, w, wcap)) { fail = 2; break; }
            if (addr >= n_anchors || !anchor_fits(apos, len, k)) { fail = 1; break; }
            const uint64_t nN = decode_numeric(d, G_NUMERIC);
            if (nN > len) { fail = 3; break; }
            uint32_t* Npos = nN > DC_LIST_CAP ? pool_alloc(nN) : Nblk;
            if (!Npos) { fail = 3; break; }
            uint64_t pv = 0;
            for (uint64_t i = 0; i < nN; i++) { pv += decode_numeric(d, G_NPOS); if (lane == 0) Npos[i] = (uint32_t)pv; }
            const uint64_t nErr = decode_numeric(d, G_LEFT_ERROR);
            if (nErr > len) { fail = 3; break; }
            uint32_t* Epos = nErr > DC_LIST_CAP ? pool_alloc(nErr) : Eblk;
            if (!Epos) { fail = 3; break; }
            pv = 0;
            for (uint64_t i = 0; i < nErr; i++) { pv += decode_numeric(d, G_ERRPOS); if (lane == 0) Epos[i] = (uint32_t)pv; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();

            K anchor = load_kmer<K>(anchors + addr * W);
            if (rev) anchor = revcomp(anchor, k);
            uint8_t* s = out + w;
            if (lane < k) s[apos + lane] = bin2nt((uint32_t)(uint64_t)(anchor >> (2 * (k - 1 - lane))) & 3u);
            // DnaDecoder::extendAnchor, left then right.  The position lists are ascending: the left walk consumes them
            // from their ends, the right walk from the first entry beyond the anchor.
            for (int dir = 0; dir < 2; dir++) {
                K kmer = anchor;
                int64_t pos = dir == 0 ? (int64_t)apos - 1 : (int64_t)(apos + k);
                const int64_t step = dir == 0 ? -1 : 1;
                // cursors: index of the next list entry this walk can meet
                int64_t ni, ei;
                if (dir == 0) {
                    ni = (int64_t)nN - 1; while (ni >= 0 && __hip_atomic_load(Npos + ni, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (uint64_t)(pos < 0 ? 0 : pos) ) ni--;
                    if (pos < 0) ni = -1;
                    ei = (int64_t)nErr - 1; while (ei >= 0 && __hip_atomic_load(Epos + ei, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (uint64_t)(pos < 0 ? 0 : pos)) ei--;
                    if (pos < 0) ei = -1;
                } else {
                    ni = 0; while (ni < (int64_t)nN && __hip_atomic_load(Npos + ni, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint64_t)pos) ni++;
                    ei = 0; while (ei < (int64_t)nErr && __hip_atomic_load(Epos + ei, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint64_t)pos) ei++;
                }
                auto cur = [&](const uint32_t* list, int64_t idx, int64_t cnt) -> int64_t {
                    return (idx >= 0 && idx < cnt) ? (int64_t)__hip_atomic_load(list + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
                };
                int64_t nextN = cur(Npos, ni, (int64_t)nN), nextE = cur(Epos, ei, (int64_t)nErr);
                // One position: the decoded base goes out, the k-mer moves on by the base the GRAPH follows.
                auto advance = [&](uint32_t res4) {
                    uint32_t nt_out, nt_seed;
                    if (pos == nextN) {                       // an N: 'A' in the k-mer, no probe, no symbol
                        ni += step; nextN = cur(Npos, ni, (int64_t)nN);
                        if (pos == nextE) { ei += step; nextE = cur(Epos, ei, (int64_t)nErr); }   // (never both, kept in step)
                        nt_out = 4; nt_seed = 0;
                    } else {
                        const uint32_t cnt = (uint32_t)__popc(res4);
                        const uint32_t first = res4 ? (uint32_t)__builtin_ctz(res4) : 0u;
                        if (pos == nextE) {                   // a recorded sequencing error: the true base, the graph's successor
                            ei += step; nextE = cur(Epos, ei, (int64_t)nErr);
                            const uint32_t nt = decode_sym(d, M_BIFURCATION);
                            nt_out = nt; nt_seed = res4 ? first : (nt & 3u);
                        } else if (cnt == 1) { nt_out = first; nt_seed = first; }
                        else if (cnt == 2) {
                            const uint32_t second = (uint32_t)__builtin_ctz(res4 & (res4 - 1));
                            const uint32_t nt = decode_sym(d, M_BIFURCATION_BINARY) == 0 ? first : second;
                            nt_out = nt; nt_seed = nt;
                        } else {
                            const uint32_t nt = decode_sym(d, M_BIFURCATION);
                            nt_out = nt; nt_seed = nt & 3u;
                        }
                    }
                    if (lane == 0) s[pos] = bin2nt(nt_out);
                    kmer = dir == 1 ? (((kmer << 2) | (K)nt_seed) & kmk) : ((kmer >> 2) | ((K)nt_seed << (2 * (k - 1))));
                    pos += step;
                    return nt_seed;
                };
                // Two (DEEP: three) positions per memory round trip: lane 0 probes the current k-mer, lanes 1..4 its four
                // possible successors, DEEP lanes 5..20 the sixteen k-mers two steps ahead (the other lanes repeat lane
                // 0's addresses, which costs no traffic), so when a position is decided the probe of the k-mer it leads
                // to is already there.  DEEP asks for 21 probe sets per round; with 2 000 waves in flight that is ~45 G sectors/s,
                // still under the chip's random-sector rate.
                auto succ = [&](K x, uint32_t nt) -> K {
                    return dir == 1 ? (((x << 2) | (K)nt) & kmk) : ((x >> 2) | ((K)nt << (2 * (k - 1))));
                };
                while (pos >= 0 && pos < (int64_t)len) {
                    K km = kmer;
                    if (lane >= 1 && lane <= 4) km = succ(kmer, (lane - 1) & 3u);
                    if (DEEP && lane >= 5 && lane <= 20) km = succ(succ(kmer, ((lane - 5) >> 2) & 3u), (lane - 5) & 3u);
                    const uint32_t res = bloom_contains4<K>(B, rv16, km, revcomp(km, k), dir == 1);
                    const uint32_t seed0 = advance((uint32_t)__builtin_amdgcn_readlane((int)res, 0));
                    if (pos >= 0 && pos < (int64_t)len) {
                        const uint32_t seed1 = advance((uint32_t)__builtin_amdgcn_readlane((int)res, (int)(1 + seed0)));
                        if (DEEP && pos >= 0 && pos < (int64_t)len)
                            (void)advance((uint32_t)__builtin_amdgcn_readlane((int)res, (int)(5 + 4 * seed0 + seed1)));
                    }
                }
            }
            if (lane == 0) for (uint64_t i = 0; i < nN; i++) {                                       // also inside the anchor
                const uint32_t q = __hip_atomic_load(Npos + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (q < len) s[q] = 'N';
            }
            if (lane == 0) out_len[r0 + r] = (uint32_t)len;
            w += len;
        }
        if (!fail && d.bad) fail = 4;
        if (!fail && w != wcap) fail = 2;                    // the block table promised exactly wcap - blk_out0[b] bases
        if (fail && lane == 0) { if (atomicCAS(err, 0, fail) == 0) err[1] = (int)b; }
    }
}

void launch_decode_blocks(hipStream_t s, BloomDev B, const uint16_t* rv16, const uint64_t* anchors, uint64_t n_anchors,
                          const uint8_t* payloads, const uint64_t* pay_off, const uint32_t* blk_reads, const uint64_t* blk_read0,
                          const uint64_t* blk_out0, uint64_t n_blocks, uint8_t* out, uint32_t* out_len, uint32_t* scratch,
                          uint32_t* pool, unsigned long long* pool_cursor, uint64_t pool_words, int* err) {
    if (!n_blocks) return;
    const uint32_t g = (uint32_t)(n_blocks > 256 * 9 ? 256 * 9 : n_blocks);
    static const char* force = getenv("LEON_DC_DEEP");       // measurement override
    const bool deep = force ? force[0] == '1' : true;        // 21 probe sets per round and wave: measured better at 200 and at 2 000 blocks
#define DC_LAUNCH(KT, D) hipLaunchKernelGGL((k_decode_blocks<KT, D>), dim3(g), dim3(64), 0, s, B, rv16, anchors, n_anchors, payloads, pay_off, \
                                            blk_reads, blk_read0, blk_out0, n_blocks, out, out_len, scratch, pool, pool_cursor, pool_words, err)
    if (B.k >= 32) { if (deep) DC_LAUNCH(u128, true); else DC_LAUNCH(u128, false); }
    else { if (deep) DC_LAUNCH(uint64_t, true); else DC_LAUNCH(uint64_t, false); }
#undef DC_LAUNCH
}

}  // namespace leon

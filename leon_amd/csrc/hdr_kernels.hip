// hdr_kernels.hip -- the header stream's symbols (gatb HeaderCoder.cpp HeaderEncoder [RECALLED lo]).
// Upstream codes headers per read block through the same RangeCoder, field by field against the previous header (the
// file's FIRST header at the start of every block, AbstractHeaderCoder::startBlock).  What a header contributes to the
// stream depends on the previous header's TEXT only, never on coder state, so the records of all headers of a batch
// are produced in parallel -- one lane per header, count pass / scan / emit pass like k_symbols -- and the per-block
// serial part is left to k_rc_encode (rc_kernels.hip) with this stream's model set.  The record layout is the one
// DESIGN.md section 1.3 states; the tests compare payload bytes with the CPU restatement of the same rules.
#include "kernels.h"
#include <algorithm>

namespace leon {

namespace {

struct HField { uint32_t len, tok; uint32_t sep; bool has_sep; uint32_t kind, zeros; uint64_t value; };
enum : uint32_t { HK_ASCII = 0, HK_NUM = 1, HK_ZERO_ONLY = 2, HK_ZERO_NUM = 3 };

__device__ inline bool h_isalnum(uint32_t c) { return (c - '0') < 10u || ((c | 32u) - 'a') < 26u; }

// the field starting at h[pos] (pos < len): an alphanumeric run plus the one separator byte after it, classified in the same pass
__device__ inline HField h_field(const uint8_t* h, uint32_t len, uint32_t pos) {
    HField f; f.sep = 0; f.has_sep = false; f.kind = HK_ASCII; f.zeros = 0; f.value = 0;
    uint32_t e = pos, z = 0, sig = 0;
    bool digits = true, lead = true;
    uint64_t v = 0;
    while (e < len) {
        const uint32_t c = h[e];
        if (!h_isalnum(c)) { f.has_sep = true; f.sep = c; break; }
        const uint32_t d = c - '0';
        if (d < 10u) {
            if (lead && d == 0) z++; else { lead = false; if (sig < 18) v = v * 10 + d; sig++; }
        } else digits = false;
        e++;
    }
    f.tok = e - pos;
    f.len = f.tok + (f.has_sep ? 1u : 0u);
    if (digits && f.tok > 0 && !(f.has_sep && f.sep == 0)) {
        if (f.tok == 1 || z == 0) { if (f.tok <= 18) { f.kind = HK_NUM; f.value = f.tok == 1 ? (uint64_t)(h[pos] - '0') : v; } }
        else if (z == f.tok) { f.kind = HK_ZERO_ONLY; f.zeros = z; }
        else if (sig <= 18) { f.kind = HK_ZERO_NUM; f.zeros = z; f.value = v; }
    }
    return f;
}

struct HSink {
    uint16_t* p;          // nullptr: count only
    uint64_t n;
    __device__ inline void put(uint32_t model, uint32_t sym) { if (p) p[n] = (uint16_t)(model | (sym << 8)); n++; }
    __device__ inline void numeric(uint64_t v) {                    // CompressionUtils::encodeNumeric on _numericModels
        uint32_t bc = 1;
        while (bc < 8 && (v >> (8 * bc)) != 0) bc++;
        put(HM_NUMERIC0, bc);
        for (uint32_t b = 0; b < bc; b++) put(HM_NUMERIC0 + 1 + b, (uint32_t)(v >> (8 * b)) & 0xff);
    }
    __device__ inline void count(uint32_t model, uint64_t x) {
        if (x < 255) put(model, (uint32_t)x); else { put(model, 255); numeric(x - 255); }
    }
};

}  // namespace

// EMIT = false: sym_off[i] = number of symbols of header i; EMIT = true: the symbols at syms + 2 * sym_off[i].
template <bool EMIT>
__global__ void __launch_bounds__(256) k_hdr_symbols(const uint8_t* hdr, const uint64_t* off, uint64_t n, uint32_t rpb,
                                                    const uint8_t* first, uint32_t first_len, uint64_t* sym_off, uint8_t* syms) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* cur = hdr + off[i];
    const uint32_t lc = (uint32_t)(off[i + 1] - off[i]);
    const bool block_start = i % rpb == 0;                          // AbstractHeaderCoder::startBlock: previous = the file's first header
    const uint8_t* prev = block_start ? first : hdr + off[i - 1];
    const uint32_t lp = block_start ? first_len : (uint32_t)(off[i] - off[i - 1]);
    HSink S;
    S.p = EMIT ? (uint16_t*)syms + sym_off[i] : nullptr;
    S.n = 0;
    uint32_t pc = 0, pp = 0, fi = 0, fprev = 0;
    while (pc < lc) {
        const HField c = h_field(cur, lc, pc);
        const bool have_p = pp < lp;
        HField p; p.len = 0; p.kind = HK_ASCII; p.sep = 0; p.has_sep = false; p.value = 0; p.tok = 0; p.zeros = 0;
        uint32_t col = 0;
        bool same = false;
        if (have_p) {
            p = h_field(prev, lp, pp);
            const uint32_t m = c.len < p.len ? c.len : p.len;
            while (col < m && cur[pc + col] == prev[pp + col]) col++;
            same = col == c.len && c.len == p.len;
            fprev++;
        }
        if (!same) {
            if (c.kind == HK_NUM) {
                if (have_p && p.kind == HK_NUM && p.has_sep == c.has_sep && p.sep == c.sep && p.value != c.value) {
                    const bool up = c.value > p.value;
                    S.put(HM_TYPE, up ? H_FIELD_DELTA : H_FIELD_DELTA_2);
                    S.count(HM_FIELD_INDEX, fi);
                    S.numeric(up ? c.value - p.value : p.value - c.value);
                } else {
                    S.put(HM_TYPE, H_FIELD_NUMERIC);
                    S.count(HM_FIELD_INDEX, fi);
                    S.numeric(c.value);
                    S.put(HM_ASCII, c.has_sep ? c.sep : 0u);
                }
            } else if (c.kind != HK_ASCII) {
                S.put(HM_TYPE, c.kind == HK_ZERO_ONLY ? H_FIELD_ZERO_ONLY : H_FIELD_ZERO_AND_NUMERIC);
                S.count(HM_FIELD_INDEX, fi);
                S.count(HM_ZERO, c.zeros);
                if (c.kind == HK_ZERO_NUM) S.numeric(c.value);
                S.put(HM_ASCII, c.has_sep ? c.sep : 0u);
            } else {
                S.put(HM_TYPE, H_FIELD_ASCII);
                S.count(HM_FIELD_INDEX, fi);
                S.count(HM_FIELD_COLUMN, col);
                S.count(HM_MIS_SIZE, c.len - col);
                if (EMIT) { for (uint32_t j = col; j < c.len; j++) S.put(HM_ASCII, cur[pc + j]); }
                else S.n += c.len - col;
            }
        }
        pc += c.len; pp += p.len; fi++;
    }
    while (pp < lp) { pp += h_field(prev, lp, pp).len; fprev++; }
    if (fi >= fprev) S.put(HM_TYPE, H_END_MATCH);
    else { S.put(HM_TYPE, H_END); S.count(HM_FIELD_INDEX, fi); }
    if (!EMIT) sym_off[i] = S.n;
}

void launch_hdr_symbols(hipStream_t s, const uint8_t* hdr, const uint64_t* off, uint64_t n, uint32_t rpb, const uint8_t* first,
                        uint32_t first_len, uint64_t* sym_off, uint8_t* syms) {
    if (!n) return;
    const uint32_t g = (uint32_t)((n + 255) / 256);
    if (syms) hipLaunchKernelGGL(k_hdr_symbols<true>, dim3(g), dim3(256), 0, s, hdr, off, n, rpb, first, first_len, sym_off, syms);
    else hipLaunchKernelGGL(k_hdr_symbols<false>, dim3(g), dim3(256), 0, s, hdr, off, n, rpb, first, first_len, sym_off, syms);
}

}  // namespace leon

// ================================================================================================
// quality stream, lossy form (the default without -lossless): DnaEncoder::storeSolidCoverageInfo + smoothQuals
// [RECALLED med].  cover[i] = number of the read's k-mers that are in the bloom (canonical, N read as 'A') and span
// position i; a quality becomes '@' where cover[i] >= 2 (_smoothing_threshold) or where it is above '@' (truncation);
// reads shorter than k are left as they are.  One wave per read, lane = k-mer position (64 per step): the solid flags
// of a step are one ballot, and a position's coverage is a popcount over the k flags that span it.
// ================================================================================================
namespace leon {

// ---- the smoothing's bloom probes, shared between the reads of a locus ----
// In file order every k-mer of every read costs its random sectors (100 M reads: 1.1 s at the chip's rate; taking the reads in a
// better ORDER alone gives 7 %: a read's sectors are gone from L2 long before its neighbour is looked at).
// The walk shares probes because the reads of an anchor sit in neighbouring lanes and step through the same genome k-mers in the
// same instruction.  The same here, without the dictionary: every read is anchored on its MINIMIZER (its canonical k-mer of
// smallest hash); reads are sorted by it, one lane per read, and every lane goes through its k-mers outwards from the minimizer,
// one step to either side per iteration, the side towards the canonical k-mer's "right" first whatever the read's strand -- so at
// iteration g all reads that share a minimizer ask for the SAME genome k-mer, and identical addresses coalesce in the wave's own
// memory instruction.  The flags (k-mer solid or not, one bit per position) go to an array laid out like the N mask; a second,
// streaming kernel turns them into coverages and rewrites the qualities.  Measured at 100 M x 150 bp: 1 106 -> 580 ms, FETCH_SIZE
// 1.76 TB = 2.3 sectors per k-mer position; what is left is mostly the reads whose minimizer a sequencing error destroyed (a
// quarter of them at 1 % errors, k = 31): they land in clusters of their own, and the k-mers over errors, which no read shares.
template <typename K>
__global__ void __launch_bounds__(256) k_read_minimizer(ReadsDev R, uint32_t* key, uint32_t* mpos) {
    const uint32_t lane = lane_id(), k = R.k;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < R.n; i += nwaves) {
        const uint32_t len = R.len[i];
        if (len < k) { if (lane == 0) { key[i] = 0xFFFFFFFFu; mpos[i] = 0; } continue; }     // reads shorter than k: last, untouched
        const uint32_t nk = len - k + 1;
        const uint32_t* pk = R.packed + 2 * R.slot_off[i];
        uint64_t best = ~0ull;                                   // (hash << 32) | position: the smallest hash, its first position
        for (uint32_t base = 0; base < nk; base += 64) {
            const uint32_t p = base + lane;
            const K cn = canon_from_words<K>(pass_words(pk, base, lane), base, p < nk ? p : nk - 1, k);
            const uint64_t cand = ((uint64_t)(key_hash(cn) >> 33) << 32) | p;
            if (p < nk && cand < best) best = cand;
        }
        for (int d = 32; d; d >>= 1) {
            const uint64_t o = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), d) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)best, d);
            best = o < best ? o : best;
        }
        if (lane == 0) {
            const uint32_t m = (uint32_t)best;
            const K km = kmer_at<K>(pk, m, k);
            const uint32_t fwd = km <= revcomp(km, k) ? 1u : 0u;   // the read carries the minimizer in its canonical orientation
            key[i] = (uint32_t)(best >> 32);
            mpos[i] = (m << 1) | fwd;
        }
    }
}

__global__ void k_mpos_from_anchors(ReadsDev R, const int32_t* anchor_pos, const uint32_t* anchor_addr, const uint8_t* flags, uint32_t* key, uint32_t* mpos) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < R.n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t len = R.len[i], k = R.k;
        const int32_t a = anchor_pos[i];
        if (len < k) { key[i] = 0xFFFFFFFFu; mpos[i] = 0; }
        else if (a < 0 || (uint32_t)a > len - k) { key[i] = 0xFFFFFFFEu; mpos[i] = 1; }          // no (usable) anchor: on its own, from its first k-mer
        else { key[i] = anchor_addr[i]; mpos[i] = ((uint32_t)a << 1) | ((flags[i] & 1u) ? 0u : 1u); }   // flags bit 0: the anchor is reverse-complemented in the read
    }
}
void launch_mpos_from_anchors(hipStream_t s, ReadsDev R, const int32_t* anchor_pos, const uint32_t* anchor_addr, const uint8_t* flags, uint32_t* key, uint32_t* mpos) {
    if (!R.n) return;
    hipLaunchKernelGGL(k_mpos_from_anchors, dim3((uint32_t)std::min<uint64_t>((R.n + 255) / 256, 8192)), dim3(256), 0, s, R, anchor_pos, anchor_addr, flags, key, mpos);
}

template <typename K, uint32_t NH>       // NH: the bloom's number of hash functions when built for it (7), 0 = at run time (leon_device.h bloom_keys)
__global__ void __launch_bounds__(256) k_solid_flags(ReadsDev R, BloomDev B, const uint16_t* rv16g, const uint32_t* perm, const uint32_t* mpos,
                                                    uint32_t* flags) {
    __shared__ uint16_t rv16[256];
    load_rv16(rv16, rv16g, NH ? B.block_mask : 0xFFFFu);
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= R.n) return;
    const uint32_t i = perm[t], k = R.k, len = R.len[i];
    if (len < k) return;
    const uint32_t nk = len - k + 1, m = mpos[i] >> 1;
    const bool fwd = (mpos[i] & 1u) != 0;
    const uint32_t* pk = R.packed + 2 * R.slot_off[i];
    uint32_t* fl = flags + R.slot_off[i];
    const K mask = kmask<K>(k);
    const uint32_t top = 2 * (k - 1);
    // side P goes to higher positions, side M to lower ones; each keeps its k-mer, the reverse complement and a word of the read
    K xP = kmer_at<K>(pk, m, k), rP = revcomp(xP, k), xM = xP, rM = rP;
    uint32_t wP = 0, wP_idx = 0xFFFFFFFFu, wM = 0, wM_idx = 0xFFFFFFFFu;
    // flags collected per 32-position word and side, handed over when the side moves on to another word
    uint32_t aP = 0, aP_idx = m >> 5, aM = 0, aM_idx = m >> 5;
    if (bloom_contains_xr<K, NH>(B, rv16, xP, rP)) aP |= 1u << (m & 31);
    const uint32_t nP = nk - 1 - m, nM = m, nmax = nP > nM ? nP : nM;
    for (uint32_t g = 1; g <= nmax; g++) {
        // "A" = the side towards the canonical minimizer's right (P for a read that carries it forwards, M otherwise), "B" = the other
        const bool onP = g <= nP, onM = g <= nM;
        if (onP) {                                               // position m + g: the new last base is m + g + k - 1
            const uint32_t q = m + g + k - 1;
            if ((q >> 4) != wP_idx) { wP_idx = q >> 4; wP = pk[wP_idx]; }
            const uint32_t b = (wP >> (30 - 2 * (q & 15))) & 3u;
            xP = ((xP << 2) | (K)b) & mask;
            rP = (rP >> 2) | ((K)(b ^ 2u) << top);
        }
        if (onM) {                                               // position m - g: the new first base is m - g
            const uint32_t q = m - g;
            if ((q >> 4) != wM_idx) { wM_idx = q >> 4; wM = pk[wM_idx]; }
            const uint32_t b = (wM >> (30 - 2 * (q & 15))) & 3u;
            xM = (xM >> 2) | ((K)b << top);
            rM = ((rM << 2) | (K)(b ^ 2u)) & mask;
        }
        const bool onA = fwd ? onP : onM, onB = fwd ? onM : onP;
        // (both sides' first hashes in flight together, then the rest of both: measured, no different -- 487 against 471 ms; the
        // kernel is bound by its sectors, 1.76 TB at 100 M reads, not by their latency)
        const bool sA = onA && bloom_contains_xr<K, NH>(B, rv16, fwd ? xP : xM, fwd ? rP : rM);
        const bool sB = onB && bloom_contains_xr<K, NH>(B, rv16, fwd ? xM : xP, fwd ? rM : rP);
        const bool sP = fwd ? sA : sB, sM = fwd ? sB : sA;
        if (onP) {
            const uint32_t p = m + g;
            if ((p >> 5) != aP_idx) { if (aP) atomicOr(fl + aP_idx, aP); aP = 0; aP_idx = p >> 5; }
            if (sP) aP |= 1u << (p & 31);
        }
        if (onM) {
            const uint32_t p = m - g;
            if ((p >> 5) != aM_idx) { if (aM) atomicOr(fl + aM_idx, aM); aM = 0; aM_idx = p >> 5; }
            if (sM) aM |= 1u << (p & 31);
        }
    }
    if (aP) atomicOr(fl + aP_idx, aP);
    if (aM) atomicOr(fl + aM_idx, aM);
}

// coverages from the flags, qualities rewritten: one wave per read, lane = position (DnaEncoder::smoothQuals as restated in DESIGN 1.4)
__global__ void __launch_bounds__(256) k_qual_rewrite(ReadsDev R, const uint32_t* flags, uint8_t* quals) {
    const uint32_t lane = lane_id(), k = R.k;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < R.n; i += nwaves) {
        const uint32_t len = R.len[i];
        if (len < k) continue;
        const uint32_t* fl = flags + R.slot_off[i];
        uint8_t* q = quals + (R.base_off[i] - R.base_off[0]);
        unsigned long long prev = 0;                              // solid flags of k-mer positions [base - 64, base)
        for (uint32_t base = 0; base < len; base += 64) {
            const uint32_t p = base + lane;
            // positions [base, base + 64): two words of the read's flag array (it has a word per 32-base slot; bits past the
            // last k-mer are zero)
            const uint32_t nw = (len + 31) >> 5, w0 = base >> 5;
            const unsigned long long cur = (unsigned long long)(w0 < nw ? fl[w0] : 0u) | ((unsigned long long)(w0 + 1 < nw ? fl[w0 + 1] : 0u) << 32);
            if (p < len) {
                const unsigned __int128 both = ((unsigned __int128)cur << 64) | prev;
                const uint32_t lo = lane + 65 - k;                // k <= 63: lo >= 2
                const unsigned __int128 win = (both >> lo) & ((((unsigned __int128)1) << k) - 1);
                const uint32_t cover = (uint32_t)__popcll((unsigned long long)win) + (uint32_t)__popcll((unsigned long long)(win >> 64));
                const uint8_t c = q[p];
                if (cover >= 2u || c > (uint8_t)'@') q[p] = (uint8_t)'@';
            }
            prev = cur;
        }
    }
}

// the smoothing in file order, every read probing for itself: small batches, and the measurement reference (LEON_QUAL_ORDER=0)
template <typename K>
__global__ void __launch_bounds__(256) k_qual_smooth(ReadsDev R, BloomDev B, const uint16_t* rv16g, uint8_t* quals) {
    __shared__ uint16_t rv16[256];
    load_rv16(rv16, rv16g);
    const uint32_t lane = lane_id(), k = R.k;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < R.n; i += nwaves) {
        const uint32_t len = R.len[i];
        if (len < k) continue;
        const uint32_t nk = len - k + 1;
        const uint32_t* pk = R.packed + 2 * R.slot_off[i];
        uint8_t* q = quals + (R.base_off[i] - R.base_off[0]);
        unsigned long long prev = 0;                              // solid flags of k-mer positions [base - 64, base)
        for (uint32_t base = 0; base < len; base += 64) {
            const uint32_t p = base + lane;
            bool solid = false;
            const uint32_t kb = base < nk ? base : ((nk - 1) & ~63u);    // past the last k-mer only the coverage counts go on
            const K cn = canon_from_words<K>(pass_words(pk, kb, lane), kb, p < nk ? p : nk - 1, k);
            if (p < nk) solid = bloom_contains<K>(B, rv16, cn);
            const unsigned long long cur = __ballot(solid);
            if (p < len) {
                // k-mer positions spanning p: [p - k + 1, p] = bits [lane + 65 - k, lane + 64] of the 128 flags prev:cur
                const unsigned __int128 both = ((unsigned __int128)cur << 64) | prev;
                const uint32_t lo = lane + 65 - k;                // k <= 63: lo >= 2
                const unsigned __int128 win = (both >> lo) & ((((unsigned __int128)1) << k) - 1);
                const uint32_t cover = (uint32_t)__popcll((unsigned long long)win) + (uint32_t)__popcll((unsigned long long)(win >> 64));
                const uint8_t c = q[p];
                if (cover >= 2u || c > (uint8_t)'@') q[p] = (uint8_t)'@';
            }
            prev = cur;
        }
    }
}
static uint32_t wave_grid(uint64_t n) { uint64_t g = (n + 3) / 4; return (uint32_t)(g > 256 * 16 ? 256 * 16 : g); }
void launch_read_minimizer(hipStream_t s, ReadsDev R, uint32_t* key, uint32_t* mpos) {
    if (!R.n) return;
    if (R.k >= 32) hipLaunchKernelGGL(k_read_minimizer<u128>, dim3(wave_grid(R.n)), dim3(256), 0, s, R, key, mpos);
    else hipLaunchKernelGGL(k_read_minimizer<uint64_t>, dim3(wave_grid(R.n)), dim3(256), 0, s, R, key, mpos);
}
void launch_solid_flags(hipStream_t s, ReadsDev R, BloomDev B, const uint16_t* rv16, const uint32_t* perm, const uint32_t* mpos, uint32_t* flags) {
    if (!R.n) return;
    const uint32_t g = (uint32_t)((R.n + 255) / 256);
#define SF_LAUNCH(KT, NHV) hipLaunchKernelGGL((k_solid_flags<KT, NHV>), dim3(g), dim3(256), 0, s, R, B, rv16, perm, mpos, flags)
    if (R.k >= 32) { if (B.n_hash == 7) SF_LAUNCH(u128, 7); else SF_LAUNCH(u128, 0); }
    else { if (B.n_hash == 7) SF_LAUNCH(uint64_t, 7); else SF_LAUNCH(uint64_t, 0); }
#undef SF_LAUNCH
}
void launch_qual_rewrite(hipStream_t s, ReadsDev R, const uint32_t* flags, uint8_t* quals) {
    if (!R.n) return;
    hipLaunchKernelGGL(k_qual_rewrite, dim3(wave_grid(R.n)), dim3(256), 0, s, R, flags, quals);
}
void launch_qual_smooth(hipStream_t s, ReadsDev R, BloomDev B, const uint16_t* rv16, uint8_t* quals) {
    if (!R.n) return;
    if (R.k >= 32) hipLaunchKernelGGL(k_qual_smooth<u128>, dim3(wave_grid(R.n)), dim3(256), 0, s, R, B, rv16, quals);
    else hipLaunchKernelGGL(k_qual_smooth<uint64_t>, dim3(wave_grid(R.n)), dim3(256), 0, s, R, B, rv16, quals);
}

}  // namespace leon

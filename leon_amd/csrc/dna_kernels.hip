// dna_kernels.hip -- gfx950 kernels of Leon's DNA encode path, everything except the range coder.
// Upstream functions named in comments are gatb-core names [RECALLED]; the source
// is absent from /root/reference, so there is no file:line to cite.
// Kernels that touch k-mers are templates on the k-mer type K (uint64_t for k < 32, unsigned __int128 for
// 32 <= k < 64); the launchers pick the instance from k.
#include "kernels.h"

namespace leon {

static inline uint32_t grid_for(uint64_t items, uint32_t per_block, uint32_t cap = 256 * 32) {
    uint64_t g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (uint32_t)g;
}
#define DISPATCH_K(k, CALL) do { if ((k) >= 32) { typedef u128 K; CALL; } else { typedef uint64_t K; CALL; } } while (0)

// ================================================================================================
// bloom: BloomNeighborCoherent::insert / contains / contains4      (k-mers: W words each, low word first)
// ================================================================================================
template <typename K>
__global__ void __launch_bounds__(256) k_bloom_insert(BloomDev B, const uint16_t* rv16g, const uint64_t* kmers, uint64_t n) {
    __shared__ uint16_t rv16[256];
    load_rv16(rv16, rv16g);
    uint32_t* words = (uint32_t*)B.bits;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        BloomKeys Kk;
        const uint32_t pv = bloom_item_keys<K>(B, rv16, load_kmer<K>(kmers + i * KT<K>::W), Kk);
        for (uint32_t h = 0; h < B.n_hash; h++) {
            uint64_t pos = Kk.racine + Kk.key[h] + pv;
            atomicOr(&words[pos >> 5], 1u << (pos & 31));
        }
    }
}
void launch_bloom_insert(hipStream_t s, BloomDev B, const uint16_t* rv16, const uint64_t* kmers, uint64_t n) {
    if (!n) return;
    DISPATCH_K(B.k, hipLaunchKernelGGL(k_bloom_insert<K>, dim3(grid_for(n, 256)), dim3(256), 0, s, B, rv16, kmers, n));
}

template <typename K>
__global__ void __launch_bounds__(256) k_bloom_query(BloomDev B, const uint16_t* rv16g, const uint64_t* kmers, uint64_t n,
                                                    int mode, uint8_t* out) {
    __shared__ uint16_t rv16[256];
    load_rv16(rv16, rv16g);
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const K km = load_kmer<K>(kmers + i * KT<K>::W);
        if (mode == 0) out[i] = bloom_contains<K>(B, rv16, km) ? 1 : 0;
        else out[i] = (uint8_t)bloom_contains4<K>(B, rv16, km, revcomp(km, B.k), mode == 2);
    }
}
void launch_bloom_query(hipStream_t s, BloomDev B, const uint16_t* rv16, const uint64_t* kmers, uint64_t n, int mode, uint8_t* out) {
    if (!n) return;
    DISPATCH_K(B.k, hipLaunchKernelGGL(k_bloom_query<K>, dim3(grid_for(n, 256)), dim3(256), 0, s, B, rv16, kmers, n, mode, out));
}

// ================================================================================================
// pack: ASCII -> 2-bit words + N mask  (DnaEncoder::buildKmers replaces N by 'A' and remembers _Npos)
// ================================================================================================
__global__ void k_read_slots(const uint64_t* off, uint64_t n, uint64_t* slots, uint32_t* bad) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t len = 0;
        if (i < n) {
            len = off[i + 1] - off[i];
            if (off[i + 1] < off[i] || len >= (1ull << 31)) { atomicOr(bad, 1u); len = 0; }     // not an offsets array: refused by the caller
        }
        slots[i] = (len + 31) / 32;
    }
}
__global__ void k_rebase_offsets(uint64_t* off, uint64_t n, uint64_t base) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) off[i] -= base;
}
void launch_rebase_offsets(hipStream_t s, uint64_t* off, uint64_t n, uint64_t base) {
    hipLaunchKernelGGL(k_rebase_offsets, dim3(grid_for(n, 256)), dim3(256), 0, s, off, n, base);
}
void launch_read_slots(hipStream_t s, const uint64_t* off, uint64_t n, uint64_t* slots, uint32_t* bad) {
    hipLaunchKernelGGL(k_read_slots, dim3(grid_for(n + 1, 256)), dim3(256), 0, s, off, n, slots, bad);
}

// One lane per OUTPUT dword (16 bases): a workgroup takes 64 consecutive reads, stages their slot and base offsets in LDS and
// walks the dwords of those reads in order, so neighbouring lanes load neighbouring 16-byte pieces of the caller's bases --
// coalesced loads in, coalesced dword stores out.  (It was one lane per read: 64 lanes striding 150 bytes apart, 0.83 TB/s.)
constexpr uint32_t PACK_READS = 256;
constexpr uint32_t PACK_MAP = 4096;          // slots whose read a byte of LDS names (256 reads of up to 512 bases; longer ones are found by bisection)
__global__ void __launch_bounds__(256) k_pack(const uint8_t* bases, const uint64_t* off, const uint64_t* slot_off, uint64_t n,
                                             uint32_t* packed, uint32_t* nmask, uint32_t* len_out, uint32_t* ncount) {
    __shared__ uint64_t so[PACK_READS + 1], bo[PACK_READS + 1];
    __shared__ uint32_t nn[PACK_READS];
    __shared__ uint8_t rmap[PACK_MAP];
    const uint64_t r0 = blockIdx.x * (uint64_t)PACK_READS;
    const uint32_t nr = (uint32_t)(n - r0 < PACK_READS ? n - r0 : PACK_READS);
    for (uint32_t t = threadIdx.x; t <= nr; t += blockDim.x) { so[t] = slot_off[r0 + t]; bo[t] = off[r0 + t]; }     // (nr + 1 entries: one more than the workgroup has threads)
    for (uint32_t t = threadIdx.x; t < PACK_READS; t += blockDim.x) nn[t] = 0;
    __syncthreads();
    const uint64_t s0 = so[0];
    const uint64_t total_dw = 2 * (so[nr] - s0);
    const bool mapped = so[nr] - s0 <= PACK_MAP;              // (round 5: six dependent LDS reads per dword were most of this kernel's time)
    if (mapped) {
        for (uint32_t t = threadIdx.x; t < nr; t += blockDim.x) for (uint64_t q = so[t]; q < so[t + 1]; q++) rmap[q - s0] = (uint8_t)t;
        __syncthreads();
    }
    uint16_t* nmask16 = (uint16_t*)nmask;
    for (uint64_t d = threadIdx.x; d < total_dw; d += blockDim.x) {
        const uint64_t slot = s0 + (d >> 1);
        uint32_t lo = 0, hi = nr;                              // the read r with so[r] <= slot < so[r + 1]
        if (mapped) lo = rmap[d >> 1];
        else while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (so[mid] <= slot) lo = mid; else hi = mid; }
        const uint32_t r = lo;
        const uint32_t len = (uint32_t)(bo[r + 1] - bo[r]);
        const uint32_t j0 = (uint32_t)(2 * s0 + d - 2 * so[r]) * 16;   // first base of this dword inside the read
        const uint8_t* src = bases + bo[r];
        uint32_t fours[4] = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};     // "AAAA" past the end of the read
        if (j0 + 16 <= len) __builtin_memcpy(fours, src + j0, 16);                   // one (unaligned) 16-byte load
        else {
#pragma unroll
            for (uint32_t q4 = 0; q4 < 4; q4++) {
                const uint32_t p0 = j0 + 4 * q4;
                uint32_t four = 0x41414141u;
                if (p0 + 4 <= len) __builtin_memcpy(&four, src + p0, 4);
                else if (p0 < len) { for (uint32_t j = 0; p0 + j < len; j++) four = (four & ~(0xFFu << (8 * j))) | ((uint32_t)src[p0 + j] << (8 * j)); }
                fours[q4] = four;
            }
        }
        // four bases at a time: a byte is valid when it equals 'A', 'C', 'G' or 'T' (exact per-byte zero test of the XOR: the 7-bit add cannot carry
        // into the next byte), its code is bits 2..1; the four codes / flags are gathered into a byte / nibble by one multiply each
        uint32_t word = 0, nb = 0;
#pragma unroll
        for (uint32_t q4 = 0; q4 < 4; q4++) {
            const uint32_t four = fours[q4];
            auto eq = [&](uint32_t letter) { const uint32_t x = four ^ (letter * 0x01010101u); return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u; };
            const uint32_t ok = (eq('A') | eq('C') | eq('G') | eq('T')) >> 7;              // 1 per valid byte
            const uint32_t codes = (four >> 1) & 0x03030303u & (ok * 3u);
            word |= ((codes * 0x40100401u) >> 24) << (24 - 8 * q4);
            nb |= (((ok ^ 0x01010101u) * 0x01020408u) >> 24) << (4 * q4);
        }
        packed[2 * s0 + d] = word;
        nmask16[2 * s0 + d] = (uint16_t)nb;                    // the even dword's 16 flags are the low half of the slot's mask word
        if (nb) atomicAdd(&nn[r], (uint32_t)__popc(nb));
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < nr; t += blockDim.x) { len_out[r0 + t] = (uint32_t)(bo[t + 1] - bo[t]); ncount[r0 + t] = nn[t]; }
}
void launch_pack(hipStream_t s, const uint8_t* bases, const uint64_t* off, const uint64_t* slot_off, uint64_t n,
                 uint32_t* packed, uint32_t* nmask, uint32_t* len, uint32_t* ncount) {
    if (!n) return;
    hipLaunchKernelGGL(k_pack, dim3((uint32_t)((n + PACK_READS - 1) / PACK_READS)), dim3(256), 0, s, bases, off, slot_off, n, packed, nmask, len, ncount);
}


// ================================================================================================
// anchor dictionary (Leon::anchorExist / findAndInsertAnchor with sequential, -nb-cores 1 semantics)
// Slot layout (kernels.h DictDev): one-word keys { key, fin }, KEY_EMPTY when free; two-word keys { lo, hi, fin, tent },
// the high word (< 2^62 for a real k-mer) doubles as the claim word: KEY_EMPTY free, KEY_LOCKED while the low word is written.
// A look-up reads the key AND fin with one 16-byte load (one-word keys) or from the same 32-byte half sector.
// ================================================================================================
template <typename K> struct DS;
template <> struct DS<uint64_t> { static constexpr uint32_t STRIDE = 2, CLAIM = 0, FIN = 1; };
template <> struct DS<u128> { static constexpr uint32_t STRIDE = 4, CLAIM = 1, FIN = 2; };
template <typename K> __device__ inline uint64_t* slot_ptr(const DictDev& D, uint64_t slot) { return D.slots + slot * DS<K>::STRIDE; }
template <typename K> __device__ inline uint64_t* fin_ptr(const DictDev& D, uint64_t slot) { return slot_ptr<K>(D, slot) + DS<K>::FIN; }
__device__ inline uint64_t* tent_ptr(const DictDev& D, uint64_t slot) { return D.tent + slot * D.tstride; }

__global__ void k_dict_init(DictDev D, uint64_t cap, uint32_t W) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * blockDim.x) {
        if (W == 2) { D.slots[4 * i] = KEY_EMPTY; D.slots[4 * i + 1] = KEY_EMPTY; D.slots[4 * i + 2] = IDX_INF; D.slots[4 * i + 3] = IDX_INF; }
        else { D.slots[2 * i] = KEY_EMPTY; D.slots[2 * i + 1] = IDX_INF; D.tent[i] = IDX_INF; }
        D.addr[i] = 0;
    }
}
void launch_dict_init(hipStream_t s, DictDev D, uint64_t cap, uint32_t W) {
    hipLaunchKernelGGL(k_dict_init, dim3(grid_for(cap, 256)), dim3(256), 0, s, D, cap, W);
}

// ---- look-up (any lane; 0xFFFFFFFF if absent); fin = the slot's fin word as the same load saw it.  A key being inserted
// by the running kernel may be missed: such a key is only proposed (fin = INF), so missing it changes nothing; a fin
// older than a concurrent k_check store reads as "not final yet", which that round treats as blocked (tent <= the storer).
__device__ inline uint32_t dict_find(const DictDev& D, uint64_t key, uint64_t& fin) {
    uint64_t slot = key_hash(key) & D.mask;
    for (;;) {
        const ulonglong2 v = *(const ulonglong2*)(D.slots + 2 * slot);          // one 16-byte load: key and fin
        if (v.x == key) { fin = v.y; return (uint32_t)slot; }
        if (v.x == KEY_EMPTY) return 0xFFFFFFFFu;
        slot = (slot + 1) & D.mask;
    }
}
__device__ inline uint32_t dict_find(const DictDev& D, u128 key, uint64_t& fin) {
    const uint64_t klo = (uint64_t)key, khi = (uint64_t)(key >> 64);
    uint64_t slot = key_hash(key) & D.mask;
    for (;;) {
        const ulonglong2 v = *(const ulonglong2*)(D.slots + 4 * slot);          // { lo, hi }
        if (v.y == KEY_EMPTY) return 0xFFFFFFFFu;
        if (v.y == khi && v.x == klo) { fin = D.slots[4 * slot + 2]; return (uint32_t)slot; }   // same sector as the key
        slot = (slot + 1) & D.mask;
    }
}
// (the bit of a key in the per-window filters: one 32-bit multiply, where key_hash -- the dictionary's, two 64-bit multiplies per k-mer
// of every read that k_check and k_final_pos look at -- was a good part of those kernels' instructions)
__device__ inline uint32_t fold32(uint64_t x) { return (uint32_t)x ^ (uint32_t)(x >> 32); }
__device__ inline uint32_t fold32(u128 x) { return fold32((uint64_t)x) ^ (fold32((uint64_t)(x >> 64)) * 0x85EBCA6Bu); }
template <typename K> __device__ inline uint32_t window_bit(K key) { return ((fold32(key) ^ 0x5bd1e995u) * 0xCC9E2D51u) >> (32 - WBITS_LOG2); }
// ---- the final keys' filter (DictDev::fbits, kernels.h minimizer_geometry) ----
// Look-ups of it were most of the resolution's memory traffic when a key's bit sat at a place of the key's own hash: ~50 k-mers of
// every read, each in another 64-byte sector of a table that no L2 keeps beside the dictionary's probes.  A key's WORD is now chosen by
// its minimizer, which ~P/2 consecutive k-mers of a read share, and only the two bits inside the word by the key's hash: the lanes
// of a step ask for two or three words between them, and the next step mostly for the same ones.
__device__ inline uint32_t mmer_hash(uint32_t x, uint32_t m) {          // x: an m-mer, right-aligned; the hash of its canonical form
    const uint32_t y = rev2bit32(x) >> (32 - 2 * m);
    const uint32_t h = (y < x ? y : x) * 0x9E3779B1u;
    return h ^ (h >> 15);
}
template <typename K> __device__ inline uint32_t mmer_of(K km, uint32_t k, uint32_t m, uint32_t t) {    // the m-mer at offset t of a k-mer
    const uint32_t x = (uint32_t)(km >> (2 * (k - m - t)));
    return m == 16 ? x : x & ((1u << (2 * m)) - 1u);
}
__device__ inline uint32_t filter_word(const DictDev& D, uint32_t hmin) { return (hmin * 0x85EBCA6Bu) >> D.fwshift; }
// (a hash of its own, one 32-bit multiply: the dictionary's key_hash -- two 64-bit multiplies -- is now only formed behind a "maybe")
template <typename K> __device__ inline uint64_t filter_bits(K key) {
    const uint32_t h = fold32(key) * 0x9E3779B1u;
    return (1ull << (h >> 26)) | (1ull << ((h >> 20) & 63u));
}
// the minimizer of ONE key, by a whole wave: lane t < P takes the m-mer at offset c + t (every lane returns the minimum of lanes 0..31)
template <typename K> __device__ inline uint32_t key_minimizer_wave(const DictDev& D, K key, uint32_t k, uint32_t lane) {
    uint32_t h = 0xFFFFFFFFu;
    if (lane < D.mm_P) h = mmer_hash(mmer_of(key, k, D.mm_m, D.mm_c + lane), D.mm_m);
    for (int d = 1; d < 32; d <<= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)h, d); h = o < h ? o : h; }
    return h;
}
// Sliding minima along a read, 16 positions (one per lane of a quarter-wave = a DPP row) per step: level j of a position is the
// minimum over the 2^j positions ending there; what lies before the row's first lane comes from the step before (`prev`).
struct MinLevels { uint32_t L0, L1, L2, L3, L4; };
#define LEON_ROW_SHR(x, n) ((uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)(x), 0x110 + (n), 0xF, 0xF, false))   /* lane l <- lane l - n */
#define LEON_ROW_SHL(x, n) ((uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)(x), 0x100 + (n), 0xF, 0xF, false))   /* lane l <- lane l + n */
__device__ inline uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { const uint32_t t = a < b ? a : b; return t < c ? t : c; }
__device__ inline MinLevels sliding_levels(uint32_t L0, const MinLevels& prev) {
    MinLevels c;
    c.L0 = L0;
    c.L1 = umin3(c.L0, LEON_ROW_SHR(c.L0, 1), LEON_ROW_SHL(prev.L0, 15));
    c.L2 = umin3(c.L1, LEON_ROW_SHR(c.L1, 2), LEON_ROW_SHL(prev.L1, 14));
    c.L3 = umin3(c.L2, LEON_ROW_SHR(c.L2, 4), LEON_ROW_SHL(prev.L2, 12));
    c.L4 = umin3(c.L3, LEON_ROW_SHR(c.L3, 8), LEON_ROW_SHL(prev.L3, 8));
    return c;
}
__device__ inline uint32_t window_minimum(const MinLevels& c, const MinLevels& prev, uint32_t P) {
    return P == 32 ? (c.L4 < prev.L4 ? c.L4 : prev.L4) : P == 16 ? c.L4 : P == 8 ? c.L3 : P == 4 ? c.L2 : P == 2 ? c.L1 : c.L0;
}
// (`created` is set when the key was not there: the caller counts new keys, one atomic per wave where it matters)
// ---- find or insert.  SPIN = true: called by ONE lane per wave (a lane may wait for another wave's insert to
// complete); SPIN = false: the keys being inserted are all distinct (rehash), a locked slot is someone else's.
template <bool SPIN> __device__ inline uint32_t dict_find_or_insert(const DictDev& D, uint64_t key, bool& created) {
    uint64_t slot = key_hash(key) & D.mask;
    for (;;) {
        uint64_t* pk = D.slots + 2 * slot;
        uint64_t cur = *pk;
        if (cur == key) return (uint32_t)slot;
        if (cur == KEY_EMPTY) {
            uint64_t old = atomicCAS((unsigned long long*)pk, (unsigned long long)KEY_EMPTY, (unsigned long long)key);
            if (old == KEY_EMPTY) { created = true; return (uint32_t)slot; }
            if (old == key) return (uint32_t)slot;
        }
        slot = (slot + 1) & D.mask;
    }
}
template <bool SPIN> __device__ inline uint32_t dict_find_or_insert(const DictDev& D, u128 key, bool& created) {
    const uint64_t klo = (uint64_t)key, khi = (uint64_t)(key >> 64);
    uint64_t slot = key_hash(key) & D.mask;
    for (;;) {
        uint64_t* plo = D.slots + 4 * slot;
        uint64_t* phi = plo + 1;
        uint64_t hi = __hip_atomic_load(phi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (hi == KEY_EMPTY) {
            hi = atomicCAS((unsigned long long*)phi, (unsigned long long)KEY_EMPTY, (unsigned long long)KEY_LOCKED);
            if (hi == KEY_EMPTY) {                                       // ours: low word first, then publish the high word
                // Both words are written with relaxed read-modify-write atomics, which execute in issue order at the one L2
                // channel that owns the slot's cache line: whoever sees the high word sees the low word.  (A release store
                // at agent scope instead costs a write-back of the XCD's whole L2 -- `buffer_wbl2 sc1` -- PER INSERT:
                // a window in which every read inserts took 33 ms at k = 63 against 2.4 ms at k = 31.)
                (void)__hip_atomic_exchange(plo, klo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                (void)__hip_atomic_exchange(phi, khi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                created = true;
                return (uint32_t)slot;
            }
        }
        if (SPIN) {
            // the other inserter is a lane of ANOTHER wave that is two atomics away from publishing: it cannot be waiting for
            // us.  Should the wait ever run out (a stalled wave), the batch is failed through D.err rather than probing on
            // and inserting the same key twice.
            uint32_t spin = 0;
            for (; hi == KEY_LOCKED && spin < (1u << 22); spin++) {
                __builtin_amdgcn_s_sleep(1);
                hi = __hip_atomic_load(phi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (hi == KEY_LOCKED) { atomicExch(D.err, 1); return (uint32_t)slot; }
        }
        if (hi == khi) {
            uint64_t lo = __hip_atomic_load(plo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // ORDERING ASSUMPTION: the inserter's two exchanges reach the slot's L2 channel in issue order, so a
            // reader that sees the high word sees the low word.  That is observed behaviour of this memory system, not a
            // documented guarantee, so it is checked, not trusted: a low word that still reads as the free slot's ~0 is given
            // a moment, and if it stays ~0 the batch is FAILED through D.err (like the locked-slot timeout) -- probing on
            // would insert the same k-mer twice and give a silently different dictionary.  (Not checkable for the one low
            // word that IS ~0, a k-mer ending in 32 G: there ~0 is the answer either way.)
            if (SPIN && lo == KEY_EMPTY && klo != KEY_EMPTY) {
                for (uint32_t spin = 0; lo == KEY_EMPTY && spin < (1u << 16); spin++) {
                    __builtin_amdgcn_s_sleep(1);
                    lo = __hip_atomic_load(plo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (lo == KEY_EMPTY) { atomicExch(D.err, 1); return (uint32_t)slot; }
            }
            if (lo == klo) return (uint32_t)slot;
        }
        slot = (slot + 1) & D.mask;
    }
}
template <typename K> __device__ inline K dict_key(const DictDev& D, uint32_t slot);
template <> __device__ inline uint64_t dict_key<uint64_t>(const DictDev& D, uint32_t slot) { return D.slots[2 * (uint64_t)slot]; }
template <> __device__ inline u128 dict_key<u128>(const DictDev& D, uint32_t slot) { return ((u128)D.slots[4 * (uint64_t)slot + 1] << 64) | D.slots[4 * (uint64_t)slot]; }

template <typename K> __global__ void k_dict_rehash(DictDev from, uint64_t from_cap, DictDev to) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = blockIdx.x * (uint64_t)blockDim.x; i0 < from_cap; i0 += stride) {     // (whole waves stay in the loop: ballot below)
        const uint64_t i = i0 + threadIdx.x;
        bool created = false;
        if (i < from_cap && from.slots[i * DS<K>::STRIDE + DS<K>::CLAIM] != KEY_EMPTY) {
            uint32_t s = dict_find_or_insert<false>(to, dict_key<K>(from, (uint32_t)i), created);
            *fin_ptr<K>(to, s) = *fin_ptr<K>(from, i); *tent_ptr(to, s) = IDX_INF; to.addr[s] = from.addr[i];
        }
        const unsigned long long cm = __ballot(created);                                     // one atomic per wave on the key counter
        if (cm && lane_id() == (uint32_t)__builtin_ctzll(cm)) atomicAdd(to.n_keys, (unsigned long long)__popcll(cm));
    }
}
void launch_dict_rehash(hipStream_t s, DictDev from, uint64_t from_cap, DictDev to, uint32_t k) {
    DISPATCH_K(k, hipLaunchKernelGGL(k_dict_rehash<K>, dim3(grid_for(from_cap, 256)), dim3(256), 0, s, from, from_cap, to));
}

template <typename K> __device__ inline K canon_at(const uint32_t* pk, uint32_t p, uint32_t k) {
    const K km = kmer_at<K>(pk, p, k);
    const K rc = revcomp(km, k);
    return rc < km ? rc : km;
}

// 16-lane variant of canon_from_words: lanes gbase .. gbase+7 of each quarter-wave hold the dwords of that quarter's read
template <typename K> __device__ inline K fwd_from_words16(uint32_t words, uint32_t gbase, uint32_t base, uint32_t p, uint32_t k);
template <> __device__ inline uint64_t fwd_from_words16<uint64_t>(uint32_t words, uint32_t gbase, uint32_t base, uint32_t p, uint32_t k) {
    const int d = (int)(gbase + (p >> 4) - (base >> 4));
    return kmer_from3((uint32_t)__shfl((int)words, d), (uint32_t)__shfl((int)words, d + 1), (uint32_t)__shfl((int)words, d + 2), p & 15, k);
}
template <> __device__ inline u128 fwd_from_words16<u128>(uint32_t words, uint32_t gbase, uint32_t base, uint32_t p, uint32_t k) {
    const int d = (int)(gbase + (p >> 4) - (base >> 4));             // p - base < 16: d - gbase <= 1, d + 4 - gbase <= 5 < 8
    return kmer_from5((uint32_t)__shfl((int)words, d), (uint32_t)__shfl((int)words, d + 1), (uint32_t)__shfl((int)words, d + 2),
                      (uint32_t)__shfl((int)words, d + 3), (uint32_t)__shfl((int)words, d + 4), p & 15, k);
}
template <typename K> __device__ inline K canon_from_words16(uint32_t words, uint32_t gbase, uint32_t base, uint32_t p, uint32_t k) {
    const K km = fwd_from_words16<K>(words, gbase, base, p, k);
    const K rc = revcomp(km, k);
    return rc < km ? rc : km;
}

// Pass A of a window: DnaEncoder::findExistingAnchor against the dictionary as it stood before the window,
// else the candidate Leon::findAndInsertAnchor would insert.  Four reads per wave, 16 k-mer positions per read and
// step: the first anchor sits ~20 positions into a read, so quarter-waves look up ~1.7x fewer k-mers than whole
// waves would, and a wave keeps four independent reads' memory requests in flight.
// TRACE (LEON_TRACE_RESOLVE=1, a measurement aid): what a read costs by its outcome -- per class (found an anchor / goes on to insert / no
// anchor at all) the reads, their filter probes, the dictionary probes behind a filter "maybe" and their bloom probes (k-mers
// tested), summed per wave and added to trace[12] once per wave.
template <typename K, bool TRACE>
__global__ void __launch_bounds__(256, (KT<K>::W == 2 ? 5 : 8)) k_lookup_cand(ReadsDev R, BloomDev B, const uint16_t* rv16g, DictDev D, ResolveDev V,
                                                    uint64_t w0, uint64_t w1, uint64_t first_global,
                                                    uint32_t* ulist, uint32_t* ucount, unsigned long long* trace,
                                                    uint64_t* xres, uint64_t xbase, uint32_t dry) {
    // xres (the look-ups of a window divided among the ranks of a job, leon_dna_set_gather): what the pass found for read i, as
    // (position << 8) | status at xres[i - xbase], for the ranks that did not look this read up; dry: ONLY that -- no status, no
    // proposal, no list entry (LEON_XCH_EMULATE: another rank's share, computed here in its stead)
    __shared__ uint16_t rv16[256];
    load_rv16(rv16, rv16g);
    const uint32_t lane = lane_id(), k = R.k;
    const uint32_t grp = lane >> 4, l = lane & 15, gbase = grp * 16;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    enum : uint32_t { PH_LOOKUP = 0, PH_SEG_A = 1, PH_SEG_B = 2, PH_SEG_C = 3, PH_DONE = 4 };
    __shared__ uint32_t lists[4][64];                             // per wave: window-relative indices of its unresolved reads
    uint32_t* my_list = lists[threadIdx.x >> 6];
    uint32_t n_list = 0, n_created = 0;                           // wave-uniform
    auto flush_list = [&]() {
        if (n_list) {
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(ucount, n_list);
            at = (uint32_t)__shfl((int)at, 0);
            __builtin_amdgcn_wave_barrier();
            if (lane < n_list) ulist[at + lane] = my_list[lane];
            __builtin_amdgcn_wave_barrier();
        }
        if (n_created && lane == 0) atomicAdd(D.n_keys, (unsigned long long)n_created);
        n_list = 0; n_created = 0;
    };
    // Every quarter-wave walks ITS OWN sequence of reads (i, i + 4 * nwaves, ...) and takes the next one as soon as it is done
    // with the current one: the kernel is bound by the number of wave-wide memory instructions (a read that inserts an anchor
    // needs ~40, one that finds an anchor ~10; with four reads in lockstep the wave paid for the slowest), so the other three
    // quarters must not idle while one read goes through all of its k-mers and the bloom scan.
    uint32_t t_filter = 0, t_dict = 0, t_bloom = 0;               // (TRACE) this quarter-wave's current read: probes so far, counted on lane l == 0
    unsigned long long tr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto trace_done = [&](uint32_t cls) { if (TRACE && l == 0) { tr[cls * 4] += 1; tr[cls * 4 + 1] += t_filter; tr[cls * 4 + 2] += t_dict; tr[cls * 4 + 3] += t_bloom; } t_filter = t_dict = t_bloom = 0; };
    uint64_t i = w0 + 4 * wave + grp;
    bool have = false, started = false, exhausted = false;
    uint32_t len = 0, nk = 0, iMin = 0, iMax = 0, phase = PH_DONE, base = 0, limit = 0;
    uint64_t g = 0;
    const uint32_t* pk = R.packed;
    const uint32_t mm_m = D.mm_m, mm_P = D.mm_P, mm_c = D.mm_c;
    const uint64_t* const fwords = (const uint64_t*)D.fbits;
    MinLevels prev = {~0u, ~0u, ~0u, ~0u, ~0u};                   // this quarter-wave's read: the m-mers' sliding minima as of the step before
    for (;;) {
        if (phase == PH_DONE && !exhausted) {                     // this quarter-wave's next read
            if (started) i += 4 * nwaves;
            started = true;
            have = i < w1;
            exhausted = !have;
            if (have) {
                len = R.len[i];
                g = first_global + i;
                pk = R.packed + 2 * R.slot_off[i];
                nk = len >= k ? len - k + 1 : 0;
                // Leon::findAndInsertAnchor scan order: [n/2, n/2+10), [0, n/2), [n/2+10, n)
                iMin = nk / 2; iMax = nk / 2 + 10 > nk ? nk : nk / 2 + 10;
                base = 0; limit = nk;
                if (nk) phase = PH_LOOKUP;
                else { if (l == 0) { if (!dry) V.status[i] = ST_NOANCHOR; if (xres) xres[i - xbase] = ST_NOANCHOR; } trace_done(2); }      // (stays PH_DONE: the next round takes another read)
            }
        }
        if (!__any(have)) break;                                  // every quarter-wave has run out of reads
        {
            const bool active = phase != PH_DONE;
            // a phase whose range is exhausted moves on (empty segments fall through in later iterations)
            if (active && base >= limit) {
                if (phase == PH_LOOKUP) { phase = PH_SEG_A; base = iMin; limit = iMax; }
                else if (phase == PH_SEG_A) { phase = PH_SEG_B; base = 0; limit = iMin; }
                else if (phase == PH_SEG_B) { phase = PH_SEG_C; base = iMax; limit = nk; }
                else { phase = PH_DONE; if (l == 0) { if (!dry) V.status[i] = ST_NOANCHOR; if (xres) xres[i - xbase] = ST_NOANCHOR; } trace_done(2); }
            }
            const bool run = phase != PH_DONE && base < limit;
            const uint32_t p = base + l;
            const bool valid = run && p < limit;
            const uint32_t words = (run && l < 8) ? pk[(base >> 4) + l] : 0u;
            const K km = fwd_from_words16<K>(words, gbase, base, valid ? p : (run ? limit - 1 : 0), k);     // all 64 lanes
            const K rck = revcomp(km, k);
            const K cn = rck < km ? rck : km;
            // The filter word of every k-mer of the step: the smallest of the P m-mer hashes ending at offset c + P - 1 of the lane's own
            // k-mer -- one new m-mer per lane and step, the ones before it from the lanes below and from the step before.
            const bool look = run && phase == PH_LOOKUP;              // (the same for the 16 lanes of a quarter-wave: a DPP row)
            if (look && base == 0) {                                  // a read begins: the P - 1 m-mers before its first step's
                prev = MinLevels{~0u, ~0u, ~0u, ~0u, ~0u};
                for (int r = mm_P == 32 ? 1 : 0; r >= 0; r--) {
                    const int x = (int)(mm_c + mm_P - 1) - 16 * (r + 1) + (int)l;
                    const uint32_t xc = x > 0 ? (uint32_t)x : 0u;
                    const uint32_t w0 = (uint32_t)__shfl((int)words, (int)(gbase + (xc >> 4))), w1 = (uint32_t)__shfl((int)words, (int)(gbase + (xc >> 4) + 1));
                    const uint32_t mm = (uint32_t)((((((uint64_t)w0) << 32) | w1) << (2 * (xc & 15))) >> (64 - 2 * mm_m));
                    prev = sliding_levels(x >= (int)mm_c && xc + mm_m <= len ? mmer_hash(mm, mm_m) : ~0u, prev);
                }
            }
            const MinLevels cur = sliding_levels(look && valid ? mmer_hash(mmer_of(km, k, mm_m, mm_c + mm_P - 1), mm_m) : ~0u, prev);
            const uint32_t hmin = window_minimum(cur, prev, mm_P);
            if (look) prev = cur;
            bool hit = false; uint32_t slot = 0xFFFFFFFFu;
            if (valid) {
                if (phase == PH_LOOKUP) {
                    // ~95 % of the look-ups miss (one k-mer in ~35 is an anchor): the filter answers most of them without a random
                    // sector of the dictionary
                    const uint64_t fb = filter_bits(cn);
                    const bool maybe = (fwords[filter_word(D, hmin)] & fb) == fb;
                    if (maybe) { uint64_t fin = IDX_INF; slot = dict_find(D, cn, fin); hit = slot != 0xFFFFFFFFu && fin < g; }
                    if (TRACE) {
                        const uint32_t nf = (uint32_t)__popcll(__ballot(true) >> gbase & 0xFFFFull), nd = (uint32_t)__popcll(__ballot(maybe) >> gbase & 0xFFFFull);
                        t_filter += nf; t_dict += nd;
                    }
                }
                else { hit = bloom_contains<K>(B, rv16, cn); if (TRACE) t_bloom += (uint32_t)__popcll(__ballot(true) >> gbase & 0xFFFFull); }
            }
            const unsigned long long bal = __ballot(hit);
            const uint32_t gb = (uint32_t)(bal >> gbase) & 0xFFFFu;
            bool want_insert = false; uint32_t cpos = 0;
            if (run) {
                if (gb) {
                    const uint32_t f = (uint32_t)__builtin_ctz(gb);
                    const uint32_t hs = __shfl(slot, (int)(gbase + f));
                    if (phase == PH_LOOKUP) {
                        if (l == 0) {
                            if (!dry) { V.status[i] = ST_HIT; V.hit_pos[i] = base + f; V.hit_slot[i] = hs; }
                            if (xres) xres[i - xbase] = ((uint64_t)(base + f) << 8) | ST_HIT;
                        }
                        trace_done(0);
                    } else {
                        want_insert = (l == 0) && !dry; cpos = base + f;
                        if (l == 0 && xres) xres[i - xbase] = ((uint64_t)cpos << 8) | ST_UNRESOLVED;
                        trace_done(1);
                    }
                    phase = PH_DONE;
                } else base += 16;
            }
            // candidates go into the dictionary one quarter-wave at a time: with two-word keys an inserting lane may
            // wait for another inserter, which must not be a lane of its own wave
            bool created = false;
            uint32_t sl = 0;
            for (uint32_t q = 0; q < 4; q++) {
                if (want_insert && grp == q) {
                    const K ck = canon_at<K>(pk, cpos, k);
                    sl = dict_find_or_insert<true>(D, ck, created);
                    atomicMin((unsigned long long*)tent_ptr(D, sl), (unsigned long long)g);
                    const uint32_t pb = window_bit(ck);                     // k_check only looks up keys that are final or proposed
                    atomicOr(&D.pbits[pb >> 5], 1u << (pb & 31));
                    V.status[i] = ST_UNRESOLVED; V.cand_pos[i] = cpos; V.cand_slot[i] = sl;
                }
            }
            // The window's two counters get one atomic per ~60 inserting reads, not one per wave and step: the unresolved reads
            // are collected in the wave's own LDS list and handed over in batches (same-address atomics serialise in L2; one
            // per inserting wave-step was ~10^5 per window and most of this kernel's time).
            const unsigned long long wm = __ballot(want_insert);
            if (wm) {
                if (want_insert) my_list[n_list + (uint32_t)__popcll(wm & ((1ull << lane) - 1))] = (uint32_t)i;
                n_list += (uint32_t)__popcll(wm);
                n_created += (uint32_t)__popcll(__ballot(created));
                if (n_list > 64 - 4) flush_list();
            }
        }
    }
    flush_list();
    if (TRACE) {
        for (uint32_t j = 0; j < 12; j++) {
            unsigned long long v = tr[j];
            for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
            if (lane == 0 && v) atomicAdd(trace + j, v);
        }
    }
}
void launch_lookup_cand(hipStream_t s, ReadsDev R, BloomDev B, const uint16_t* rv16, DictDev D, ResolveDev V,
                        uint64_t w0, uint64_t w1, uint64_t first_global, uint32_t* ulist, uint32_t* ucount, unsigned long long* trace,
                        uint64_t* xres, uint64_t xbase, bool dry) {
    if (w1 <= w0) return;
    if (trace && !dry) { DISPATCH_K(R.k, hipLaunchKernelGGL((k_lookup_cand<K, true>), dim3(grid_for(w1 - w0, 16, 256 * 16)), dim3(256), 0, s, R, B, rv16, D, V, w0, w1,
                                                           first_global, ulist, ucount, trace, xres, xbase, 0u)); return; }
    // 4 reads per wave, 64 waves launched per CU (32 resident).  Measured and no better (resolve stage, 100 M reads): 32 waves per
    // CU in a grid-stride loop 312 ms, 128 per CU 287, workgroups of 64 or 128 threads 310-330, against 290 for this geometry.
    const uint32_t per_cu = [] { const char* e = getenv("LEON_LOOKUP_BLOCKS_PER_CU"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 1024 ? (uint32_t)v : 16u; }();   // (measurement)
    DISPATCH_K(R.k, hipLaunchKernelGGL((k_lookup_cand<K, false>), dim3(grid_for(w1 - w0, 16, 256 * per_cu)), dim3(256), 0, s, R, B, rv16, D, V, w0, w1,
                                       first_global, ulist, ucount, trace, xres, xbase, dry ? 1u : 0u));
}

// The look-ups of a window divided among the ranks of a job: what ANOTHER rank's pass found for the reads of [w0, w1) outside this
// rank's own share [s0, s1) -- xres, gathered from all ranks -- becomes this rank's state exactly as its own pass would have left it:
// the status and position, the slot of the key found (one probe of the dictionary, which holds the same final keys on every rank),
// and for a read that goes on to propose: its candidate in the dictionary, its index in `tent`, the window's filter bit, the list.
template <typename K>
__global__ void __launch_bounds__(256) k_lookup_apply(ReadsDev R, DictDev D, ResolveDev V, uint64_t w0, uint64_t w1, uint64_t s0, uint64_t s1,
                                                     uint64_t first_global, const uint64_t* xres, uint32_t* ulist, uint32_t* ucount) {
    __shared__ uint32_t wave_n[4], wave_new[4], block_at;
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6, k = R.k;
    const uint64_t before = s0 - w0, n_out = (w1 - w0) - (s1 - s0);
    for (uint64_t e0 = blockIdx.x * (uint64_t)blockDim.x; e0 < n_out; e0 += (uint64_t)gridDim.x * blockDim.x) {   // (whole workgroups stay in the loop)
        const uint64_t e = e0 + threadIdx.x;
        const bool have = e < n_out;
        const uint64_t i = !have ? w0 : (e < before ? w0 + e : s1 + (e - before));
        const uint64_t x = have ? xres[i - w0] : (uint64_t)ST_NOANCHOR;
        const uint32_t st = (uint32_t)(x & 0xFFu), pos = (uint32_t)(x >> 8);
        const uint64_t g = first_global + i;
        const uint32_t* pk = R.packed + 2 * R.slot_off[i];
        const bool sane = have && (st == ST_NOANCHOR || ((st == ST_HIT || st == ST_UNRESOLVED) && R.len[i] >= k && pos <= R.len[i] - k));
        if (have && !sane) atomicExch(D.err, 2);                   // (what another rank sent is input)
        if (sane && st == ST_NOANCHOR) V.status[i] = ST_NOANCHOR;
        if (sane && st == ST_HIT) {
            uint64_t fin = IDX_INF;
            const uint32_t slot = dict_find(D, canon_at<K>(pk, pos, k), fin);
            if (slot == 0xFFFFFFFFu || !(fin < g)) atomicExch(D.err, 2);     // the ranks' dictionaries differ
            else { V.status[i] = ST_HIT; V.hit_pos[i] = pos; V.hit_slot[i] = slot; }
        }
        const bool want = sane && st == ST_UNRESOLVED;
        bool created = false;
        auto propose = [&]() {
            const K ck = canon_at<K>(pk, pos, k);
            const uint32_t sl = dict_find_or_insert<true>(D, ck, created);
            atomicMin((unsigned long long*)tent_ptr(D, sl), (unsigned long long)g);
            const uint32_t pb = window_bit(ck);
            atomicOr(&D.pbits[pb >> 5], 1u << (pb & 31));
            V.status[i] = ST_UNRESOLVED; V.cand_pos[i] = pos; V.cand_slot[i] = sl;
        };
        const unsigned long long wm = __ballot(want);
        if (KT<K>::W == 1) { if (want) propose(); }
        else {                                                     // two-word keys: one inserting lane of a wave at a time (dict_find_or_insert)
            for (unsigned long long m = wm; m; m &= m - 1) if (lane == (uint32_t)__builtin_ctzll(m)) propose();
        }
        // one atomic per workgroup and pass on each of the window's two counters
        const unsigned long long cm = __ballot(created);
        if (lane == 0) { wave_n[wv] = (uint32_t)__popcll(wm); wave_new[wv] = (uint32_t)__popcll(cm); }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tot = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3], fresh = wave_new[0] + wave_new[1] + wave_new[2] + wave_new[3];
            block_at = tot ? atomicAdd(ucount, tot) : 0u;
            if (fresh) atomicAdd(D.n_keys, (unsigned long long)fresh);
        }
        __syncthreads();
        if (want) {
            uint32_t at = block_at + (uint32_t)__popcll(wm & ((1ull << lane) - 1));
            for (uint32_t w = 0; w < wv; w++) at += wave_n[w];
            ulist[at] = (uint32_t)i;
        }
        __syncthreads();
    }
}
void launch_lookup_apply(hipStream_t s, ReadsDev R, DictDev D, ResolveDev V, uint64_t w0, uint64_t w1, uint64_t s0, uint64_t s1,
                         uint64_t first_global, const uint64_t* xres, uint32_t* ulist, uint32_t* ucount) {
    if (w1 - w0 <= s1 - s0) return;
    DISPATCH_K(R.k, hipLaunchKernelGGL(k_lookup_apply<K>, dim3(grid_for((w1 - w0) - (s1 - s0), 256, 256 * 16)), dim3(256), 0, s, R, D, V, w0, w1, s0, s1,
                                       first_global, xres, ulist, ucount));
}

// One resolution round over the unresolved reads: a read becomes a non-inserter as soon as one of its
// k-mers is finally owned by an earlier read, an inserter when no earlier read even proposes one of them.
// Four reads per wave, 16 positions per read and step (like k_lookup_cand and k_final_pos): the round is a chain of dependent
// loads per read -- list entry, length and slot, packed words, filter word, dictionary slot -- and with one read per wave the
// vector units sat at a fifth of their issue slots (profiles/r4_minimizer_filter.txt); four reads keep four chains in flight.
template <typename K>
__global__ void __launch_bounds__(256) k_check(ReadsDev R, DictDev D, ResolveDev V, uint64_t first_global,
                                              const uint32_t* ulist, const uint32_t* ucount,
                                              uint32_t* next_list, uint32_t* next_count) {
    const uint32_t lane = lane_id(), k = R.k;
    const uint32_t grp = lane >> 4, l = lane & 15, gbase = grp * 16;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t count = *ucount;
    __shared__ uint32_t lists[4][64];                             // per wave: the reads it found blocked, handed over in batches
    uint32_t* my_list = lists[threadIdx.x >> 6];
    uint32_t n_list = 0;                                          // wave-uniform
    auto flush_list = [&]() {
        if (!n_list) return;
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(next_count, n_list);
        at = (uint32_t)__shfl((int)at, 0);
        __builtin_amdgcn_wave_barrier();
        if (lane < n_list) next_list[at + lane] = my_list[lane];
        __builtin_amdgcn_wave_barrier();
        n_list = 0;
    };
    for (uint64_t e0 = 4 * wave; e0 < count; e0 += 4 * nwaves) {
        const uint64_t e = e0 + grp;
        const bool have = e < count;
        const uint32_t i = have ? ulist[e] : 0u;
        const uint64_t g = first_global + i;
        const uint32_t nk = have ? R.len[i] - k + 1 : 0u;
        const uint32_t* pk = R.packed + 2 * (have ? R.slot_off[i] : 0);
        bool anyfin = false, anyblock = false;                    // (the same in the 16 lanes of a quarter-wave)
        uint32_t base = 0;
        bool done = !have;
        while (__any(!done)) {
            const bool run = !done;
            const uint32_t p = base + l;
            const bool valid = run && p < nk;
            const uint32_t words = (run && l < 8) ? pk[(base >> 4) + l] : 0u;
            const K cn = canon_from_words16<K>(words, gbase, base, valid ? p : (run ? nk - 1 : 0), k);
            bool f = false, t = false;
            if (valid) {
                // Every read on this list went through ALL of its k-mers in k_lookup_cand without meeting a key that was final
                // before the window, so only keys PROPOSED in this window (which include the ones made final in it) can matter:
                // the 1 MiB array of those, a few per cent full, answers for all but a handful of the read's ~120 k-mers
                const uint32_t pb = window_bit(cn);
                if ((D.pbits[pb >> 5] >> (pb & 31)) & 1u) {
                    uint64_t fin = IDX_INF;
                    const uint32_t slot = dict_find(D, cn, fin);
                    if (slot != 0xFFFFFFFFu) {
                        f = fin < g;
                        t = *tent_ptr(D, slot) < g;
                    }
                }
            }
            const uint32_t fq = (uint32_t)(__ballot(f) >> gbase) & 0xFFFFu, tq = (uint32_t)(__ballot(t) >> gbase) & 0xFFFFu;
            if (run) {
                anyfin = anyfin || fq != 0;
                anyblock = anyblock || tq != 0;
                base += 16;
                done = anyfin || base >= nk;
            }
        }
        // blocked reads go on to the next round
        const bool blocked = have && !anyfin && anyblock;
        const unsigned long long bm = __ballot(blocked && l == 0);
        if (bm) {
            if (blocked && l == 0) my_list[n_list + (uint32_t)__popcll(bm & ((1ull << lane) - 1))] = i;
            n_list += (uint32_t)__popcll(bm);
            if (n_list > 64 - 4) flush_list();
        }
        const bool inserter = have && !anyfin && !anyblock;
        if (have && l == 0) {
            if (anyfin) V.status[i] = ST_HITNEW;
            else if (inserter) {
                V.status[i] = ST_INSERTER;
                const uint32_t slot = V.cand_slot[i];
                __hip_atomic_store(fin_ptr<K>(D, slot), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t wb = window_bit(dict_key<K>(D, slot));              // k_final_pos only looks up keys whose bit is set
                atomicOr(&D.wbits[wb >> 5], 1u << (wb & 31));
            }
        }
        // ... and k_lookup_cand, from the next window on: the inserters' keys into the filter of the final keys, one read of the wave
        // at a time (the key's minimizer takes the wave: 32 lanes)
        for (unsigned long long im = __ballot(inserter && l == 0); im; im &= im - 1) {
            const uint32_t src = (uint32_t)__builtin_ctzll(im);
            const uint32_t ii = (uint32_t)__shfl((int)i, (int)src);
            const K key = dict_key<K>(D, V.cand_slot[ii]);
            const uint32_t hmin = key_minimizer_wave(D, key, k, lane);
            if (lane == 0) atomicOr((unsigned long long*)D.fbits + filter_word(D, hmin), (unsigned long long)filter_bits(key));
        }
    }
    flush_list();
}
void launch_check(hipStream_t s, ReadsDev R, DictDev D, ResolveDev V, uint64_t first_global, const uint32_t* ulist,
                  const uint32_t* ucount, uint32_t max_count, uint32_t* next_list, uint32_t* next_count) {
    if (!max_count) return;
    DISPATCH_K(R.k, hipLaunchKernelGGL(k_check<K>, dim3(grid_for(max_count, 16, 256 * 16)), dim3(256), 0, s, R, D, V, first_global, ulist,
                                       ucount, next_list, next_count));
}

__global__ void k_reset_tent(DictDev D, ResolveDev V, const uint32_t* list, const uint32_t* count) {
    uint32_t n = *count;
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x)
        *tent_ptr(D, V.cand_slot[list[e]]) = IDX_INF;
}
void launch_reset_tent(hipStream_t s, DictDev D, ResolveDev V, const uint32_t* list, const uint32_t* count, uint32_t max_count) {
    if (!max_count) return;
    hipLaunchKernelGGL(k_reset_tent, dim3(grid_for(max_count, 256)), dim3(256), 0, s, D, V, list, count);
}
__global__ void k_propose(DictDev D, ResolveDev V, uint64_t first_global, const uint32_t* list, const uint32_t* count) {
    uint32_t n = *count;
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t i = list[e];
        atomicMin((unsigned long long*)tent_ptr(D, V.cand_slot[i]), (unsigned long long)(first_global + i));
    }
}
void launch_propose(hipStream_t s, DictDev D, ResolveDev V, uint64_t first_global, const uint32_t* list,
                    const uint32_t* count, uint32_t max_count) {
    if (!max_count) return;
    hipLaunchKernelGGL(k_propose, dim3(grid_for(max_count, 256)), dim3(256), 0, s, D, V, first_global, list, count);
}

// After the window's fixpoint: findExistingAnchor's answer = FIRST position whose k-mer an earlier read owns.
// Reads that hit the old dictionary at position h only re-check positions < h (this window's inserts); four reads
// per wave, 16 positions per step, like k_lookup_cand.
template <typename K>
__global__ void __launch_bounds__(256) k_final_pos(ReadsDev R, DictDev D, ResolveDev V, uint64_t w0, uint64_t w1, uint64_t first_global) {
    const uint32_t lane = lane_id(), k = R.k;
    const uint32_t grp = lane >> 4, l = lane & 15, gbase = grp * 16;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i0 = w0 + 4 * wave; i0 < w1; i0 += 4 * nwaves) {
        const uint64_t i = i0 + grp;
        const bool have = i < w1;
        const uint8_t st = have ? V.status[i] : (uint8_t)ST_NOANCHOR;
        const bool mine = st == ST_HIT || st == ST_HITNEW;
        const uint64_t g = first_global + i;
        const uint32_t limit = !mine ? 0u : (st == ST_HIT ? V.hit_pos[i] : R.len[i] - k + 1);
        const uint32_t* pk = R.packed + 2 * (mine ? R.slot_off[i] : 0);
        uint32_t base = 0;
        bool done = limit == 0;
        while (__any(!done)) {
            const bool run = !done;
            const uint32_t p = base + l;
            const bool valid = run && p < limit;
            const uint32_t words = (run && l < 8) ? pk[(base >> 4) + l] : 0u;
            const K cn = canon_from_words16<K>(words, gbase, base, valid ? p : (run ? limit - 1 : 0), k);
            bool hit = false; uint32_t slot = 0xFFFFFFFFu;
            if (valid) {
                // every position examined here missed the dictionary as it stood before the window, so only a key made
                // final in THIS window can match: a 1 MiB bit filter (L2 resident, a few per cent full) answers
                // "not one of those" for nearly all of them without touching the dictionary
                const uint32_t wb = window_bit(cn);
                if ((D.wbits[wb >> 5] >> (wb & 31)) & 1u) {
                    uint64_t fin = IDX_INF;
                    slot = dict_find(D, cn, fin);
                    hit = slot != 0xFFFFFFFFu && fin < g;
                }
            }
            const unsigned long long bal = __ballot(hit);
            const uint32_t gb = (uint32_t)(bal >> gbase) & 0xFFFFu;
            if (run) {
                if (gb) {
                    const uint32_t f = (uint32_t)__builtin_ctz(gb);
                    const uint32_t hs = __shfl(slot, (int)(gbase + f));
                    if (l == 0) { V.hit_pos[i] = base + f; V.hit_slot[i] = hs; }
                    done = true;
                } else { base += 16; done = base >= limit; }
            }
        }
    }
}
void launch_final_pos(hipStream_t s, ReadsDev R, DictDev D, ResolveDev V, uint64_t w0, uint64_t w1, uint64_t first_global) {
    if (w1 <= w0) return;
    DISPATCH_K(R.k, hipLaunchKernelGGL(k_final_pos<K>, dim3(grid_for(w1 - w0, 16, 256 * 16)), dim3(256), 0, s, R, D, V, w0, w1, first_global));
}

__global__ void k_ins_flags(ResolveDev V, uint64_t w0, uint64_t w1) {
    for (uint64_t i = w0 + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < w1; i += (uint64_t)gridDim.x * blockDim.x)
        V.ins_flag[i - w0] = V.status[i] == ST_INSERTER ? 1u : 0u;
}
void launch_ins_flags(hipStream_t s, ResolveDev V, uint64_t w0, uint64_t w1) {
    if (w1 <= w0) return;
    hipLaunchKernelGGL(k_ins_flags, dim3(grid_for(w1 - w0, 256)), dim3(256), 0, s, V, w0, w1);
}
// addresses in insertion (= read) order: Leon::findAndInsertAnchor's `_anchorAdress++`
template <typename K>
__global__ void k_assign_addr(DictDev D, ResolveDev V, uint64_t w0, uint64_t w1, const uint32_t* rank, uint64_t addr_base,
                              uint64_t* anchor_kmers) {
    for (uint64_t i = w0 + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < w1; i += (uint64_t)gridDim.x * blockDim.x) {
        if (V.status[i] != ST_INSERTER) continue;
        uint64_t a = addr_base + rank[i - w0];
        uint32_t slot = V.cand_slot[i];
        D.addr[slot] = (uint32_t)a;
        store_kmer(anchor_kmers + a * KT<K>::W, dict_key<K>(D, slot));
        V.hit_pos[i] = V.cand_pos[i];
        V.hit_slot[i] = slot;
    }
}
void launch_assign_addr(hipStream_t s, DictDev D, ResolveDev V, uint64_t w0, uint64_t w1, const uint32_t* rank,
                        uint64_t addr_base, uint64_t* anchor_kmers, uint32_t k) {
    if (w1 <= w0) return;
    DISPATCH_K(k, hipLaunchKernelGGL(k_assign_addr<K>, dim3(grid_for(w1 - w0, 256)), dim3(256), 0, s, D, V, w0, w1, rank, addr_base, anchor_kmers));
}
template <typename K>
__global__ void k_finalize_reads(ReadsDev R, DictDev D, ResolveDev V, uint64_t w0, uint64_t w1) {
    uint32_t k = R.k;
    for (uint64_t i = w0 + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < w1; i += (uint64_t)gridDim.x * blockDim.x) {
        uint8_t st = V.status[i];
        if (st == ST_NOANCHOR) {
            V.anchor_pos[i] = -1; V.anchor_addr[i] = 0; V.flags[i] = 0; V.sort_key[i] = 1ull << 32;   // after every anchored read
            continue;
        }
        uint32_t pos = V.hit_pos[i];
        uint32_t addr = D.addr[V.hit_slot[i]];
        const uint32_t* pk = R.packed + 2 * R.slot_off[i];
        const K km = kmer_at<K>(pk, pos, k);
        uint32_t rev = revcomp(km, k) < km ? 1u : 0u;      // anchor != min(anchor, revcomp(anchor))
        V.anchor_pos[i] = (int32_t)pos; V.anchor_addr[i] = addr;
        V.flags[i] = (uint8_t)(rev | (st == ST_INSERTER ? 2u : 0u));
        V.sort_key[i] = addr;                              // both strands of an anchor walk together (k_walk)
    }
}
void launch_finalize_reads(hipStream_t s, ReadsDev R, DictDev D, ResolveDev V, uint64_t w0, uint64_t w1) {
    if (w1 <= w0) return;
    DISPATCH_K(R.k, hipLaunchKernelGGL(k_finalize_reads<K>, dim3(grid_for(w1 - w0, 256)), dim3(256), 0, s, R, D, V, w0, w1));
}

// ================================================================================================
// The exact sequential pass behind the rounds (round 5).  The rounds above settle a window in as many launches as its longest
// chain of reads each waiting for the one before it: 3-6 for reads placed at random, but reads in genome-position order
// (a sorted BAM turned back into FASTQ, tiled amplicons) make ONE chain of the whole window -- read j + 1 contains the k-mer
// read j proposes -- and a window of 2^21 reads would need ~3 * 10^5 rounds.  What the rounds leave is therefore handed, in read
// order, to ONE wave that walks it like Leon's single thread does, with everything that can be done ahead done ahead and in parallel:
//   k_chain_prep  (parallel, like a round of k_check): per unresolved read r, `dead` -- one of its k-mers is ALREADY final with an
//                 earlier owner -- else its ENTRIES: for every k-mer of r that an earlier unresolved read proposes, the chain index
//                 of that key's FIRST proposer (the key's name in this pass), and `own`: the name of r's own candidate;
//   k_chain_seq   (one workgroup): for r in read order: r inserts <=> not dead and no entry names a key inserted so far.  A bit per key
//                 name in LDS; 64 reads per step, lane per read: entries that name keys of earlier steps are tested against the bits,
//                 entries that name keys of THIS step become a 64-bit mask of the lanes the read waits for, and the step's own order
//                 is settled with ballots (as many iterations as the longest chain inside the 64 reads).  Three producer waves
//                 stage the steps' records from global memory into an LDS ring ahead of the consumer wave;
//   k_chain_apply (parallel): what k_check does for the reads it settles (status, fin, the filters' bits).
// A key's name is the chain index of its first proposer in the chunk, so at most 2^CHAIN_LOG2 reads go through one k_chain_seq;
// a longer list goes chunk by chunk (tent re-proposed by what is left).  The result is the file-order result by construction:
// `inserted` is only ever consulted for keys, and set by reads, in read order.
// ================================================================================================
__global__ void k_chain_flags(const uint8_t* status, uint64_t w0, uint64_t w1, uint32_t* flag) {
    for (uint64_t i = w0 + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < w1; i += (uint64_t)gridDim.x * blockDim.x)
        flag[i - w0] = status[i] == ST_UNRESOLVED ? 1u : 0u;
}
__global__ void k_chain_compact(const uint8_t* status, uint64_t w0, uint64_t w1, const uint32_t* rank, uint32_t* list) {
    for (uint64_t i = w0 + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < w1; i += (uint64_t)gridDim.x * blockDim.x)
        if (status[i] == ST_UNRESOLVED) list[rank[i - w0]] = (uint32_t)i;
}
void launch_chain_flags(hipStream_t s, ResolveDev V, uint64_t w0, uint64_t w1, uint32_t* flag) {
    if (w1 <= w0) return;
    hipLaunchKernelGGL(k_chain_flags, dim3(grid_for(w1 - w0, 256)), dim3(256), 0, s, V.status, w0, w1, flag);
}
void launch_chain_compact(hipStream_t s, ResolveDev V, uint64_t w0, uint64_t w1, const uint32_t* rank, uint32_t* list) {
    if (w1 <= w0) return;
    hipLaunchKernelGGL(k_chain_compact, dim3(grid_for(w1 - w0, 256)), dim3(256), 0, s, V.status, w0, w1, rank, list);
}
// tent of the candidates of list[0 .. n): cleared, then proposed again (a later chunk of the chain: what is left proposes)
__global__ void k_chain_reset(DictDev D, ResolveDev V, const uint32_t* list, uint32_t n) {
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x)
        *tent_ptr(D, V.cand_slot[list[e]]) = IDX_INF;
}
__global__ void k_chain_propose(DictDev D, ResolveDev V, uint64_t first_global, const uint32_t* list, uint32_t n) {
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t i = list[e];
        atomicMin((unsigned long long*)tent_ptr(D, V.cand_slot[i]), (unsigned long long)(first_global + i));
    }
}
void launch_chain_repropose(hipStream_t s, DictDev D, ResolveDev V, uint64_t first_global, const uint32_t* reset_list, uint32_t n_reset,
                            const uint32_t* list, uint32_t n) {
    if (n_reset) hipLaunchKernelGGL(k_chain_reset, dim3(grid_for(n_reset, 256)), dim3(256), 0, s, D, V, reset_list, n_reset);
    if (n) hipLaunchKernelGGL(k_chain_propose, dim3(grid_for(n, 256)), dim3(256), 0, s, D, V, first_global, list, n);
}

// list: the chunk's reads in read order; rank: the window's exclusive ranks of the unresolved reads (a read's chain index), c0 the
// chunk's first chain index.  A STEP of k_chain_seq is 64 consecutive reads of the chunk; read e sits in lane e % 64 of step e / 64.
// FILL = false: own[e], and cnt[e] = dead << 31 | the number of its entries that name keys of EARLIER steps;
// FILL = true (after k_chain_tables): those entries, transposed per step -- entry j of read e at ent[(gbase[e / 64] + j) * 64 + e % 64],
//   so that a wave takes a step's entries row by row -- and dep[e]: the lanes of its own step the read waits for (every earlier
//   lane that proposes one of its k-mers: through `om` for keys first proposed in the step, through the step's late lanes for the others).
// Four reads per wave, 16 positions per read and step, like k_check.
template <typename K, bool FILL>
__global__ void __launch_bounds__(256) k_chain_prep(ReadsDev R, DictDev D, ResolveDev V, uint64_t first_global, uint64_t w0,
                                                   const uint32_t* list, uint32_t n, uint32_t c0, const uint32_t* rank,
                                                   uint32_t* cnt, uint32_t* own, const uint64_t* gbase, uint32_t* ent,
                                                   const unsigned long long* om, const unsigned long long* late, unsigned long long* dep_out,
                                                   unsigned long long* xdep_out) {
    const uint32_t lane = lane_id(), k = R.k;
    const uint32_t grp = lane >> 4, l = lane & 15, gbase16 = grp * 16;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t e0 = 4 * wave; e0 < n; e0 += 4 * nwaves) {
        const uint64_t e = e0 + grp;
        const bool have = e < n;
        const uint32_t i = have ? list[e] : 0u;
        const uint64_t g = first_global + i;
        const uint32_t nk = have ? R.len[i] - k + 1 : 0u;
        const uint32_t* pk = R.packed + 2 * (have ? R.slot_off[i] : 0);
        const uint32_t g0 = (uint32_t)e & ~63u, le = (uint32_t)e & 63u;       // the read's step and its lane in it
        const uint32_t p0 = g0 >= 64 ? g0 - 64 : 0u;                          // the step before it (none for the chunk's first step: p0 == g0)
        uint32_t total = 0;
        bool dead = false;
        bool done = !have;
        uint64_t at = 0;
        unsigned long long dep = 0ull, xdep = 0ull, late_below = 0ull, late_prev = 0ull;
        if (FILL) {
            done = !have || (cnt[e] >> 31);                                  // nothing to do for a read that is dead already
            at = have ? gbase[e >> 6] * 64 + le : 0;
            late_below = have ? late[e >> 6] & ((1ull << le) - 1ull) : 0ull;
            late_prev = have && g0 >= 64 ? late[(e >> 6) - 1] : 0ull;
        }
        uint32_t base = 0;
        while (__any(!done)) {
            const bool run = !done;
            const uint32_t p = base + l;
            const bool valid = run && p < nk;
            const uint32_t words = (run && l < 8) ? pk[(base >> 4) + l] : 0u;
            const K cn = canon_from_words16<K>(words, gbase16, base, valid ? p : (run ? nk - 1 : 0), k);
            bool f = false, t = false;
            uint32_t kid = 0;
            if (valid) {
                const uint32_t pb = window_bit(cn);                        // (keys proposed in this window, final or not: k_check's filter)
                if ((D.pbits[pb >> 5] >> (pb & 31)) & 1u) {
                    uint64_t fin = IDX_INF;
                    const uint32_t slot = dict_find(D, cn, fin);
                    if (slot != 0xFFFFFFFFu) {
                        f = fin < g;
                        const uint64_t tv = *tent_ptr(D, slot);
                        t = tv < g;
                        if (t) kid = rank[tv - first_global - w0] - c0;     // the key's name: the chain index of its first proposer
                    }
                }
            }
            const bool outer = t && kid < p0;                               // names a key older than the step before: an entry
            const uint32_t fq = (uint32_t)(__ballot(f) >> gbase16) & 0xFFFFu, oq = (uint32_t)(__ballot(outer) >> gbase16) & 0xFFFFu;
            if (FILL) {
                if (outer) ent[at + 64ull * (total + (uint32_t)__popc(oq & ((1u << l) - 1u)))] = kid;
                if (t && kid >= g0) dep |= om[kid];                          // first proposed in this step: whoever proposes it here
                else if (t) {
                    if (kid >= p0) xdep |= om[kid];                          // first proposed in the step before: whoever proposes it there ...
                    for (unsigned long long m = late_prev; m; m &= m - 1) {  // ... or an older key that a lane of the step before proposes too (rare)
                        const uint32_t a = (uint32_t)__builtin_ctzll(m);
                        if (own[p0 + a] == kid) xdep |= 1ull << a;
                    }
                    for (unsigned long long m = late_below; m; m &= m - 1) { // an older key that an earlier lane of THIS step proposes too (rare)
                        const uint32_t a = (uint32_t)__builtin_ctzll(m);
                        if (own[g0 + a] == kid) dep |= 1ull << a;
                    }
                }
            } else dead = dead || fq != 0;
            if (run) {
                total += (uint32_t)__popc(oq);
                base += 16;
                done = dead || base >= nk;
            }
        }
        if (FILL) {
            for (int d = 1; d < 16; d <<= 1) { dep |= (unsigned long long)__shfl_xor((long long)dep, d); xdep |= (unsigned long long)__shfl_xor((long long)xdep, d); }
            if (have && l == 0) { const bool dd = (cnt[e] >> 31) != 0; dep_out[e] = dd ? 0ull : dep & ((1ull << le) - 1ull); xdep_out[e] = dd ? 0ull : xdep; }
        } else if (have && l == 0) {
            cnt[e] = dead ? 0x80000000u : total;
            own[e] = rank[*tent_ptr(D, V.cand_slot[i]) - first_global - w0] - c0;     // (the read proposes its candidate itself: tent <= g)
        }
    }
}
void launch_chain_prep(hipStream_t s, bool fill, ReadsDev R, DictDev D, ResolveDev V, uint64_t first_global, uint64_t w0, const uint32_t* list,
                       uint32_t n, uint32_t c0, const uint32_t* rank, uint32_t* cnt, uint32_t* own, const uint64_t* gbase, uint32_t* ent,
                       const unsigned long long* om, const unsigned long long* late, unsigned long long* dep, unsigned long long* xdep) {
    if (!n) return;
    if (fill) DISPATCH_K(R.k, hipLaunchKernelGGL((k_chain_prep<K, true>), dim3(grid_for(n, 16, 256 * 16)), dim3(256), 0, s, R, D, V, first_global, w0, list, n, c0, rank, cnt, own, gbase, ent, om, late, dep, xdep));
    else DISPATCH_K(R.k, hipLaunchKernelGGL((k_chain_prep<K, false>), dim3(grid_for(n, 16, 256 * 16)), dim3(256), 0, s, R, D, V, first_global, w0, list, n, c0, rank, cnt, own, gbase, ent, om, late, dep, xdep));
}
// per step of 64 reads, a wave: the rows its entries take (the longest list among its reads; a 64-bit count for the scan), om[step][x] =
// the lanes that propose the key first proposed by lane x of the step, late[step] = the lanes (not dead already) whose key was first
// proposed in an EARLIER step
__global__ void __launch_bounds__(256) k_chain_tables(const uint32_t* cnt, const uint32_t* own, uint32_t n, uint64_t* rows, unsigned long long* om, unsigned long long* late) {
    __shared__ unsigned long long tab[4][64];
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t nG = ((uint64_t)n + 63) / 64;
    for (uint64_t G = wave; G <= nG; G += nwaves) {
        const uint64_t e = G * 64 + lane;
        const bool have = G < nG && e < n;
        const uint32_t cd = have ? cnt[e] : 0x80000000u, o = have ? own[e] : 0u;
        const uint32_t g0 = (uint32_t)G * 64;
        uint32_t c = (cd >> 31) ? 0u : cd;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t x = (uint32_t)__shfl_xor((int)c, d); c = x > c ? x : c; }
        if (lane == 0) rows[G] = c;                                   // (rows[nG] = 0: the scan's total lands there)
        if (G < nG) {
            tab[wv][lane] = 0ull;
            __builtin_amdgcn_wave_barrier();
            if (have && o >= g0) atomicOr(&tab[wv][o - g0], 1ull << lane);
            __builtin_amdgcn_wave_barrier();
            om[G * 64 + lane] = tab[wv][lane];
            const unsigned long long lm = __ballot(have && !(cd >> 31) && o < g0);
            if (lane == 0) late[G] = lm;
        }
    }
}
void launch_chain_tables(hipStream_t s, const uint32_t* cnt, const uint32_t* own, uint32_t n, uint64_t* rows, unsigned long long* om, unsigned long long* late) {
    hipLaunchKernelGGL(k_chain_tables, dim3(grid_for((n + 63) / 64 + 1, 4)), dim3(256), 0, s, cnt, own, n, rows, om, late);
}

struct ChainSlot { uint32_t own[64], cnt[64], dep_lo[64], dep_hi[64], xdep_lo[64], xdep_hi[64], hit[64], ent[CHAIN_EL][64]; uint64_t gb; uint32_t rows, pad; };
constexpr uint32_t CHAIN_BITS_WORDS = 1u << (CHAIN_LOG2 - 5);
size_t chain_seq_lds_bytes() { return CHAIN_BITS_WORDS * 4 + CHAIN_DEPTH * sizeof(ChainSlot) + (2 * CHAIN_DEPTH + 2) * 4; }
// One workgroup, a step (64 reads) at a time through a ring of CHAIN_DEPTH slots in LDS:
//   waves 2, 3, 6, 7  PRODUCERS  step G's records from global memory into slot G % CHAIN_DEPTH (every fourth step each), loads in flight while they wait for the slot;
//   wave 1      TESTER      step G's entries -- keys older than step G - 1 -- against the bits, as soon as step G - 2 is settled: a 64-bit hit mask;
//   wave 0      SETTLER     step G: dead = dead already | hit | waits for an inserter of step G - 1 (xdep & that step's inserters); then the step's own
//                           order by ballots over `dep`; the inserters' bits; ins[].
// The tester works on step G while the settler is on step G - 1: a lone wave issues an instruction every ~8 cycles whatever it is, so what
// bounds the pass is the instruction count of its slowest wave, and the two halves of a step now run side by side.
// Flags are LDS words written with relaxed stores behind a compiler barrier: LDS operations of a wave execute in order and the LDS is one
// memory for the workgroup (an atomic release would also wait for every global access in flight).  Every wait is on a wave of the same
// workgroup that does not wait for the waiter: producers wait for the settler (slot reuse), the tester for producers and the settler
// (two steps back), the settler for the tester.
template <bool TRACE>
__global__ void __launch_bounds__(512) k_chain_seq(uint32_t n, const uint32_t* cnt, const uint32_t* own, const unsigned long long* depv, const unsigned long long* xdepv,
                                                  const uint64_t* gbase /* steps + 1 */, const uint32_t* ent, unsigned long long* insmask /* per step: its inserters */,
                                                  unsigned long long* trace /* nullptr or 8 counters */) {
    extern __shared__ uint32_t chain_lds[];
    uint32_t* bits = chain_lds;                                                  // a bit per key name: inserted so far
    ChainSlot* ring = reinterpret_cast<ChainSlot*>(chain_lds + CHAIN_BITS_WORDS);
    uint32_t* ready = reinterpret_cast<uint32_t*>(ring + CHAIN_DEPTH);           // [CHAIN_DEPTH]: step + 1 once the slot holds it
    uint32_t* tested = ready + CHAIN_DEPTH;                                      // [CHAIN_DEPTH]: step + 1 once its hit mask is in the slot
    uint32_t* settled = tested + CHAIN_DEPTH;                                    // steps the settler is done with
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t nG = (n + 63) / 64;
    for (uint32_t w = threadIdx.x; w < CHAIN_BITS_WORDS; w += blockDim.x) bits[w] = 0u;
    if (threadIdx.x < 2 * CHAIN_DEPTH) ready[threadIdx.x] = 0u;
    if (threadIdx.x == 0) *settled = 0u;
    __syncthreads();
    auto flag = [](uint32_t* f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto raise = [](uint32_t* f, uint32_t v) { __atomic_signal_fence(__ATOMIC_SEQ_CST); __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    if (wv >= 2) {
        // ---- producers: waves 2, 3, 6, 7 (two on each of the SIMDs the tester and the settler do not use; waves 4 and 5 would share
        // theirs and leave at once).  A producer's step costs it a global round trip (~2 us): four of them keep ahead of a settler
        // that takes ~0.5 us per step; two did not (the settler waited 400-800 cycles per step).
        if (wv == 4 || wv == 5) return;
        for (uint32_t G = wv < 4 ? wv - 2 : wv - 4; G < nG; G += 4) {
            const uint32_t e = G * 64 + lane;
            const uint64_t gb = gbase[G];
            const uint32_t rows = (uint32_t)(gbase[G + 1] - gb);
            const uint32_t o = e < n ? own[e] : 0u, cd = e < n ? cnt[e] : 0x80000000u;
            const unsigned long long dp = e < n ? depv[e] : 0ull, xd = e < n ? xdepv[e] : 0ull;
            // the entry rows in passes of 32: the first pass's loads are in flight while the slot is waited for
            constexpr uint32_t PASS = 32;
            uint32_t v[PASS];
#pragma unroll
            for (uint32_t j = 0; j < PASS; j++) v[j] = j < rows ? ent[(gb + j) * 64 + lane] : 0u;
            while (G >= flag(settled) + CHAIN_DEPTH) __builtin_amdgcn_s_sleep(2);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            ChainSlot& S = ring[G % CHAIN_DEPTH];
            S.own[lane] = o; S.cnt[lane] = cd; S.dep_lo[lane] = (uint32_t)dp; S.dep_hi[lane] = (uint32_t)(dp >> 32);
            S.xdep_lo[lane] = (uint32_t)xd; S.xdep_hi[lane] = (uint32_t)(xd >> 32);
            if (lane == 0) { S.gb = gb; S.rows = rows; }
#pragma unroll
            for (uint32_t j = 0; j < PASS; j++) if (j < rows) S.ent[j][lane] = v[j];
            for (uint32_t p0 = PASS; p0 < rows && p0 < CHAIN_EL; p0 += PASS) {
#pragma unroll
                for (uint32_t j = 0; j < PASS; j++) v[j] = p0 + j < rows ? ent[(gb + p0 + j) * 64 + lane] : 0u;
#pragma unroll
                for (uint32_t j = 0; j < PASS; j++) if (p0 + j < rows) S.ent[p0 + j][lane] = v[j];
            }
            raise(&ready[G % CHAIN_DEPTH], G + 1);
        }
        return;
    }
    if (wv == 1) {
        // ---- the tester: a step's entries name keys older than the step before it; whoever inserts one of those does so in a step that is
        // settled by now, or is a lane of the step before (then the settler sees it through xdep)
        for (uint32_t G = 0; G < nG; G++) {
            while (flag(&ready[G % CHAIN_DEPTH]) != G + 1) __builtin_amdgcn_s_sleep(0);
            while (flag(settled) + 1 < G) __builtin_amdgcn_s_sleep(0);              // steps 0 .. G - 2 settled
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            ChainSlot& S = ring[G % CHAIN_DEPTH];
            const uint64_t gb = S.gb;
            const uint32_t rows = S.rows;
            const uint32_t cd = S.cnt[lane];
            const uint32_t c = (cd >> 31) ? 0u : cd;
            uint32_t hitw = 0;
            const uint32_t rows_lds = rows < CHAIN_EL ? rows : CHAIN_EL;
            for (uint32_t j0 = 0; j0 < rows_lds; j0 += 8) {
                // (every load unconditional and its address independent of the others: the eight bit words are in flight together.
                // Rows past a lane's own count hold whatever the buffer held: masked into range here, ignored below.)
                uint32_t kid[8], w[8];
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) kid[u] = S.ent[(j0 + u) % CHAIN_EL][lane] & ((1u << CHAIN_LOG2) - 1u);
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) w[u] = bits[kid[u] >> 5];
                uint32_t hb = 0;
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) hb |= ((w[u] >> (kid[u] & 31)) & 1u) << u;
                const uint32_t nv = c > j0 ? c - j0 : 0u;                            // this lane's rows in the eight
                hitw |= hb & (nv >= 8 ? 0xFFu : (1u << nv) - 1u);
            }
            for (uint32_t j0 = CHAIN_EL; j0 < rows; j0 += 8) {                       // (lists longer than the ring's rows: the rest from global memory, eight loads in flight)
                uint32_t kid[8], w[8];
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) kid[u] = (j0 + u < rows ? ent[(gb + j0 + u) * 64 + lane] : 0u) & ((1u << CHAIN_LOG2) - 1u);
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) w[u] = bits[kid[u] >> 5];
                uint32_t hb = 0;
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) hb |= ((w[u] >> (kid[u] & 31)) & 1u) << u;
                const uint32_t nv = c > j0 ? c - j0 : 0u;
                hitw |= hb & (nv >= 8 ? 0xFFu : (1u << nv) - 1u);
            }
            S.hit[lane] = hitw;
            raise(&tested[G % CHAIN_DEPTH], G + 1);
        }
        return;
    }
    // ---- the settler: steps in order
    unsigned long long tr_iter = 0, tr_rows = 0, tr_ins = 0, tr_wait = 0, tr_res = 0;
    unsigned long long ins_prev = 0ull;                                          // the inserters of the step before
    for (uint32_t G = 0; G < nG; G++) {
        const unsigned long long t_a = TRACE ? __builtin_amdgcn_s_memtime() : 0ull;
        // the slot's words are read in the same breath as its flag -- LDS operations execute in order, so what follows a flag read that
        // saw G + 1 is the step's data -- one LDS round trip per step instead of two when the tester is ahead (it nearly always is)
        const ChainSlot& S = ring[G % CHAIN_DEPTH];
        uint32_t o, cd, dlo, dhi, xlo, xhi, hitf;
        for (;;) {
            const uint32_t f = flag(&tested[G % CHAIN_DEPTH]);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            o = S.own[lane]; cd = S.cnt[lane]; dlo = S.dep_lo[lane]; dhi = S.dep_hi[lane]; xlo = S.xdep_lo[lane]; xhi = S.xdep_hi[lane]; hitf = S.hit[lane];
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            if (f == G + 1) break;
            __builtin_amdgcn_s_sleep(0);
        }
        const unsigned long long t_b = TRACE ? __builtin_amdgcn_s_memtime() : 0ull;
        const unsigned long long dep = ((unsigned long long)dhi << 32) | dlo;
        const unsigned long long xdep = ((unsigned long long)xhi << 32) | xlo;
        const bool dead = (cd >> 31) != 0 || hitf != 0 || (xdep & ins_prev) != 0ull;    // (lanes past the list's end arrive dead)
        // the step's own order: a lane is settled once every lane it waits for is; it inserts iff none of those did.  The masks are wave-wide
        // scalars: per iteration two ANDs of `dep` with a scalar and two compares in the vector unit, the rest on the scalar unit.
        unsigned long long decided = __ballot(dead || dep == 0ull), insm = __ballot(!dead && dep == 0ull);
        while (~decided) {
            const unsigned long long nd = ~decided;
            const unsigned long long blocked = __ballot((dep & nd) != 0ull);      // still waits for an unsettled lane
            const unsigned long long killed = __ballot((dep & insm) != 0ull);     // waits for a lane that inserted
            insm |= nd & ~blocked & ~killed;
            decided |= nd & (~blocked | killed);
            if (TRACE) tr_iter++;
        }
        if ((insm >> lane) & 1ull) atomicOr(&bits[o >> 5], 1u << (o & 31));
        if (lane == 0) insmask[G] = insm;                                        // (one 8-byte store per step; k_chain_apply takes a read's bit)
        ins_prev = insm;
        raise(settled, G + 1);
        if (TRACE) {
            const unsigned long long t_d = __builtin_amdgcn_s_memtime();
            tr_ins += (unsigned long long)__popcll(insm); tr_rows += S.rows;
            tr_wait += t_b - t_a; tr_res += t_d - t_b;
        }
    }
    if (TRACE && lane == 0) { atomicAdd(trace + 0, (unsigned long long)nG); atomicAdd(trace + 1, tr_iter); atomicAdd(trace + 2, tr_rows); atomicAdd(trace + 3, tr_ins);
                              atomicAdd(trace + 4, tr_wait); atomicAdd(trace + 6, tr_res); }
}
int launch_chain_seq(hipStream_t s, uint32_t n, const uint32_t* cnt, const uint32_t* own, const unsigned long long* dep, const unsigned long long* xdep,
                     const uint64_t* gbase, const uint32_t* ent, unsigned long long* ins, unsigned long long* trace) {
    if (!n) return 0;
    if (n > (1u << CHAIN_LOG2)) return 1;
    const void* fn = trace ? reinterpret_cast<const void*>(k_chain_seq<true>) : reinterpret_cast<const void*>(k_chain_seq<false>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_seq_lds_bytes()) != hipSuccess) return 2;
    if (trace) hipLaunchKernelGGL(k_chain_seq<true>, dim3(1), dim3(512), chain_seq_lds_bytes(), s, n, cnt, own, dep, xdep, gbase, ent, ins, trace);
    else hipLaunchKernelGGL(k_chain_seq<false>, dim3(1), dim3(512), chain_seq_lds_bytes(), s, n, cnt, own, dep, xdep, gbase, ent, ins, trace);
    return 0;
}
// what k_check does for the reads it settles: status; an inserter's fin, its bit in the window's filter, its key in the final keys' filter
template <typename K>
__global__ void __launch_bounds__(256) k_chain_apply(DictDev D, ResolveDev V, uint64_t first_global, const uint32_t* list, uint32_t n, const unsigned long long* ins, uint32_t k) {
    const uint32_t lane = lane_id();
    for (uint64_t e0 = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) & ~63ull; e0 < n; e0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t e = e0 + lane;
        const bool have = e < n;
        const uint32_t i = have ? list[e] : 0u;
        const bool inserter = have && ((ins[e >> 6] >> (e & 63)) & 1ull) != 0;
        if (have) V.status[i] = inserter ? ST_INSERTER : ST_HITNEW;
        if (inserter) {
            const uint32_t slot = V.cand_slot[i];
            __hip_atomic_store(fin_ptr<K>(D, slot), first_global + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t wb = window_bit(dict_key<K>(D, slot));
            atomicOr(&D.wbits[wb >> 5], 1u << (wb & 31));
        }
        for (unsigned long long im = __ballot(inserter); im; im &= im - 1) {
            const uint32_t src = (uint32_t)__builtin_ctzll(im);
            const uint32_t ii = (uint32_t)__shfl((int)i, (int)src);
            const K key = dict_key<K>(D, V.cand_slot[ii]);
            const uint32_t hmin = key_minimizer_wave(D, key, k, lane);
            if (lane == 0) atomicOr((unsigned long long*)D.fbits + filter_word(D, hmin), (unsigned long long)filter_bits(key));
        }
    }
}
void launch_chain_apply(hipStream_t s, DictDev D, ResolveDev V, uint64_t first_global, const uint32_t* list, uint32_t n, const unsigned long long* ins, uint32_t k) {
    if (!n) return;
    DISPATCH_K(k, hipLaunchKernelGGL(k_chain_apply<K>, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, D, V, first_global, list, n, ins, k));
}

// the dictionary stream's symbols: k bases per anchor, first base first, on a 5-symbol Order0Model (_anchorDictModel)
template <typename K>
__global__ void k_anchor_symbols(const uint64_t* kmers, uint64_t n_anchors, uint32_t k, uint8_t* syms) {
    const uint64_t total = n_anchors * k;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t a = i / k; const uint32_t j = (uint32_t)(i % k);
        syms[2 * i] = (uint8_t)M_NOANCHOR_READ;
        syms[2 * i + 1] = (uint8_t)((uint64_t)(load_kmer<K>(kmers + a * KT<K>::W) >> (2 * (k - 1 - j))) & 3u);
    }
}
void launch_anchor_symbols(hipStream_t s, const uint64_t* kmers, uint64_t n_anchors, uint32_t k, uint8_t* syms) {
    if (!n_anchors) return;
    DISPATCH_K(k, hipLaunchKernelGGL(k_anchor_symbols<K>, dim3(grid_for(n_anchors * k, 256, 8192)), dim3(256), 0, s, kmers, n_anchors, k, syms));
}

// ================================================================================================
// walk: DnaEncoder::encodeAnchorRead's two loops over buildBifurcationList.  One lane per read, reads taken
// in anchor-sorted order so that neighbouring lanes probe the same bloom windows at the same step.
// ================================================================================================
// One extension step.  Every lane extends to the RIGHT on its own strand: a left walk over the read is a right walk
// over the reverse complement (x = revcomp of the current k-mer, y = the k-mer), with the read base complemented.
// The bloom is strand-symmetric, so contains4_left(kmer)[n] == contains4_right(revcomp(kmer))[n ^ 2]: the left-walking
// lanes only swap the mask's bit pairs back.  One instruction stream for both directions, no duplicated arithmetic.
template <typename K>
__device__ inline void walk_apply(uint32_t res4, uint32_t k, K kmask_k, K& x, K& y, uint32_t nt, bool left, uint8_t* ev_pos) {
    if (left) res4 = ((res4 >> 2) & 3u) | ((res4 & 3u) << 2);         // back to the read strand's base codes
    const uint32_t cnt = __popc(res4);
    const bool solid = (res4 >> nt) & 1u;
    const uint32_t first = res4 ? (uint32_t)__builtin_ctz(res4) : 0u;
    uint32_t follow = nt;
    if (solid) {
        if (cnt == 2) *ev_pos = (uint8_t)(first == nt ? EV_BIN0 : EV_BIN1);
        else if (cnt > 2) *ev_pos = (uint8_t)(EV_NT0 + nt);
    } else {
        if (cnt >= 1) { *ev_pos = (uint8_t)((EV_NT0 + nt) | EV_ERROR); follow = first; }
        else *ev_pos = (uint8_t)(EV_NT0 + nt);
    }
    // AbstractDnaCoder::codeSeedBin on this lane's strand, keeping the other strand alongside
    const uint32_t f = left ? follow ^ 2u : follow;
    x = ((x << 2) | (K)f) & kmask_k;
    y = (y >> 2) | ((K)(f ^ 2u) << (2 * (k - 1)));
}

// ---- the walk's path cache (round 5) ---------------------------------------------------------------------------------------------
// Every genome position is walked by ~5.6 anchor groups (an anchor every ~35 bases, walks of ~100 steps each way), ~2.8 of them in
// each direction, and each of them asks the bloom the same question at the same oriented k-mer: "which successors are solid?".
// Where the answer is "exactly one, b" the walk's next k-mer is x.b whatever the read holds (a read base other than b is a
// sequencing error that FOLLOWS b: walk_step), so a run of such steps is a fact about the bloom alone.  The first walker through
// a region leaves those facts behind: at every HOP POINT -- an oriented k-mer whose hash ends in four zero bits, the same for every
// walker -- it starts recording the unique successors of the steps that follow (up to 28, to the next hop point, or to the first
// step that is not unique) and publishes them under the hop point's k-mer; a later walker that reaches the hop point takes the run
// and makes those steps without a probe.  Write-once slots in 64-byte buckets: a key is claimed by compare-and-swap, its payload
// (count << 56 | bases, first base lowest) only grows (atomic max: two runs from one k-mer are prefixes of one another), readers
// take a key without a payload, a bucket that is full or a line that is stale in their L1 as a miss.  The events are the probes'
// own, byte for byte: a cached step hands walk_apply the mask the probe would have returned.
constexpr uint64_t WC_EMPTY = ~0ull;
constexpr uint32_t WC_MAX = 28;
template <typename K> struct WCL;
template <> struct WCL<uint64_t> { static constexpr uint32_t SLOTS = 4, WORDS = 2, PAY = 1; };     // {key, payload}
template <> struct WCL<u128> { static constexpr uint32_t SLOTS = 2, WORDS = 4, PAY = 2; };         // {lo, hi, payload, -}
template <typename K> __device__ inline bool wc_is_hop(const WalkCache& C, K x) { return ((fold32(x) * 0x9E3779B1u) >> C.hop_shift) == 0u; }
template <typename K> __device__ inline uint64_t wc_lookup(const WalkCache& C, K x) {
    const uint64_t* b = C.slots + (key_hash(x) & C.bucket_mask) * 8;
    const uint64_t lo = (uint64_t)x, hi = KT<K>::W == 2 ? (uint64_t)(x >> (KT<K>::W == 2 ? 64 : 0)) : 0;
    uint64_t pay = 0;
#pragma unroll
    for (uint32_t i = 0; i < WCL<K>::SLOTS; i++) {
        const uint64_t* sp = b + i * WCL<K>::WORDS;
        const bool same = sp[0] == lo && (KT<K>::W == 1 || sp[1] == hi);
        const uint64_t pv = sp[WCL<K>::PAY];
        if (same) pay = pv;
    }
    return lo == WC_EMPTY ? 0ull : pay;
}
template <typename K> __device__ inline void wc_insert(const WalkCache& C, K x, uint64_t pay) {
    uint64_t* b = C.slots + (key_hash(x) & C.bucket_mask) * 8;
    const uint64_t lo = (uint64_t)x, hi = KT<K>::W == 2 ? (uint64_t)(x >> (KT<K>::W == 2 ? 64 : 0)) : 0;
    if (lo == WC_EMPTY) return;
    for (uint32_t i = 0; i < WCL<K>::SLOTS; i++) {
        uint64_t* sp = b + i * WCL<K>::WORDS;
        unsigned long long expect = WC_EMPTY;
        const bool mine = __hip_atomic_compare_exchange_strong((unsigned long long*)sp, &expect, (unsigned long long)lo, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mine) {
            if (KT<K>::W == 2) __hip_atomic_store(sp + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_max(sp + WCL<K>::PAY, pay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (expect == lo) {
            if (KT<K>::W == 1) { (void)__hip_atomic_fetch_max(sp + WCL<K>::PAY, pay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
            const uint64_t h = __hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (h == hi) { (void)__hip_atomic_fetch_max(sp + WCL<K>::PAY, pay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
            if (h == WC_EMPTY) return;                               // its writer is between its two stores: let it be
        }
    }
}
// one side's state, packed (the kernel's registers decide how many waves a SIMD holds): st = run | rec_n << 8 | rec_on << 16 | rec_new << 17 --
// `run` steps left of a run taken from the cache, whose bases are rec_b's own (base i of the run at bits 2i: the next one is base
// rec_n - run); rec_n bases recorded under rec_key since the last hop point (rec_new: more of them than the cache holds)
template <typename K> struct WalkRun { uint32_t st = 0; uint64_t rec_b = 0; K rec_key = 0; };
constexpr uint32_t WR_ON = 1u << 16, WR_NEW = 1u << 17;
// the lanes of an anchor group walk in lockstep and publish the same run at the same moment: only the first of a row of equal
// neighbours goes to the slot (same-address atomics serialise in L2)
template <typename K> __device__ inline void wc_publish(const WalkCache& C, WalkRun<K>& W) {
    const uint64_t pay = ((uint64_t)((W.st >> 8) & 31u) << 56) | W.rec_b, klo = (uint64_t)W.rec_key;
    const unsigned long long m = __ballot(true);
    const uint32_t lane = lane_id();
    const uint64_t up_k = (uint64_t)__shfl_up((long long)klo, 1), up_p = (uint64_t)__shfl_up((long long)pay, 1);
    const bool dup = lane > 0 && ((m >> (lane - 1)) & 1ull) && up_k == klo && up_p == pay;
    if (!dup) wc_insert<K>(C, W.rec_key, pay);
    W.st &= ~WR_NEW;
}
__global__ void k_wc_init(uint64_t* slots, uint64_t n_words, uint32_t words_per_slot, uint32_t pay) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x)
        slots[i] = (uint32_t)(i % words_per_slot) < pay ? WC_EMPTY : 0ull;
}
void launch_walk_cache_init(hipStream_t s, WalkCache C, uint32_t k) {
    if (!C.slots) return;
    const uint64_t n_words = (C.bucket_mask + 1) * 8;
    hipLaunchKernelGGL(k_wc_init, dim3(grid_for(n_words, 256, 256 * 64)), dim3(256), 0, s, C.slots, n_words, k >= 32 ? 4u : 2u, k >= 32 ? 2u : 1u);
}
// (waves per SIMD: seven for one-word k-mers -- 72 registers, five of them spilt, 198-201 ms against 209-214 at six waves -- five for two-word ones)
template <typename K, uint32_t NH, bool CACHE>
__global__ void __launch_bounds__(256, (KT<K>::W == 2 ? 5 : 7)) k_walk(ReadsDev R, BloomDev B, const uint16_t* rv16g, const int32_t* anchor_pos,
                                             const uint8_t* flags, const uint32_t* perm, uint64_t n_walk, uint8_t* events, const uint64_t* ev_off, WalkCache WC) {
    __shared__ uint16_t rv16[256];
    load_rv16(rv16, rv16g, NH ? B.block_mask : 0xFFFFu);
    // (measured, round 3: giving every XCD one contiguous eighth of the order -- workgroup b takes chunk (b % 8) * n/8 + b / 8 --
    // changes nothing, 311 -> 313 ms; nor does it with the reads in true genome order, 277 -> 281 ms: profiles/r3_walk_order.txt)
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n_walk) return;
    uint32_t i = perm[t];
    int32_t a = anchor_pos[i];
    if (a < 0) return;
    uint32_t k = R.k, len = R.len[i];
    const K kmask_k = kmask<K>(k);
    const uint32_t* pk = R.packed + 2 * R.slot_off[i];
    const uint32_t* nm = R.nmask + R.slot_off[i];
    bool hasN = R.n_count[i] != 0;
    // where this read's event bytes go: at its bases' place in the rank's block range, or -- the walk divided by anchor (k_ev_* below) --
    // at the t-th read's place in the slice's own buffer
    uint8_t* ev = events + (ev_off ? ev_off[t] : R.base_off[i] - R.base_off[R.ev_origin]);
    const K anchor = kmer_at<K>(pk, (uint32_t)a, k);
    const K anchor_rc = revcomp(anchor, k);

    // Left and right walks are independent (events are indexed by position), so BOTH run in the same loop, one step of
    // each per iteration: "side A" is the walk towards the genome's left of the anchor (the read's left walk, or its right
    // walk when the anchor is reverse-complemented in the read), "side B" the other one.  Step j of side A probes the same
    // genome k-mer in EVERY read of the anchor, whatever its strand and wherever the anchor sits in it -- so the reads of an
    // anchor, neighbours in the anchor-sorted order, stay in lockstep on both sides for the whole walk and their probes
    // coalesce in the wave's own memory instruction (with one walk after the other, reads left the first walk at different
    // iterations and the second walks of an anchor's reads were out of step).  Two independent chains per lane also keep
    // twice the probes in flight.
    const bool rev = (flags[i] & 1u) != 0;
    const bool leftA = !rev, leftB = rev;                          // direction, in the read, of side A / side B
    K xA = leftA ? anchor_rc : anchor, yA = leftA ? anchor : anchor_rc;          // x: the strand being extended rightwards
    K xB = leftB ? anchor_rc : anchor, yB = leftB ? anchor : anchor_rc;
    const uint32_t nL = (uint32_t)a, nR = len - k - (uint32_t)a;
    const uint32_t nA = leftA ? nL : nR, nB = leftB ? nL : nR;
    const uint32_t nmax = nA > nB ? nA : nB;
    // the read's 2-bit word (16 bases) and N-mask word (32 bases) of each side stay in registers between reloads
    uint32_t pwA = 0, pwA_idx = 0xFFFFFFFFu, nwA = 0, nwA_idx = 0xFFFFFFFFu;
    uint32_t pwB = 0, pwB_idx = 0xFFFFFFFFu, nwB = 0, nwB_idx = 0xFFFFFFFFu;
    // one step of a side in two halves, so that the probes of BOTH sides are in flight before either is waited for:
    // side_probe returns the successor mask (0x100 = nothing to probe: past the side's end, or an N position, whose
    // 'A' is pushed into the k-mer right away), side_apply classifies it, stores the event and moves the k-mer on
    WalkRun<K> WA, WB;
    auto side_probe = [&](bool on, bool left, uint32_t j, K& x, K& y, uint32_t& pw, uint32_t& pw_idx, uint32_t& nw, uint32_t& nw_idx,
                          uint32_t& pos, uint32_t& nt, WalkRun<K>& W) -> uint32_t {
        if (!on) return 0x100u;
        pos = left ? (uint32_t)a - 1 - j : (uint32_t)a + k + j;
        if ((pos >> 4) != pw_idx) { pw_idx = pos >> 4; pw = pk[pw_idx]; }
        nt = (pw >> (30 - 2 * (pos & 15))) & 3u;
        if (hasN) {
            if ((pos >> 5) != nw_idx) { nw_idx = pos >> 5; nw = nm[nw_idx]; }
            if ((nw >> (pos & 31)) & 1u) {                            // N: coded as 'A' on the read strand, nothing stored
                if (CACHE) {                                         // the 'A' leaves the path the cache knows; what was recorded up to here stands
                    if ((W.st & (WR_ON | WR_NEW)) == (WR_ON | WR_NEW)) wc_publish<K>(WC, W);
                    W.st = 0;
                }
                const uint32_t f = left ? 2u : 0u;
                x = ((x << 2) | (K)f) & kmask_k;
                y = (y >> 2) | ((K)(f ^ 2u) << (2 * (k - 1)));
                return 0x100u;
            }
        }
        if (CACHE) {
            if (W.st & 31u) {                                        // a step the cache knows: no probe
                const uint32_t i = ((W.st >> 8) & 31u) - (W.st & 31u);
                W.st--;
                return 1u << ((uint32_t)(W.rec_b >> (2 * i)) & 3u);
            }
            if (wc_is_hop(WC, x)) {
                if ((W.st & (WR_ON | WR_NEW)) == (WR_ON | WR_NEW)) wc_publish<K>(WC, W);
                const uint64_t pay = wc_lookup<K>(WC, x);
                // (a run that ends before the next hop point -- its recorder's read ended there -- is EXTENDED by whoever takes it: the
                // walker goes on recording under the same key from where the run stops)
                const uint32_t m = (uint32_t)(pay >> 56);
                W.rec_key = x; W.rec_b = pay & ((1ull << 56) - 1);
                W.st = WR_ON | (m << 8) | (m ? m - 1 : 0u);
                if (m) return 1u << ((uint32_t)pay & 3u);
            }
            const uint32_t r4 = bloom_contains4<K, NH>(B, rv16, x, y, true);
            if (W.st & WR_ON) {
                const uint32_t n = (W.st >> 8) & 31u;
                if (__popc(r4) == 1 && n < WC_MAX) { W.rec_b |= (uint64_t)__builtin_ctz(r4) << (2 * n); W.st += 1u << 8; W.st |= WR_NEW; }
                else { if (W.st & WR_NEW) wc_publish<K>(WC, W); W.st = 0; }
            }
            return r4;
        }
        return bloom_contains4<K, NH>(B, rv16, x, y, true);
    };
    // (Measured and dropped: every side with its own step count, a side that holds a run making ALL of its steps at once -- no probe, no
    // look-up -- before the two sides' probing step: 232 ms against 200.  The lanes of a wave belong to ~9 anchor groups at
    // different places between their hop points; in nearly every iteration one of them starts a run, and the other lanes wait for it.)
    for (uint32_t j = 0; j < nmax; j++) {
        uint32_t posA = 0, ntA = 0, posB = 0, ntB = 0;
        const uint32_t rA = side_probe(j < nA, leftA, j, xA, yA, pwA, pwA_idx, nwA, nwA_idx, posA, ntA, WA);
        const uint32_t rB = side_probe(j < nB, leftB, j, xB, yB, pwB, pwB_idx, nwB, nwB_idx, posB, ntB, WB);
        if (rA != 0x100u) walk_apply<K>(rA, k, kmask_k, xA, yA, ntA, leftA, ev + posA);
        if (rB != 0x100u) walk_apply<K>(rB, k, kmask_k, xB, yB, ntB, leftB, ev + posB);
    }
    if (CACHE) {                                                         // what was being recorded when a side ended is a valid (shorter) run
        if ((WA.st & (WR_ON | WR_NEW)) == (WR_ON | WR_NEW)) wc_publish<K>(WC, WA);
        if ((WB.st & (WR_ON | WR_NEW)) == (WR_ON | WR_NEW)) wc_publish<K>(WC, WB);
    }
}
// ---- measurement only (LEON_WALK_TILE=1): the LDS-tile form of the probe.  The seven probes of a step all fall in the
// 514 bytes that start at the k-mer's `racine`; here a wave stages that window in LDS once per DISTINCT racine among its lanes
// (130 dwords, cooperative, coalesced) and the lanes that share it read their seven words from the tile.  Same events, byte for byte.
template <typename K>
__device__ inline uint32_t bloom_contains4_tile(const BloomDev& B, const uint16_t* rv16, uint32_t* tile, bool need, K kmer, K rc) {
    const uint32_t k = B.k, lane = lane_id();
    BloomKeys Kk;
    uint32_t pv4 = 0;
    Kk.racine = 0;
    if (need) {                                                    // (the `right` form of bloom_contains4: every lane extends rightwards)
        const K mkm2 = kmask<K>(k - 2);
        const uint32_t p = (uint32_t)(uint64_t)(kmer >> (2 * (k - 2))) & 3u;
        pv4 = cano2(p << 2) | (cano2((p << 2) | 1) << 4) | (cano2((p << 2) | 2) << 8) | (cano2((p << 2) | 3) << 12);
        bloom_keys<K>(B, rv16, kmer & mkm2, rc >> 4, Kk);
    }
    uint32_t w = 0xFFFFFFFFu;
    unsigned long long todo = __ballot(need);
    while (todo) {
        const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
        const uint64_t r = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(Kk.racine >> 32), (int)leader) << 32) |
                           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)Kk.racine, (int)leader);
        const bool mine = need && Kk.racine == r;
        const unsigned long long same = __ballot(mine);
        const uint8_t* src = B.bits + (r >> 3);
        uint32_t v0, v1;
        __builtin_memcpy(&v0, src + 4 * lane, 4);
        __builtin_memcpy(&v1, src + 256 + 4 * lane, 4);
        tile[lane] = v0; tile[64 + lane] = v1;
        if (lane < 3) { uint32_t v2; __builtin_memcpy(&v2, src + 512 + 4 * lane, 4); tile[128 + lane] = v2; }
        __builtin_amdgcn_wave_barrier();
        if (mine) {
            const uint32_t sh0 = (uint32_t)(r & 7);
#pragma unroll
            for (uint32_t i = 0; i < 10; i++) {
                if (i < B.n_hash) {
                    const uint32_t bp = sh0 + Kk.key[i], d = bp >> 5, sft = bp & 31;
                    const uint32_t lo = ((volatile uint32_t*)tile)[d], hi = ((volatile uint32_t*)tile)[d + 1];
                    w &= sft ? ((lo >> sft) | (hi << (32 - sft))) : lo;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        todo &= ~same;
    }
    return ((w >> (pv4 & 15)) & 1u) | (((w >> ((pv4 >> 4) & 15)) & 1u) << 1) |
           (((w >> ((pv4 >> 8) & 15)) & 1u) << 2) | (((w >> ((pv4 >> 12) & 15)) & 1u) << 3);
}
template <typename K>
__global__ void __launch_bounds__(256) k_walk_tile(ReadsDev R, BloomDev B, const uint16_t* rv16g, const int32_t* anchor_pos,
                                                  const uint8_t* flags, const uint32_t* perm, uint64_t n_walk, uint8_t* events) {
    __shared__ uint16_t rv16[256];
    __shared__ uint32_t tiles[4][132];
    load_rv16(rv16, rv16g);
    uint32_t* const tile = tiles[threadIdx.x >> 6];
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    // (every lane of a wave stays to the end: the staging is cooperative)
    bool live = t < n_walk;
    const uint32_t i = live ? perm[t] : 0u;
    const int32_t a = live ? anchor_pos[i] : -1;
    live = live && a >= 0;
    const uint32_t k = R.k, len = live ? R.len[i] : k;
    const K kmask_k = kmask<K>(k);
    const uint32_t* pk = R.packed + 2 * R.slot_off[i];
    const uint32_t* nm = R.nmask + R.slot_off[i];
    const bool hasN = live && R.n_count[i] != 0;
    uint8_t* ev = events + (R.base_off[i] - R.base_off[R.ev_origin]);
    const K anchor = live ? kmer_at<K>(pk, (uint32_t)a, k) : (K)0;
    const K anchor_rc = revcomp(anchor, k);
    const bool rev = live && (flags[i] & 1u) != 0;
    const bool leftA = !rev, leftB = rev;
    K xA = leftA ? anchor_rc : anchor, yA = leftA ? anchor : anchor_rc;
    K xB = leftB ? anchor_rc : anchor, yB = leftB ? anchor : anchor_rc;
    const uint32_t nL = live ? (uint32_t)a : 0u, nR = live ? len - k - (uint32_t)a : 0u;
    const uint32_t nA = leftA ? nL : nR, nB = leftB ? nL : nR;
    uint32_t nmax = nA > nB ? nA : nB;
    for (uint32_t d = 32; d; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)nmax, (int)d); nmax = o > nmax ? o : nmax; }   // the wave's longest walk
    uint32_t pwA = 0, pwA_idx = 0xFFFFFFFFu, nwA = 0, nwA_idx = 0xFFFFFFFFu;
    uint32_t pwB = 0, pwB_idx = 0xFFFFFFFFu, nwB = 0, nwB_idx = 0xFFFFFFFFu;
    // what k_walk's side_probe does before it probes: the position, the read's base, an N pushed into the k-mer; true = probe
    auto side_prep = [&](bool on, bool left, uint32_t j, K& x, K& y, uint32_t& pw, uint32_t& pw_idx, uint32_t& nw, uint32_t& nw_idx,
                         uint32_t& pos, uint32_t& nt) -> bool {
        if (!on) return false;
        pos = left ? (uint32_t)a - 1 - j : (uint32_t)a + k + j;
        if ((pos >> 4) != pw_idx) { pw_idx = pos >> 4; pw = pk[pw_idx]; }
        nt = (pw >> (30 - 2 * (pos & 15))) & 3u;
        if (hasN) {
            if ((pos >> 5) != nw_idx) { nw_idx = pos >> 5; nw = nm[nw_idx]; }
            if ((nw >> (pos & 31)) & 1u) {
                const uint32_t f = left ? 2u : 0u;
                x = ((x << 2) | (K)f) & kmask_k;
                y = (y >> 2) | ((K)(f ^ 2u) << (2 * (k - 1)));
                return false;
            }
        }
        return true;
    };
    for (uint32_t j = 0; j < nmax; j++) {
        uint32_t posA = 0, ntA = 0, posB = 0, ntB = 0;
        const bool pA = side_prep(j < nA, leftA, j, xA, yA, pwA, pwA_idx, nwA, nwA_idx, posA, ntA);
        const bool pB = side_prep(j < nB, leftB, j, xB, yB, pwB, pwB_idx, nwB, nwB_idx, posB, ntB);
        const uint32_t rA = bloom_contains4_tile<K>(B, rv16, tile, pA, xA, yA);
        const uint32_t rB = bloom_contains4_tile<K>(B, rv16, tile, pB, xB, yB);
        if (pA) walk_apply<K>(rA, k, kmask_k, xA, yA, ntA, leftA, ev + posA);
        if (pB) walk_apply<K>(rB, k, kmask_k, xB, yB, ntB, leftB, ev + posB);
    }
}
void launch_walk(hipStream_t s, ReadsDev R, BloomDev B, const uint16_t* rv16, const int32_t* anchor_pos, const uint8_t* flags,
                 const uint32_t* perm, uint64_t n_walk, uint8_t* events, const uint64_t* ev_off, WalkCache wc) {
    if (!n_walk) return;
    uint64_t g = (n_walk + 255) / 256;
    static const bool tile = getenv("LEON_WALK_TILE") != nullptr && getenv("LEON_WALK_TILE")[0] == '1';       // measurement only
    if (tile && !ev_off) { DISPATCH_K(R.k, hipLaunchKernelGGL(k_walk_tile<K>, dim3((uint32_t)g), dim3(256), 0, s, R, B, rv16, anchor_pos, flags, perm, n_walk, events)); return; }
    // (Leon's seven hash functions: an instantiation without the per-hash branches; any other number takes them)
    if (B.n_hash == 7 && wc.slots) { DISPATCH_K(R.k, hipLaunchKernelGGL((k_walk<K, 7, true>), dim3((uint32_t)g), dim3(256), 0, s, R, B, rv16, anchor_pos, flags, perm, n_walk, events, ev_off, wc)); return; }
    if (B.n_hash == 7) { DISPATCH_K(R.k, hipLaunchKernelGGL((k_walk<K, 7, false>), dim3((uint32_t)g), dim3(256), 0, s, R, B, rv16, anchor_pos, flags, perm, n_walk, events, ev_off, wc)); return; }
    DISPATCH_K(R.k, hipLaunchKernelGGL((k_walk<K, 0, false>), dim3((uint32_t)g), dim3(256), 0, s, R, B, rv16, anchor_pos, flags, perm, n_walk, events, ev_off, wc));
}

// ================================================================================================
// The walk divided by ANCHOR among the ranks of one job (leon_dna_set_exchange): the batch's reads, sorted by anchor address, are cut
// into `world` contiguous slices; rank r walks slice r -- whole anchor groups, the sharing that makes the one-GPU walk cheap --
// into a buffer of its own (k_walk with ev_off), and what the walk found travels to the rank that codes the read's block as a list of
// (place in that rank's event buffer, event byte) pairs, one 64-bit word each, grouped by destination: reads in file order are
// in block order, so a scan over the reads' event counts gives every destination a contiguous piece.
// ================================================================================================
// first index of `sorted` (ascending) whose key is >= bound: the number of anchored reads (unanchored ones carry the key 2^32)
__global__ void k_lower_bound(const uint64_t* sorted, uint64_t n, uint64_t bound, unsigned long long* out) {
    if (blockIdx.x || threadIdx.x) return;
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (sorted[mid] < bound) lo = mid + 1; else hi = mid; }
    *out = lo;
}
void launch_lower_bound(hipStream_t s, const uint64_t* sorted, uint64_t n, uint64_t bound, unsigned long long* out) {
    hipLaunchKernelGGL(k_lower_bound, dim3(1), dim3(64), 0, s, sorted, n, bound, out);
}
// Where the slices begin.  A slice's walk costs about 1 ns per read plus 10 ns per anchor GROUP (the group's bloom sectors are fetched
// once and shared by its reads): equal numbers of reads would give the rank with the late, thinly covered anchors twice the work
// of the rank with the early ones (64 against 32 ms at 100 M reads over 8 ranks).  Reads are weighted 1, plus SLICE_GROUP_WEIGHT
// for the first read of every anchor; the slices are cut at equal weight.  (4 until the walk's step lost a quarter of its
// instructions -- what is left is the sectors, which come per group: first / last of eight ranks 26 / 37 ms with 4, 31 / 34 with 8,
// 34 / 29 with 16, 35 / 26 with 32.)
constexpr uint64_t SLICE_GROUP_WEIGHT = 10;
__global__ void k_slice_weights(const uint64_t* sorted_keys, uint64_t n, uint64_t* w, uint64_t group_weight) {
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t <= n; t += (uint64_t)gridDim.x * blockDim.x)
        w[t] = t == n ? 0 : 1 + ((t == 0 || sorted_keys[t] != sorted_keys[t - 1]) ? group_weight : 0);
}
// cum = exclusive sums of the weights (n + 1 entries): split[d] = first t whose cum >= total * d / world
__global__ void k_slice_splits(const uint64_t* cum, uint64_t n, uint32_t world, unsigned long long* split) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d > world) return;
    const uint64_t total = cum[n];
    const uint64_t target = d == world ? total : (uint64_t)(((unsigned __int128)total * d) / world);
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (cum[mid] < target) lo = mid + 1; else hi = mid; }
    split[d] = d == world ? n : lo;
}
void launch_slice_weights(hipStream_t s, const uint64_t* sorted_keys, uint64_t n, uint64_t* w) {
    // (LEON_SLICE_GROUP_WEIGHT: a measurement override -- the SAME value on every rank, or the ranks cut different slices)
    static const uint64_t group_weight = [] { const char* e = getenv("LEON_SLICE_GROUP_WEIGHT"); const long v = e ? atol(e) : 0; return v > 0 && v <= 1024 ? (uint64_t)v : SLICE_GROUP_WEIGHT; }();
    hipLaunchKernelGGL(k_slice_weights, dim3(grid_for(n + 1, 256)), dim3(256), 0, s, sorted_keys, n, w, group_weight);
}
void launch_slice_splits(hipStream_t s, const uint64_t* cum, uint64_t n, uint32_t world, unsigned long long* split) {
    hipLaunchKernelGGL(k_slice_splits, dim3((world + 1 + 63) / 64), dim3(64), 0, s, cum, n, world, split);
}
// the slice's reads in walk order: their lengths (scanned by the caller into their places in the slice's event buffer) and, per
// read of the batch, its index in the slice (0xFFFFFFFF: not in it; the caller fills that first)
__global__ void k_slice_reads(ReadsDev R, const uint32_t* perm, uint64_t n_slice, uint64_t* len_out, uint32_t* slot_of) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j <= n_slice; j += (uint64_t)gridDim.x * blockDim.x) {
        if (j == n_slice) { len_out[j] = 0; continue; }
        const uint32_t i = perm[j];
        len_out[j] = R.len[i];
        slot_of[i] = (uint32_t)j;
    }
}
void launch_slice_reads(hipStream_t s, ReadsDev R, const uint32_t* perm, uint64_t n_slice, uint64_t* len_out, uint32_t* slot_of) {
    hipLaunchKernelGGL(k_slice_reads, dim3(grid_for(n_slice + 1, 256)), dim3(256), 0, s, R, perm, n_slice, len_out, slot_of);
}
__device__ inline uint32_t block_owner(uint64_t b, uint64_t q, uint64_t rm) {     // leon_amd/shard.py block_range, inverted
    const uint64_t cut = (q + 1) * rm;
    return (uint32_t)(b < cut ? b / (q + 1) : rm + (q ? (b - cut) / q : 0));
}
// EMIT = false: cnt[i] = number of non-zero event bytes of read i (0 for reads outside the slice); EMIT = true: the words, at
// send + off[i]: (place of the byte in the owner rank's event buffer << 8) | byte.  One lane per read of the BATCH, in file order.
template <bool EMIT>
__global__ void __launch_bounds__(256) k_ev_words(ReadsDev R, const uint32_t* slot_of, const uint64_t* ev_off, const uint8_t* events, uint64_t n,
                                                 uint32_t rpb, uint64_t n_blocks, uint32_t world, uint64_t* cnt_or_off, uint64_t* send) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { if (!EMIT) cnt_or_off[n] = 0; return; }
    const uint32_t j = slot_of[i];
    if (j == 0xFFFFFFFFu) { if (!EMIT) cnt_or_off[i] = 0; return; }
    const uint32_t len = R.len[i];
    const uint8_t* ev = events + ev_off[j];
    uint64_t w_at = 0, origin = 0;
    if (EMIT) {
        const uint64_t q = n_blocks / world, rm = n_blocks % world;
        const uint32_t owner = block_owner(i / rpb, q, rm);
        const uint64_t lb0 = (uint64_t)owner * q + (owner < rm ? owner : rm);
        origin = R.base_off[i] - R.base_off[lb0 * rpb];          // the read's first byte in its owner's event buffer
        w_at = cnt_or_off[i];
    }
    uint32_t c = 0;
    for (uint32_t p0 = 0; p0 < len; p0 += 16) {                   // (the slice's buffer is padded by 16 bytes; bytes past the read are another read's: masked)
        uint32_t w[4];
        __builtin_memcpy(w, ev + p0, 16);
        if (!(w[0] | w[1] | w[2] | w[3])) continue;
#pragma unroll
        for (uint32_t d = 0; d < 4; d++) {
            uint32_t w4 = w[d];
            const uint32_t q0 = p0 + 4 * d;
            if (q0 >= len) w4 = 0; else if (q0 + 4 > len) w4 &= 0xFFFFFFFFu >> (8 * (q0 + 4 - len));
            while (w4) {
                const uint32_t byte = (uint32_t)__builtin_ctz(w4) >> 3;
                if (EMIT) send[w_at + c] = ((origin + q0 + byte) << 8) | ((w4 >> (8 * byte)) & 0xFFu);
                c++;
                w4 &= ~(0xFFu << (8 * byte));
            }
        }
    }
    if (!EMIT) cnt_or_off[i] = c;
}
void launch_ev_words(hipStream_t s, ReadsDev R, const uint32_t* slot_of, const uint64_t* ev_off, const uint8_t* events, uint64_t n, uint32_t rpb,
                     uint64_t n_blocks, uint32_t world, uint64_t* cnt_or_off, uint64_t* send) {
    const uint32_t g = (uint32_t)((n + 1 + 255) / 256);
    if (send) hipLaunchKernelGGL(k_ev_words<true>, dim3(g), dim3(256), 0, s, R, slot_of, ev_off, events, n, rpb, n_blocks, world, cnt_or_off, send);
    else hipLaunchKernelGGL(k_ev_words<false>, dim3(g), dim3(256), 0, s, R, slot_of, ev_off, events, n, rpb, n_blocks, world, cnt_or_off, send);
}
// the receiving side: the words of every slice for this rank's blocks, into its (zeroed) event buffer; a place beyond the buffer
// (words from another rank are input, not trusted) raises the flag
__global__ void k_ev_scatter(const uint64_t* words, uint64_t n_words, uint8_t* events, uint64_t n_bytes, int* err) {
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n_words; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t w = words[e], at = w >> 8;
        if (at >= n_bytes) { atomicExch(err, 3); continue; }
        events[at] = (uint8_t)w;
    }
}
void launch_ev_scatter(hipStream_t s, const uint64_t* words, uint64_t n_words, uint8_t* events, uint64_t n_bytes, int* err) {
    if (!n_words) return;
    hipLaunchKernelGGL(k_ev_scatter, dim3(grid_for(n_words, 256)), dim3(256), 0, s, words, n_words, events, n_bytes, err);
}

// ================================================================================================
// symbols: the order in which encodeAnchorRead / encodeNoAnchorRead feed the range coder
// ================================================================================================
// prev[i] = index of the previous anchored read of the same block (-1 none): _prevReadSize/_prevAnchorPos/...
__global__ void __launch_bounds__(256) k_prev_anchored(const int32_t* anchor_pos, uint64_t n, uint32_t rpb, uint64_t first_block,
                                                      int64_t* prev) {
    __shared__ long long wmax[4];
    __shared__ long long carry_s;
    uint64_t b0 = (first_block + blockIdx.x) * rpb;
    uint64_t b1 = b0 + rpb < n ? b0 + rpb : n;
    uint32_t lane = lane_id(), w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = -1;
    __syncthreads();
    for (uint64_t base = b0; base < b1; base += 256) {
        uint64_t i = base + threadIdx.x;
        long long v = (i < b1 && anchor_pos[i] >= 0) ? (long long)i : -1;
        long long inc = v;                                   // inclusive max-scan inside the wave
        for (int d = 1; d < 64; d <<= 1) {
            long long o = __shfl_up(inc, d);
            if ((int)lane >= d && o > inc) inc = o;
        }
        if (lane == 63) wmax[w] = inc;
        __syncthreads();
        long long pre = carry_s;
        for (uint32_t j = 0; j < w; j++) if (wmax[j] > pre) pre = wmax[j];
        long long exc = __shfl_up(inc, 1);
        if (lane == 0) exc = -1;
        if (pre > exc) exc = pre;
        if (i < b1) prev[i] = exc;
        __syncthreads();
        if (threadIdx.x == 255) { long long m = inc > pre ? inc : pre; carry_s = m; }
        __syncthreads();
    }
}
void launch_prev_anchored(hipStream_t s, const int32_t* anchor_pos, uint64_t n, uint32_t rpb, uint64_t first_block,
                          uint64_t n_blocks, int64_t* prev) {
    if (!n_blocks) return;
    hipLaunchKernelGGL(k_prev_anchored, dim3((uint32_t)n_blocks), dim3(256), 0, s, anchor_pos, n, rpb, first_block, prev);
}

constexpr uint32_t SYM_STAGE = 2048;         // symbols of a wave's 64 reads staged in LDS by k_symbols (23 per 150 bp read)
struct SymSink {
    uint8_t* p;           // nullptr: count only
    uint64_t n;
    __device__ inline void put(uint32_t model, uint32_t sym) {
        if (p) ((uint16_t*)p)[n] = (uint16_t)(model | (sym << 8));        // one store per symbol
        n++;
    }
    // CompressionUtils::encodeNumeric
    __device__ inline void numeric(uint32_t group, uint64_t v) {
        uint32_t bc = 1;
        while (bc < 8 && (v >> (8 * bc)) != 0) bc++;
        put(numeric_model_id(group, 0), bc);
        for (uint32_t b = 0; b < bc; b++) put(numeric_model_id(group, b + 1), (uint32_t)(v >> (8 * b)) & 0xff);
    }
    // CompressionUtils::getDeltaValue + the delta-type symbol
    __device__ inline void delta(uint32_t type_model, uint32_t group, uint64_t value, uint64_t prev) {
        uint32_t dt = 0; uint64_t dv = value;
        if (value > prev) { uint64_t d = value - prev; if (d < value) { dt = 1; dv = d; } }
        else              { uint64_t d = prev - value; if (d < value) { dt = 2; dv = d; } }
        put(type_model, dt);
        numeric(group, dv);
    }
};

// EMIT = false: sym_off[li] = number of symbols of the read and n_err[2 li], n_err[2 li + 1] = its number of error positions and which of its
// 16-byte chunks of event bytes hold an event, from ONE scan of the read's event bytes; EMIT = true: the symbols, written at syms + 2 * sym_off[li].  One lane per read; the
// events are read as dwords (the memory pipeline charges per wave-wide instruction, not per byte).
template <bool EMIT>
__global__ void __launch_bounds__(256) k_symbols(ReadsDev R, const int32_t* anchor_pos, const uint32_t* anchor_addr,
                                                const uint8_t* flags, const int64_t* prev, const uint8_t* events,
                                                uint64_t r0, uint64_t n_local, uint64_t* sym_off, uint32_t* n_err, uint8_t* syms) {
    const uint64_t li = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    // EMIT: the symbols of a wave's 64 reads are ONE contiguous span of the stream (sym_off is a prefix sum in read order).  Lanes storing two bytes
    // at a time, each into its own read's part, made every store instruction 64 partial lines; the span is put together in LDS instead (4 KB per
    // wave) and leaves in whole lines.  A span beyond the buffer (reads without an anchor: a symbol per base) is stored directly, as before.
    __shared__ uint16_t stage[EMIT ? 4 : 1][EMIT ? SYM_STAGE : 1];
    const uint32_t wv = threadIdx.x >> 6;
    const uint64_t li0 = li - (threadIdx.x & 63);                // (wave-uniform)
    if (li0 >= n_local) return;
    uint64_t w_begin = 0, w_end = 0;
    bool staged = false;
    if (EMIT) {
        w_begin = sym_off[li0];
        w_end = sym_off[li0 + 64 < n_local ? li0 + 64 : n_local];
        staged = w_end - w_begin <= SYM_STAGE;
    }
    if (li < n_local) {
    const uint64_t i = r0 + li;
    uint32_t len = R.len[i], k = R.k;
    const uint32_t* pk = R.packed + 2 * R.slot_off[i];
    const uint32_t* nm = R.nmask + R.slot_off[i];
    uint32_t nN = R.n_count[i];
    SymSink S;
    S.p = !EMIT ? nullptr : staged ? (uint8_t*)&stage[wv][sym_off[li] - w_begin] : syms + 2 * sym_off[li];
    S.n = 0;
    int32_t a = anchor_pos[i];
    if (a < 0) {                                              // DnaEncoder::encodeNoAnchorRead
        S.put(M_READ_TYPE, 1);
        S.numeric(G_NOANCHOR_READSIZE, len);
        if (EMIT) {
            for (uint32_t p = 0; p < len; p++) {
                bool isN = nN && ((nm[p >> 5] >> (p & 31)) & 1u);
                S.put(M_NOANCHOR_READ, isN ? 4u : base_at(pk, p));
            }
        } else S.n += len;
    } else {                                                  // DnaEncoder::encodeAnchorRead
        int64_t q = prev[i];
        uint64_t pLen = 0, pPos = 0, pAddr = 0;
        if (q >= 0) { pLen = R.len[q]; pPos = (uint64_t)anchor_pos[q]; pAddr = anchor_addr[q]; }
        S.put(M_READ_TYPE, 0);
        S.delta(M_READSIZE_DT, G_READSIZE, len, pLen);
        S.delta(M_ANCHORPOS_DT, G_ANCHOR_POS, (uint64_t)a, pPos);
        S.delta(M_ANCHORADDR_DT, G_ANCHOR_ADDRESS, anchor_addr[i], pAddr);
        S.put(M_ANCHOR_REVCOMP, flags[i] & 1u);
        const uint8_t* ev = events + (R.base_off[i] - R.base_off[R.ev_origin]);
        S.numeric(G_NUMERIC, nN);                             // N positions, delta coded
        if (nN) {
            uint32_t prevN = 0;
            for (uint32_t p = 0; p < len; p++)
                if ((nm[p >> 5] >> (p & 31)) & 1u) { S.numeric(G_NPOS, p - prevN); prevN = p; }
        }
        // The event bytes are read 16 at a time (the memory pipeline charges per wave-wide instruction, not per byte), in
        // chunks aligned to the read's first base; the buffer is padded, bytes past the read's end are masked off.
        auto chunk = [&](uint32_t p0, uint32_t w[4]) {
            __builtin_memcpy(w, ev + p0, 16);
            if (p0 + 16 > len) {
#pragma unroll
                for (uint32_t d = 0; d < 4; d++) {
                    const uint32_t q0 = p0 + 4 * d;
                    w[d] = q0 >= len ? 0u : (q0 + 4 > len ? w[d] & (0xFFFFFFFFu >> (8 * (q0 + 4 - len))) : w[d]);
                }
            }
        };
        if (!EMIT) {
            // one ascending scan: error positions (count + their numerics' sizes) and the number of bifurcation symbols
            uint32_t nErr = 0, prevE = 0, nBif = 0, cm = 0;
            for (uint32_t p0 = 0; p0 < len; p0 += 16) {
                uint32_t w[4];
                chunk(p0, w);
                if (!(w[0] | w[1] | w[2] | w[3])) continue;
                cm |= 1u << ((p0 >> 4) < 31u ? (p0 >> 4) : 31u);     // which 16-byte chunks hold an event at all (bit 31: some chunk from the 32nd on)
#pragma unroll
                for (uint32_t d = 0; d < 4; d++) {
                    const uint32_t w4 = w[d], p = p0 + 4 * d;
                    const uint32_t c3 = w4 & 0x07070707u;
                    nBif += __popc((c3 | (c3 >> 1) | (c3 >> 2)) & 0x01010101u);
                    uint32_t e4 = w4 & 0x08080808u;
                    while (e4) {
                        uint32_t qq = p + ((uint32_t)__builtin_ctz(e4) >> 3);
                        S.numeric(G_ERRPOS, qq - prevE); prevE = qq; nErr++;
                        e4 &= e4 - 1;
                    }
                }
            }
            S.numeric(G_LEFT_ERROR, nErr);
            S.n += nBif;
            n_err[2 * li] = nErr; n_err[2 * li + 1] = cm;        // (the emitting pass loads only the chunks named here: ~2.5 of a 150 bp read's 10)
        } else {
            const uint32_t nErr = n_err[2 * li], cm = n_err[2 * li + 1];     // error positions, ascending
            auto holds = [&](uint32_t p0) { const uint32_t c = p0 >> 4; return ((cm >> (c < 31u ? c : 31u)) & 1u) != 0; };
            S.numeric(G_LEFT_ERROR, nErr);
            if (nErr) {
                uint32_t prevE = 0, left = nErr;
                for (uint32_t p0 = 0; p0 < len && left; p0 += 16) {
                    if (!holds(p0)) continue;
                    uint32_t w[4];
                    chunk(p0, w);
#pragma unroll
                    for (uint32_t d = 0; d < 4; d++) {
                        uint32_t e4 = w[d] & 0x08080808u;
                        while (e4) {
                            uint32_t qq = p0 + 4 * d + ((uint32_t)__builtin_ctz(e4) >> 3);
                            S.numeric(G_ERRPOS, qq - prevE); prevE = qq; left--;
                            e4 &= e4 - 1;
                        }
                    }
                }
            }
            // bifurcations: left walk (a-1 .. 0), then right walk (a+k .. len-1)
            for (int32_t p0 = a > 0 ? ((a - 1) & ~15) : -16; p0 >= 0; p0 -= 16) {     // chunks descending, bytes descending
                if (!holds((uint32_t)p0)) continue;
                uint32_t w[4];
                chunk((uint32_t)p0, w);
                if (((w[0] | w[1] | w[2] | w[3]) & 0x07070707u) == 0) continue;
#pragma unroll
                for (int32_t j = 15; j >= 0; j--) {
                    if (p0 + j >= a) continue;
                    const uint32_t cc = (w[j >> 2] >> (8 * (j & 3))) & 7u;
                    if (cc >= EV_NT0) S.put(M_BIFURCATION, cc - EV_NT0); else if (cc) S.put(M_BIFURCATION_BINARY, cc - EV_BIN0);
                }
            }
            const uint32_t rs = (uint32_t)a + k;
            for (uint32_t p0 = rs & ~15u; p0 < len; p0 += 16) {
                if (!holds(p0)) continue;
                uint32_t w[4];
                chunk(p0, w);
                if (((w[0] | w[1] | w[2] | w[3]) & 0x07070707u) == 0) continue;
#pragma unroll
                for (uint32_t j = 0; j < 16; j++) {
                    if (p0 + j < rs) continue;
                    const uint32_t cc = (w[j >> 2] >> (8 * (j & 3))) & 7u;
                    if (cc >= EV_NT0) S.put(M_BIFURCATION, cc - EV_NT0); else if (cc) S.put(M_BIFURCATION_BINARY, cc - EV_BIN0);
                }
            }
        }
    }
    if (!EMIT) sym_off[li] = S.n;
    }
    if (EMIT && staged) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (the lanes' stores went through flat addresses: both counters)
        __builtin_amdgcn_wave_barrier();
        uint16_t* const out = (uint16_t*)syms + w_begin;
        const uint32_t ns = (uint32_t)(w_end - w_begin), lane = threadIdx.x & 63;
        for (uint32_t x = lane; x < ns; x += 64) out[x] = stage[wv][x];
    }
}
void launch_symbols(hipStream_t s, ReadsDev R, const int32_t* anchor_pos, const uint32_t* anchor_addr, const uint8_t* flags,
                    const int64_t* prev, const uint8_t* events, uint64_t r0, uint64_t n_local, uint64_t* sym_off, uint32_t* n_err,
                    uint8_t* syms) {
    if (!n_local) return;
    uint64_t g = (n_local + 255) / 256;
    if (syms) hipLaunchKernelGGL(k_symbols<true>, dim3((uint32_t)g), dim3(256), 0, s, R, anchor_pos, anchor_addr, flags, prev, events, r0, n_local,
                                 sym_off, n_err, syms);
    else hipLaunchKernelGGL(k_symbols<false>, dim3((uint32_t)g), dim3(256), 0, s, R, anchor_pos, anchor_addr, flags, prev, events, r0, n_local,
                            sym_off, n_err, syms);
}

// per block: symbol range and output capacity offsets (3 bytes per symbol + 64 or more)
__global__ void k_block_ranges(const uint64_t* sym_off, uint64_t n_reads, uint32_t rpb, uint64_t n_blocks,
                               uint64_t* blk_begin, uint64_t* out_off) {
    for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b <= n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t r = b * rpb;
        if (r > n_reads) r = n_reads;
        uint64_t so = sym_off[r];                             // sym_off has n_reads+1 entries
        blk_begin[b] = so;
        out_off[b] = ((3 * so + 7) & ~7ull) + 64 * b;    // 8-byte aligned: the coder stores 8 bytes at a time
    }
}
void launch_block_ranges(hipStream_t s, const uint64_t* sym_off, uint64_t n_reads, uint32_t rpb, uint64_t n_blocks,
                         uint64_t* blk_begin, uint64_t* out_off) {
    hipLaunchKernelGGL(k_block_ranges, dim3(grid_for(n_blocks + 1, 256)), dim3(256), 0, s, sym_off, n_reads, rpb, n_blocks,
                       blk_begin, out_off);
}

// the longest block's symbol count: the host picks the range coder's instantiation by it (rc_kernels.hip)
__global__ void k_max_block_syms(const uint64_t* sym_off, uint64_t n_reads, uint32_t rpb, uint64_t n_blocks, unsigned long long* out_max) {
    for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r0 = b * rpb, r1 = r0 + rpb < n_reads ? r0 + rpb : n_reads;
        atomicMax(out_max, (unsigned long long)(sym_off[r1] - sym_off[r0]));
    }
}
void launch_max_block_syms(hipStream_t s, const uint64_t* sym_off, uint64_t n_reads, uint32_t rpb, uint64_t n_blocks, unsigned long long* out_max) {
    if (!n_blocks) return;
    hipLaunchKernelGGL(k_max_block_syms, dim3(grid_for(n_blocks, 256)), dim3(256), 0, s, sym_off, n_reads, rpb, n_blocks, out_max);
}

__global__ void k_gather_payload(const uint8_t* out, const uint64_t* out_off, const uint64_t* dst_off, const uint64_t* sizes,
                                 uint64_t n_blocks, uint8_t* dst) {
    for (uint64_t b = blockIdx.y; b < n_blocks; b += gridDim.y) {
        const uint8_t* src = out + out_off[b];
        uint8_t* d = dst + dst_off[b];
        uint64_t n = sizes[b];
        for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) d[i] = src[i];
    }
}
void launch_gather_payload(hipStream_t s, const uint8_t* out, const uint64_t* out_off, const uint64_t* dst_off,
                           const uint64_t* sizes, uint64_t n_blocks, uint8_t* dst) {
    if (!n_blocks) return;
    uint32_t gy = n_blocks > 65535 ? 65535u : (uint32_t)n_blocks;
    hipLaunchKernelGGL(k_gather_payload, dim3(8, gy), dim3(256), 0, s, out, out_off, dst_off, sizes, n_blocks, dst);
}

}  // namespace leon

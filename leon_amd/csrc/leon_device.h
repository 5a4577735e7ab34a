// leon_device.h -- device-side building blocks shared by the gfx950 kernels of the DNA encode path.
// k-mer model (gatb kmer/impl/Model.hpp [RECALLED]): 2-bit code A0 C1 T2 G3, first base in the highest bits.
// The k-mer type K is uint64_t for k < 32 (upstream LargeInt<1>) and unsigned __int128 for 32 <= k < 64
// (LargeInt<2> / NativeInt128); kernels are templates on K.  Bloom geometry: BloomNeighborCoherent (Bloom.hpp [RECALLED]).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace leon {

typedef unsigned __int128 u128;
template <typename K> struct KT;
template <> struct KT<uint64_t> { static constexpr uint32_t W = 1; };
template <> struct KT<u128> { static constexpr uint32_t W = 2; };
__host__ __device__ inline uint32_t kmer_words(uint32_t k) { return k >= 32 ? 2u : 1u; }

// ---- model ids of the symbol stream (AbstractDnaCoder's Order0Model members [RECALLED]) ----
enum : uint32_t {
    M_READ_TYPE = 0, M_NOANCHOR_READ = 1, M_BIFURCATION = 2, M_BIFURCATION_BINARY = 3,
    M_READSIZE_DT = 4, M_ANCHORPOS_DT = 5, M_ANCHORADDR_DT = 6, M_ANCHOR_REVCOMP = 7,
    N_SMALL_MODELS = 8,
    // numeric groups (CompressionUtils::encodeNumeric: byte-count model + one model per byte index)
    G_ANCHOR_ADDRESS = 0, G_ANCHOR_POS = 1, G_NOANCHOR_READSIZE = 2, G_READSIZE = 3,
    G_NPOS = 4, G_ERRPOS = 5, G_NUMERIC = 6, G_LEFT_ERROR = 7,
    N_NUM_GROUPS = 8, MODELS_PER_NUMERIC = 9,
    N_MODELS = N_SMALL_MODELS + N_NUM_GROUPS * MODELS_PER_NUMERIC   // 80
};
// alphabet sizes of the small models, one nibble per model id: the DNA stream's and the header stream's sets
constexpr uint32_t SMALL_SIZES_DNA = 0x23332552u;
constexpr uint32_t SMALL_SIZES_HEADER = 0x22222229u;
__host__ __device__ inline uint32_t small_size_of(uint32_t sizes, uint32_t m) { return (sizes >> (4 * m)) & 15u; }
__host__ __device__ inline uint32_t small_model_size(uint32_t m) {
    // readType 2, noAnchorRead 5, bifurcation 5, binary 2, three delta-type models 3, revcomp 2
    return (SMALL_SIZES_DNA >> (4 * m)) & 15u;
}
__host__ __device__ inline uint32_t numeric_model_id(uint32_t group, uint32_t idx) {
    return N_SMALL_MODELS + group * MODELS_PER_NUMERIC + idx;
}

// ---- the header stream's model set (AbstractHeaderCoder's Order0Model members [RECALLED]) and record types ----
// One small model (the record type, 9 symbols) and 14 256-symbol models; ids in the space k_rc_encode is given.
enum : uint32_t {
    HM_TYPE = 0, HM_FIELD_INDEX = 8, HM_FIELD_COLUMN = 9, HM_MIS_SIZE = 10, HM_ASCII = 11, HM_ZERO = 12, HM_NUMERIC0 = 13,   // .. 21
    H_END = 1, H_END_MATCH = 2, H_FIELD_ASCII = 3, H_FIELD_NUMERIC = 4, H_FIELD_DELTA = 5, H_FIELD_DELTA_2 = 6,
    H_FIELD_ZERO_ONLY = 7, H_FIELD_ZERO_AND_NUMERIC = 8, H_TYPE_COUNT = 9
};

// event byte written by the walk per read position
enum : uint8_t { EV_BIN0 = 1, EV_BIN1 = 2, EV_NT0 = 3, EV_ERROR = 8 };

constexpr uint64_t KEY_EMPTY = ~0ull;            // one-word keys; two-word keys: high word ~0 = empty, ~0-1 = being written
constexpr uint64_t KEY_LOCKED = ~0ull - 1;
constexpr uint64_t IDX_INF = ~0ull;

struct BloomDev {
    const uint8_t* bits;
    uint64_t reduced_tai, mod_magic, seed0;
    uint32_t k, n_hash, block_mask, pad;
};

// ---- k-mer arithmetic ----
template <typename K> __device__ inline K kmask(uint32_t nbases) { return (((K)1) << (2 * nbases)) - 1; }   // nbases < 32 * W
__device__ inline uint32_t rev2bit32(uint32_t x) {                   // reverse the 16 2-bit groups and complement them
    x = __builtin_bitreverse32(x);                                   // (v_bfrev_b32: the groups in order, the two bits of each swapped)
    return (((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1)) ^ 0xAAAAAAAAu;
}
__device__ inline uint64_t rev2bit64(uint64_t x) {                   // ... of the 32 groups: the halves change places
    return ((uint64_t)rev2bit32((uint32_t)x) << 32) | rev2bit32((uint32_t)(x >> 32));
}
__device__ inline uint64_t revcomp(uint64_t x, uint32_t k) { return rev2bit64(x) >> (64 - 2 * k); }
__device__ inline u128 revcomp(u128 x, uint32_t k) {
    u128 r = ((u128)rev2bit64((uint64_t)x) << 64) | rev2bit64((uint64_t)(x >> 64));
    return r >> (128 - 2 * k);
}
// NativeInt64::hash64 [RECALLED]
__device__ inline uint64_t hash64(uint64_t key, uint64_t seed) {
    uint64_t hash = seed;
    hash ^= (hash << 7) ^ key * (hash >> 3) ^ (~((hash << 11) + (key ^ (hash >> 5))));
    hash = (~hash) + (hash << 21);
    hash = hash ^ (hash >> 24);
    hash = (hash + (hash << 3)) + (hash << 8);
    hash = hash ^ (hash >> 14);
    hash = (hash + (hash << 2)) + (hash << 4);
    hash = hash ^ (hash >> 28);
    hash = hash + (hash << 31);
    return hash;
}
// hash1(LargeInt<precision>): XOR of hash64 over the type's 64-bit chunks [RECALLED]
__device__ inline uint64_t hash1(uint64_t x, uint64_t seed) { return hash64(x, seed); }
__device__ inline uint64_t hash1(u128 x, uint64_t seed) { return hash64((uint64_t)x, seed) ^ hash64((uint64_t)(x >> 64), seed); }
// n mod d with magic = floor((2^64-1)/d): one mulhi, one multiply, conditional subtracts
// (ONE: magic * d >= 2^64 - d, so n * magic / 2^64 > n / d - 1 and the quotient is the true one or one short: r < 2 d)
__device__ inline uint64_t fastmod(uint64_t n, uint64_t d, uint64_t magic) {
    uint64_t q = __umul64hi(n, magic);
    uint64_t r = n - q * d;
    if (r >= d) r -= d;
    return r;
}
__device__ inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__device__ inline uint64_t key_hash(uint64_t k) { return mix64(k); }
__device__ inline uint64_t key_hash(u128 k) { return mix64((uint64_t)k ^ mix64((uint64_t)(k >> 64) + 0x9E3779B97F4A7C15ULL)); }
// cano2[16] of BloomNeighborCoherent as 16 nibbles
__device__ inline uint32_t cano2(uint32_t v) { return (uint32_t)(0x51D9409873543210ULL >> (4 * v)) & 15u; }
// the four values contains4 needs at once, a nibble per neighbour nt: cano2(4 p + nt) for a fixed prefix p -- 16 consecutive bits of
// the table itself -- and cano2(4 nt + s) for a fixed suffix s (the same nibbles regrouped): one shift instead of four look-ups
__device__ inline uint32_t cano2_right(uint32_t p) { return (uint32_t)(0x51D9409873543210ULL >> (16 * p)) & 0xFFFFu; }
__device__ inline uint32_t cano2_left(uint32_t s) { return (uint32_t)(0x54731032D9519840ULL >> (16 * s)) & 0xFFFFu; }

// ---- packed reads: 16 bases per dword, base j of a read at bits 30-2*(j&15) of dword j>>4 ----
__device__ inline uint32_t base_at(const uint32_t* pk, uint32_t pos) {
    return (pk[pos >> 4] >> (30 - 2 * (pos & 15))) & 3u;
}
// k-mer from consecutive dwords w[0..] whose first base is `first` bases before the k-mer (first < 16)
__device__ inline uint64_t kmer_from3(uint64_t w0, uint64_t w1, uint64_t w2, uint32_t first, uint32_t k) {
    const uint32_t s = 2 * first;
    const uint64_t hi = (w0 << 32) | w1;
    const uint64_t x = s ? ((hi << s) | (w2 >> (32 - s))) : hi;
    return x >> (64 - 2 * k);
}
__device__ inline u128 kmer_from5(uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3, uint64_t w4, uint32_t first, uint32_t k) {
    const uint32_t s = 2 * first;
    const u128 hi = ((u128)((w0 << 32) | w1) << 64) | ((w2 << 32) | w3);
    const u128 x = s ? ((hi << s) | (u128)(w4 >> (32 - s))) : hi;
    return x >> (128 - 2 * k);
}
// k-mer starting at base p (the packed buffer is padded so that the dwords read past a read exist)
template <typename K> __device__ inline K kmer_at(const uint32_t* pk, uint32_t p, uint32_t k);
template <> __device__ inline uint64_t kmer_at<uint64_t>(const uint32_t* pk, uint32_t p, uint32_t k) {
    const uint32_t d = p >> 4;
    return kmer_from3(pk[d], pk[d + 1], pk[d + 2], p & 15, k);
}
template <> __device__ inline u128 kmer_at<u128>(const uint32_t* pk, uint32_t p, uint32_t k) {
    const uint32_t d = p >> 4;
    return kmer_from5(pk[d], pk[d + 1], pk[d + 2], pk[d + 3], pk[d + 4], p & 15, k);
}

// ---- BloomNeighborCoherent geometry ----
// racine and the per-hash offsets depend only on the canonical middle (k-2)-mer
struct BloomKeys { uint64_t racine; uint32_t key[10]; };

// NH: the number of hash functions when the kernel was built for it (the walk: Leon's 7), 0 = B.n_hash at run time -- a ladder of
// scalar branches, one per hash, in the middle of the walk's step
template <typename K, uint32_t NH = 0>
__device__ inline void bloom_keys(const BloomDev& B, const uint16_t* rv16, K hp_fwd, K hp_rc, BloomKeys& Kk) {
    const K hp = hp_rc < hp_fwd ? hp_rc : hp_fwd;
    Kk.racine = fastmod(hash1(hp, B.seed0), B.reduced_tai, B.mod_magic);
    Kk.key[0] = 0;
    const uint32_t low = (uint32_t)(uint64_t)hp;                  // simplehash16 looks at value[0] >> i, 16 bits
    const uint32_t n_hash = NH ? NH : B.n_hash;
#pragma unroll
    for (uint32_t i = 1; i < 10; i++) {
        if (i < n_hash) {
            const uint32_t in = low >> i;
            // (NH != 0: the kernel staged the table already masked -- load_rv16(.., B.block_mask) -- and spares an `and` per hash)
            Kk.key[i] = NH ? (uint32_t)(rv16[in & 255] ^ rv16[(in >> 8) & 255]) : (uint32_t)(rv16[in & 255] ^ rv16[(in >> 8) & 255]) & B.block_mask;
        }
    }
}
// 25 or more bits of the bloom starting at bit `bitpos`: ONE unaligned dword load (the array is padded by 16 bytes;
// gfx950 serves unaligned dword loads in hardware) -- the memory pipeline pays per request, not per byte
__device__ inline uint32_t bloom_window(const BloomDev& B, uint64_t bitpos) {
    uint32_t w;
    __builtin_memcpy(&w, B.bits + (bitpos >> 3), 4);
    return w >> (bitpos & 7);
}
// contains4: pv4 packs the four canonical prefix+suffix values (4 bits each, neighbour nt in nibble nt).
// Hash i of neighbour n sits at bit racine + key[i] + pv[n]: the same offset pv[n] inside every hash's window, so the
// windows are ANDed first and the four bits are extracted once (the kernel is VALU-issue bound, not memory bound).
template <uint32_t NH = 0>
__device__ inline uint32_t bloom_probe4(const BloomDev& B, const BloomKeys& Kk, uint32_t pv4) {
    uint32_t w = 0xFFFFFFFFu;
    const uint32_t n_hash = NH ? NH : B.n_hash;
#pragma unroll
    for (uint32_t i = 0; i < 10; i++) {
        if (i < n_hash) w &= bloom_window(B, Kk.racine + Kk.key[i]);
    }
    return ((w >> (pv4 & 15)) & 1u) | (((w >> ((pv4 >> 4) & 15)) & 1u) << 1) |
           (((w >> ((pv4 >> 8) & 15)) & 1u) << 2) | (((w >> ((pv4 >> 12) & 15)) & 1u) << 3);
}
// BloomNeighborCoherent positions of one k-mer (insert / contains)
template <typename K>
__device__ inline uint32_t bloom_item_keys(const BloomDev& B, const uint16_t* rv16, K item, BloomKeys& Kk) {
    const uint32_t k = B.k;
    const uint32_t pv = cano2((uint32_t)((uint64_t)(item >> (2 * (k - 1))) & 3) << 2 | ((uint32_t)(uint64_t)item & 3u));
    const K hp = (item >> 2) & kmask<K>(k - 2);
    bloom_keys<K>(B, rv16, hp, revcomp(hp, k - 2), Kk);
    return pv;
}
template <typename K> __device__ inline bool bloom_contains(const BloomDev& B, const uint16_t* rv16, K item) {
    BloomKeys Kk;
    const uint32_t pv = bloom_item_keys<K>(B, rv16, item, Kk);
    bool ok = true;
    for (uint32_t i = 0; i < B.n_hash && ok; i++) ok = (bloom_window(B, Kk.racine + Kk.key[i] + pv) & 1u) != 0;
    return ok;
}
// contains(min(x, r)) for a k-mer x whose reverse complement r is at hand (kernels that keep both incrementally): no reverse
// complement is formed
template <typename K, uint32_t NH = 0> __device__ inline bool bloom_contains_xr(const BloomDev& B, const uint16_t* rv16, K x, K r) {
    const uint32_t k = B.k;
    const K item = r < x ? r : x;
    const uint32_t pv = cano2((uint32_t)((uint64_t)(item >> (2 * (k - 1))) & 3) << 2 | ((uint32_t)(uint64_t)item & 3u));
    const K mk = kmask<K>(k - 2);
    BloomKeys Kk;
    bloom_keys<K, NH>(B, rv16, (x >> 2) & mk, (r >> 2) & mk, Kk);    // the middle (k-2)-mer and ITS reverse complement
    // the first hash alone, then the others together: a k-mer that is not there (every k-mer over a sequencing error) mostly stops
    // at its first sector, one that is there costs two round trips instead of seven
    uint32_t w = bloom_window(B, Kk.racine);
    if (!((w >> pv) & 1u)) return false;
    const uint32_t n_hash = NH ? NH : B.n_hash;
#pragma unroll
    for (uint32_t i = 1; i < 10; i++) {
        if (i < n_hash) w &= bloom_window(B, Kk.racine + Kk.key[i]);
    }
    return ((w >> pv) & 1u) != 0;
}
// BloomNeighborCoherent::contains4(item, right) from a k-mer and its reverse complement
template <typename K, uint32_t NH = 0>
__device__ inline uint32_t bloom_contains4(const BloomDev& B, const uint16_t* rv16, K kmer, K rc, bool right) {
    const uint32_t k = B.k;
    const K mkm2 = kmask<K>(k - 2);
    K hpf, hpr; uint32_t pv4;
    if (right) {          // elem = kmer[1..k-1] + X : middle = kmer[2..k-1], prefix = kmer[1], suffix varies
        hpf = kmer & mkm2; hpr = rc >> 4;
        const uint32_t p = (uint32_t)(uint64_t)(kmer >> (2 * (k - 2))) & 3u;
        pv4 = cano2_right(p);
    } else {              // elem = X + kmer[0..k-2] : middle = kmer[0..k-3], prefix varies, suffix = kmer[k-2]
        hpf = kmer >> 4; hpr = rc & mkm2;
        const uint32_t s = (uint32_t)(uint64_t)(kmer >> 2) & 3u;
        pv4 = cano2_left(s);
    }
    BloomKeys Kk;
    bloom_keys<K, NH>(B, rv16, hpf, hpr, Kk);
    return bloom_probe4<NH>(B, Kk, pv4);
}

// stage the low 16 bits of the 256-entry simplehash16 table in LDS
__device__ inline void load_rv16(uint16_t* lds, const uint16_t* g, uint32_t mask = 0xFFFFu) {
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) lds[i] = (uint16_t)(g[i] & mask);
    __syncthreads();
}

__device__ inline uint32_t lane_id() { return threadIdx.x & 63u; }

// Wave-per-read kernels: the dwords covering k-mers [base, base+64) are loaded once (lane l < 12 holds dword
// (base>>4)+l) and every lane assembles its k-mer with cross-lane reads instead of gathers.
// Must be called by all 64 lanes (p is clamped by the caller).
__device__ inline uint32_t pass_words(const uint32_t* pk, uint32_t base, uint32_t lane) {
    return lane < 12 ? pk[(base >> 4) + lane] : 0u;
}
template <typename K> __device__ inline K canon_from_words(uint32_t words, uint32_t base, uint32_t p, uint32_t k);
template <> __device__ inline uint64_t canon_from_words<uint64_t>(uint32_t words, uint32_t base, uint32_t p, uint32_t k) {
    const int d = (int)((p >> 4) - (base >> 4));
    const uint64_t km = kmer_from3((uint32_t)__shfl((int)words, d), (uint32_t)__shfl((int)words, d + 1),
                                   (uint32_t)__shfl((int)words, d + 2), p & 15, k);
    const uint64_t rc = revcomp(km, k);
    return rc < km ? rc : km;
}
template <> __device__ inline u128 canon_from_words<u128>(uint32_t words, uint32_t base, uint32_t p, uint32_t k) {
    const int d = (int)((p >> 4) - (base >> 4));                     // d + 4 <= 8 < 12
    const u128 km = kmer_from5((uint32_t)__shfl((int)words, d), (uint32_t)__shfl((int)words, d + 1), (uint32_t)__shfl((int)words, d + 2),
                               (uint32_t)__shfl((int)words, d + 3), (uint32_t)__shfl((int)words, d + 4), p & 15, k);
    const u128 rc = revcomp(km, k);
    return rc < km ? rc : km;
}

// k-mers cross the C-ABI as W 64-bit words each, low word first
template <typename K> __device__ inline K load_kmer(const uint64_t* w);
template <> __device__ inline uint64_t load_kmer<uint64_t>(const uint64_t* w) { return w[0]; }
template <> __device__ inline u128 load_kmer<u128>(const uint64_t* w) { return ((u128)w[1] << 64) | w[0]; }
__device__ inline void store_kmer(uint64_t* w, uint64_t x) { w[0] = x; }
__device__ inline void store_kmer(uint64_t* w, u128 x) { w[0] = (uint64_t)x; w[1] = (uint64_t)(x >> 64); }

}  // namespace leon

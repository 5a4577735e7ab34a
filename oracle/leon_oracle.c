/*
 * leon_oracle.c -- CPU restatement of Leon's DNA encode path.  TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED: see leon_oracle.h.  Every function cites the upstream gatb-core function it
 * restates ([RECALLED]: upstream source absent from /root/reference).
 */
#include "leon_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* ------------------------------------------------------------------------------------------ */
/* small growable byte / word vectors                                                         */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint8_t* p; uint64_t n, cap; } bytevec;

static int bv_reserve(bytevec* v, uint64_t extra) {
    if (v->n + extra <= v->cap) return 0;
    uint64_t nc = v->cap ? v->cap * 2 : 256;
    while (nc < v->n + extra) nc *= 2;
    uint8_t* q = (uint8_t*)realloc(v->p, nc);
    if (!q) return -1;
    v->p = q; v->cap = nc;
    return 0;
}
static inline void bv_push(bytevec* v, uint8_t b) {
    if (v->n == v->cap) bv_reserve(v, 1);
    v->p[v->n++] = b;
}
static void bv_append(bytevec* v, const void* src, uint64_t n) {
    bv_reserve(v, n);
    memcpy(v->p + v->n, src, n);
    v->n += n;
}

/* ------------------------------------------------------------------------------------------ */
/* k-mer model: kmer/impl/Model.hpp  -- 2-bit code (c>>1)&3 : A0 C1 T2 G3, first base highest   */
/* Leon::nt2bin / bin2nt (Leon.cpp) use the same code with N = 4.                              */
/* ------------------------------------------------------------------------------------------ */
static inline int nt2bin(char c) {
    switch (c) { case 'A': return 0; case 'C': return 1; case 'T': return 2; case 'G': return 3; default: return 4; }
}
static const char BIN2NT[5] = { 'A', 'C', 'T', 'G', 'N' };

/* k-mers: LargeInt<1> (one 64-bit word) for k < 32, LargeInt<2> / NativeInt128 for 32 <= k < 64 (gatb KSIZE_LIST
 * "32 64 96 128" [RECALLED]).  Held in 128 bits here; W = words of the upstream type, which decides hash1. */
typedef unsigned __int128 kmer_t;
#define KWORDS(k) ((k) >= 32 ? 2u : 1u)
static inline uint64_t rev2bit64(uint64_t x) {
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    return __builtin_bswap64(x) ^ 0xAAAAAAAAAAAAAAAAULL;
}
static inline kmer_t revcomp_k(kmer_t x, uint32_t k) {
    kmer_t r = ((kmer_t)rev2bit64((uint64_t)x) << 64) | rev2bit64((uint64_t)(x >> 64));
    return r >> (128 - 2 * k);
}
static inline kmer_t canonical_k(kmer_t x, uint32_t k) { kmer_t r = revcomp_k(x, k); return r < x ? r : x; }
static inline kmer_t kmask(uint32_t nbases) { return nbases >= 64 ? ~(kmer_t)0 : (((kmer_t)1) << (2 * nbases)) - 1; }
static inline kmer_t kload(const uint64_t* w, uint32_t W) { return W == 2 ? (((kmer_t)w[1] << 64) | w[0]) : (kmer_t)w[0]; }
static inline void kstore(uint64_t* w, uint32_t W, kmer_t x) { w[0] = (uint64_t)x; if (W == 2) w[1] = (uint64_t)(x >> 64); }

/* LargeInt/NativeInt64 revcomp: reverse the 2-bit groups, complement = code ^ 2 */
uint64_t lo_revcomp(uint64_t x, uint32_t k) {
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = __builtin_bswap64(x);
    x ^= 0xAAAAAAAAAAAAAAAAULL;
    return x >> (64 - 2 * k);
}
uint64_t lo_canonical(uint64_t x, uint32_t k) {
    uint64_t r = lo_revcomp(x, k);
    return r < x ? r : x;
}

/* NativeInt64::hash64 / hash1(LargeInt<1>) */
uint64_t lo_hash64(uint64_t key, uint64_t seed) {
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

/* hash1(LargeInt<precision>): XOR of hash64 over the 64-bit chunks of the TYPE (both chunks for W = 2, even if 0) */
static inline uint64_t hash1_k(kmer_t x, uint32_t W, uint64_t seed) {
    uint64_t h = lo_hash64((uint64_t)x, seed);
    if (W == 2) h ^= lo_hash64((uint64_t)(x >> 64), seed);
    return h;
}

/* HashFunctors::generate_hash_seed (Bloom.hpp), user_seed = 0 */
uint64_t lo_hash_seed(uint32_t idx) {
    static const uint64_t rbase[10] = {
        0xAAAAAAAA55555555ULL, 0x33333333CCCCCCCCULL, 0x6666666699999999ULL, 0xB5B5B5B54B4B4B4BULL,
        0xAA55AA5555335533ULL, 0x33CC33CCCC66CC66ULL, 0x6699669999B599B5ULL, 0xB54BB54B4BAA4BAAULL,
        0xAA33AA3355CC55CCULL, 0x33663366CC99CC99ULL };
    uint64_t tab[10];
    for (int i = 0; i < 10; i++) tab[i] = rbase[i];
    for (int i = 0; i < 10; i++) tab[i] = tab[i] * tab[(i + 3) % 10];
    return tab[idx % 10];
}

/*
 * simplehash16's 256-entry table `random_values` (upstream: 256 fixed 64-bit constants).
 * The constants cannot be recalled; this build DEFINES them as the splitmix64 stream seeded with
 * 0x4C454F4E ("LEON").  Only the low block_nbits (12) bits of each entry reach the bloom.  A
 * maintainer with gatb-core at hand replaces this table (and the product's copy, passed through
 * leon_dna_cfg.random_values) to restore bit parity.
 */
uint64_t lo_random_value(uint32_t idx) {
    uint64_t s = 0x4C454F4EULL + (uint64_t)(idx + 1) * 0x9E3779B97F4A7C15ULL;
    s = (s ^ (s >> 30)) * 0xBF58476D1CE4E5B9ULL;
    s = (s ^ (s >> 27)) * 0x94D049BB133111EBULL;
    return s ^ (s >> 31);
}
static uint64_t RV[256];
static int rv_ready = 0;
static void rv_init(void) {
    if (rv_ready) return;
    for (uint32_t i = 0; i < 256; i++) RV[i] = lo_random_value(i);
    rv_ready = 1;
}
/* NativeInt64::simplehash16_64 */
static inline uint64_t simplehash16(uint64_t key, int shift) {
    uint64_t input = key >> shift;
    uint64_t res = RV[input & 255];
    input >>= 8;
    res ^= RV[input & 255];
    return res;
}

/* ------------------------------------------------------------------------------------------ */
/* Bloom.hpp: Bloom -> BloomCacheCoherent -> BloomNeighborCoherent                             */
/* ------------------------------------------------------------------------------------------ */
struct lo_bloom {
    uint8_t* blooma;
    uint64_t tai, nchar, reduced_tai, mask_block;
    uint32_t k, n_hash, block_nbits, W;
    kmer_t maskkm2, kmer_mask;
    uint64_t seed0;
};
static const uint8_t bit_mask[8] = { 0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80 };
static const uint8_t cano2[16] = { 0, 1, 2, 3, 4, 5, 3, 7, 8, 9, 0, 4, 9, 13, 1, 5 };

lo_bloom* lo_bloom_new(uint64_t tai_bloom, uint32_t k, uint32_t n_hash, uint32_t block_nbits) {
    if (k < 3 || k > 63 || n_hash < 1 || n_hash > 10 || block_nbits < 4 || block_nbits > 16) return NULL;
    if (tai_bloom == 0 || tai_bloom > (1ULL << 46)) return NULL;   /* the modulus tai - 2*blk would not be positive */
    rv_init();
    lo_bloom* b = (lo_bloom*)calloc(1, sizeof(*b));
    /* BloomCacheCoherent ctor: Bloom(tai_bloom + 2*(1<<block_nbits), nbHash) */
    uint64_t tai = tai_bloom + 2 * (1ULL << block_nbits);
    b->nchar = 1 + tai / 8;                                   /* BloomContainer ctor */
    b->blooma = (uint8_t*)calloc(b->nchar + 16, 1);           /* +16: slack for word-wide readers */
    if ((tai & (tai - 1)) == 0) tai--;                        /* isSizePowOf2 => tai-- */
    b->tai = tai;
    b->k = k; b->n_hash = n_hash; b->block_nbits = block_nbits;
    b->mask_block = (1ULL << block_nbits) - 1;
    b->reduced_tai = b->tai - 2 * (1ULL << block_nbits);
    b->W = KWORDS(k);
    b->maskkm2 = kmask(k - 2);
    b->kmer_mask = kmask(k);
    b->seed0 = lo_hash_seed(0);
    return b;
}
void lo_bloom_free(lo_bloom* b) { if (b) { free(b->blooma); free(b); } }
uint8_t* lo_bloom_bits(lo_bloom* b) { return b->blooma; }
uint64_t lo_bloom_nbytes(const lo_bloom* b) { return b->nchar; }
uint64_t lo_bloom_tai(const lo_bloom* b) { return b->tai; }
uint64_t lo_bloom_reduced_tai(const lo_bloom* b) { return b->reduced_tai; }

/* positions of one k-mer: BloomNeighborCoherent::insert / contains */
static void bloom_positions(const lo_bloom* b, kmer_t item, uint64_t* pos) {
    uint32_t k = b->k;
    uint64_t suffix = (uint64_t)(item & 3);
    uint64_t prefix = (uint64_t)((item >> ((k - 1) * 2)) & 3);
    uint64_t pv = cano2[(prefix << 2) + suffix];
    kmer_t hashpart = (item >> 2) & b->maskkm2;
    kmer_t rev = revcomp_k(hashpart, k - 2);
    if (rev < hashpart) hashpart = rev;
    uint64_t racine = hash1_k(hashpart, b->W, b->seed0) % b->reduced_tai;
    uint64_t h0 = racine + (pv & b->mask_block);
    pos[0] = h0;
    /* simplehash16(LargeInt<precision>) looks at value[0] only */
    for (uint32_t i = 1; i < b->n_hash; i++) pos[i] = h0 + (simplehash16((uint64_t)hashpart, (int)i) & b->mask_block);
}
void lo_bloom_insert(lo_bloom* b, const uint64_t* kmers, uint64_t n) {       /* W words per k-mer */
    uint64_t pos[16];
    for (uint64_t j = 0; j < n; j++) {
        bloom_positions(b, kload(kmers + j * b->W, b->W), pos);
        for (uint32_t i = 0; i < b->n_hash; i++) b->blooma[pos[i] >> 3] |= bit_mask[pos[i] & 7];
    }
}
static int bloom_contains_k(const lo_bloom* b, kmer_t kmer) {
    uint64_t pos[16];
    bloom_positions(b, kmer, pos);
    for (uint32_t i = 0; i < b->n_hash; i++)
        if ((b->blooma[pos[i] >> 3] & bit_mask[pos[i] & 7]) == 0) return 0;
    return 1;
}
int lo_bloom_contains(const lo_bloom* b, uint64_t kmer) { return bloom_contains_k(b, (kmer_t)kmer); }
int lo_bloom_contains_w(const lo_bloom* b, const uint64_t* words) { return bloom_contains_k(b, kload(words, b->W)); }
/* BloomNeighborCoherent::contains4: bit nt of the result <=> neighbour with base code nt present */
static unsigned bloom_contains4_k(const lo_bloom* b, kmer_t item, int right) {
    uint32_t k = b->k;
    kmer_t elem = right ? ((item << 2) & b->kmer_mask) : (item >> 2);
    kmer_t hashpart = (elem >> 2) & b->maskkm2;
    kmer_t rev = revcomp_k(hashpart, k - 2);
    if (rev < hashpart) hashpart = rev;
    uint64_t racine = hash1_k(hashpart, b->W, b->seed0) % b->reduced_tai;
    uint64_t keys[16];
    for (uint32_t i = 1; i < b->n_hash; i++) keys[i] = simplehash16((uint64_t)hashpart, (int)i) & b->mask_block;
    unsigned res = 0;
    for (uint64_t nt = 0; nt < 4; nt++) {
        kmer_t tmp = right ? (elem + nt) : (elem + ((kmer_t)nt << ((k - 1) * 2)));
        uint64_t suffix = (uint64_t)(tmp & 3);
        uint64_t prefix = (uint64_t)((tmp >> ((k - 1) * 2)) & 3);
        uint64_t h0 = racine + (cano2[(prefix << 2) + suffix] & b->mask_block);
        int ok = (b->blooma[h0 >> 3] & bit_mask[h0 & 7]) != 0;
        for (uint32_t i = 1; ok && i < b->n_hash; i++) {
            uint64_t h1 = h0 + keys[i];
            if ((b->blooma[h1 >> 3] & bit_mask[h1 & 7]) == 0) ok = 0;
        }
        if (ok) res |= 1u << nt;
    }
    return res;
}
unsigned lo_bloom_contains4(const lo_bloom* b, uint64_t item, int right) { return bloom_contains4_k(b, (kmer_t)item, right); }
unsigned lo_bloom_contains4_w(const lo_bloom* b, const uint64_t* words, int right) { return bloom_contains4_k(b, kload(words, b->W), right); }

/* ------------------------------------------------------------------------------------------ */
/* RangeCoder.cpp: Order0Model, RangeEncoder, RangeDecoder                                     */
/* ------------------------------------------------------------------------------------------ */
#define RC_TOP       (1ULL << 56)
#define RC_BOTTOM    (1ULL << 48)
#define RC_MAX_RANGE RC_BOTTOM

typedef struct { uint32_t n; uint64_t r[257]; } o0model;          /* _charRanges, size n+1 */

static void m_clear(o0model* m) { for (uint32_t i = 0; i <= m->n; i++) m->r[i] = i; }
static void m_init(o0model* m, uint32_t n) { m->n = n; m_clear(m); }
static void m_rescale(o0model* m) {
    for (uint32_t i = 1; i <= m->n; i++) {
        m->r[i] /= 2;
        if (m->r[i] <= m->r[i - 1]) m->r[i] = m->r[i - 1] + 1;
    }
}
static void m_update(o0model* m, uint8_t c) {
    for (uint32_t i = (uint32_t)c + 1; i <= m->n; i++) m->r[i] += 1;
    if (m->r[m->n] >= RC_MAX_RANGE) m_rescale(m);
}

typedef struct { uint64_t low, range; bytevec buf; uint64_t n_sym; } rcenc;

static void enc_clear(rcenc* e) { e->low = 0; e->range = (uint64_t)-1; e->buf.n = 0; }
static void enc_encode(rcenc* e, o0model* m, uint8_t c) {
    e->range /= m->r[m->n];
    e->low += m->r[c] * e->range;
    e->range *= m->r[c + 1] - m->r[c];
    while ((e->low ^ (e->low + e->range)) < RC_TOP ||
           (e->range < RC_BOTTOM && ((e->range = (0 - e->low) & (RC_BOTTOM - 1)), 1))) {
        bv_push(&e->buf, (uint8_t)(e->low >> 56));
        e->range <<= 8;
        e->low <<= 8;
    }
    m_update(m, c);
    e->n_sym++;
}
static void enc_flush(rcenc* e) {
    for (int i = 0; i < 8; i++) { bv_push(&e->buf, (uint8_t)(e->low >> 56)); e->low <<= 8; }
}

typedef struct { uint64_t low, range, code; const uint8_t* p; uint64_t n, i; uint32_t past; } rcdec;

static inline uint8_t dec_byte(rcdec* d) { return d->i < d->n ? d->p[d->i++] : 0; }
static void dec_init(rcdec* d, const uint8_t* p, uint64_t n) {
    d->low = 0; d->range = (uint64_t)-1; d->code = 0; d->p = p; d->n = n; d->i = 0; d->past = 0;
    for (int i = 0; i < 8; i++) d->code = (d->code << 8) | dec_byte(d);
}
static uint8_t dec_next(rcdec* d, o0model* m) {
    d->range /= m->r[m->n];
    if (d->range == 0) { d->range = 1; d->past = 1000; }       /* not a stream the encoder wrote: callers see garbage, never a trap */
    uint64_t value = (d->code - d->low) / d->range;
    int c = (int)m->n - 1;
    while (c > 0 && m->r[c] > value) c--;
    d->low += m->r[c] * d->range;
    d->range *= m->r[c + 1] - m->r[c];
    while ((d->low ^ (d->low + d->range)) < RC_TOP ||
           (d->range < RC_BOTTOM && ((d->range = (0 - d->low) & (RC_BOTTOM - 1)), 1))) {
        d->code = (d->code << 8) | dec_byte(d);
        d->range <<= 8;
        d->low <<= 8;
        if (d->i >= d->n && ++d->past > 64) break;             /* far past the end (a range of 0 would spin here) */
    }
    m_update(m, (uint8_t)c);
    return (uint8_t)c;
}

/* ------------------------------------------------------------------------------------------ */
/* CompressionUtils.hpp                                                                        */
/* ------------------------------------------------------------------------------------------ */
#define NB_MODELS_PER_NUMERIC 9           /* byte-count model + one model per byte index */
typedef struct { o0model m[NB_MODELS_PER_NUMERIC]; } nummodel;

static void nm_init(nummodel* n) { for (int i = 0; i < NB_MODELS_PER_NUMERIC; i++) m_init(&n->m[i], 256); }
static void nm_clear(nummodel* n) { for (int i = 0; i < NB_MODELS_PER_NUMERIC; i++) m_clear(&n->m[i]); }

static int byte_count(uint64_t v) {
    int n = 1;
    while (n < 8 && (v >> (8 * n)) != 0) n++;
    return n;
}
static void encode_numeric(rcenc* e, nummodel* nm, uint64_t value) {
    int bc = byte_count(value);
    enc_encode(e, &nm->m[0], (uint8_t)bc);
    for (int i = 0; i < bc; i++) enc_encode(e, &nm->m[i + 1], (uint8_t)((value >> (i * 8)) & 0xff));
}
static uint64_t decode_numeric(rcdec* d, nummodel* nm) {
    int bc = dec_next(d, &nm->m[0]);
    if (bc > 8) bc = 8;
    uint64_t v = 0;
    for (int i = 0; i < bc; i++) v |= (uint64_t)dec_next(d, &nm->m[i + 1]) << (i * 8);
    return v;
}
/* getDeltaValue: 0 raw, 1 value = prev + delta, 2 value = prev - delta */
static uint8_t get_delta(uint64_t value, uint64_t prev, uint64_t* out) {
    if (value > prev) { uint64_t d = value - prev; if (d < value) { *out = d; return 1; } }
    else              { uint64_t d = prev - value; if (d < value) { *out = d; return 2; } }
    *out = value;
    return 0;
}
static uint64_t from_delta(uint8_t type, uint64_t prev, uint64_t delta) {
    return type == 0 ? delta : (type == 1 ? prev + delta : prev - delta);
}

/* ------------------------------------------------------------------------------------------ */
/* AbstractDnaCoder: models + startBlock                                                       */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    o0model readType, noAnchorRead, bifurcation, bifurcationBinary;
    o0model readSizeDeltaType, anchorPosDeltaType, anchorAddressDeltaType, readAnchorRevcomp;
    nummodel anchorAddress, anchorPos, noAnchorReadSize, readSize, Npos, leftErrorPos, numeric, leftError;
    uint64_t prevReadSize, prevAnchorPos, prevAnchorAddress;
} dnamodels;

static void dm_init(dnamodels* d) {
    m_init(&d->readType, 2); m_init(&d->noAnchorRead, 5); m_init(&d->bifurcation, 5);
    m_init(&d->bifurcationBinary, 2); m_init(&d->readSizeDeltaType, 3); m_init(&d->anchorPosDeltaType, 3);
    m_init(&d->anchorAddressDeltaType, 3); m_init(&d->readAnchorRevcomp, 2);
    nm_init(&d->anchorAddress); nm_init(&d->anchorPos); nm_init(&d->noAnchorReadSize); nm_init(&d->readSize);
    nm_init(&d->Npos); nm_init(&d->leftErrorPos); nm_init(&d->numeric); nm_init(&d->leftError);
    d->prevReadSize = d->prevAnchorPos = d->prevAnchorAddress = 0;
}
static void dm_start_block(dnamodels* d) {                    /* AbstractDnaCoder::startBlock */
    m_clear(&d->readType); m_clear(&d->noAnchorRead); m_clear(&d->bifurcation);
    m_clear(&d->bifurcationBinary); m_clear(&d->readSizeDeltaType); m_clear(&d->anchorPosDeltaType);
    m_clear(&d->anchorAddressDeltaType); m_clear(&d->readAnchorRevcomp);
    nm_clear(&d->anchorAddress); nm_clear(&d->anchorPos); nm_clear(&d->noAnchorReadSize); nm_clear(&d->readSize);
    nm_clear(&d->Npos); nm_clear(&d->leftErrorPos); nm_clear(&d->numeric); nm_clear(&d->leftError);
    d->prevReadSize = d->prevAnchorPos = d->prevAnchorAddress = 0;
}

/* AbstractDnaCoder::codeSeedBin */
static inline kmer_t code_seed(kmer_t kmer, int nt, int right, uint32_t k) {
    if (right) return ((kmer << 2) | (kmer_t)nt) & kmask(k);
    return (kmer >> 2) | ((kmer_t)nt << (2 * (k - 1)));
}

/* ------------------------------------------------------------------------------------------ */
/* anchor dictionary: Leon::_anchorKmers (Hash16) -- here open addressing                     */
/* ------------------------------------------------------------------------------------------ */
typedef struct { kmer_t* keys; uint32_t* vals; uint64_t cap, n; } amap;
#define AMAP_EMPTY (~(kmer_t)0)
static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static void amap_alloc(amap* m, uint64_t cap) {
    m->cap = cap; m->n = 0;
    m->keys = (kmer_t*)malloc(cap * sizeof(kmer_t)); m->vals = (uint32_t*)malloc(cap * 4);
    for (uint64_t i = 0; i < cap; i++) m->keys[i] = AMAP_EMPTY;
}
static inline uint64_t khash(kmer_t key) { return mix64((uint64_t)key ^ mix64((uint64_t)(key >> 64) + 0x9E3779B97F4A7C15ULL)); }
static int amap_get(const amap* m, kmer_t key, uint32_t* val) {
    uint64_t i = khash(key) & (m->cap - 1);
    while (m->keys[i] != AMAP_EMPTY) {
        if (m->keys[i] == key) { *val = m->vals[i]; return 1; }
        i = (i + 1) & (m->cap - 1);
    }
    return 0;
}
static void amap_put(amap* m, kmer_t key, uint32_t val) {
    if ((m->n + 1) * 2 > m->cap) {
        amap old = *m;
        amap_alloc(m, old.cap * 2);
        for (uint64_t i = 0; i < old.cap; i++) if (old.keys[i] != AMAP_EMPTY) amap_put(m, old.keys[i], old.vals[i]);
        free(old.keys); free(old.vals);
    }
    uint64_t i = khash(key) & (m->cap - 1);
    while (m->keys[i] != AMAP_EMPTY) i = (i + 1) & (m->cap - 1);
    m->keys[i] = key; m->vals[i] = val; m->n++;
}

/* ------------------------------------------------------------------------------------------ */
/* DnaEncoder + the Leon members it calls back into                                            */
/* ------------------------------------------------------------------------------------------ */
struct lo_encoder {
    uint32_t k, rpb, W;
    const lo_bloom* bloom;
    dnamodels dm;
    rcenc rc;
    uint64_t processed;                   /* _processedSequenceCount */
    uint64_t n_reads;
    /* finished blocks */
    bytevec blocks; uint64_t* blk_off; uint64_t* blk_size; uint32_t* blk_nreads; uint64_t n_blocks, blk_cap;
    /* anchor dictionary (Leon::_anchorRangeEncoder, _anchorDictModel, _anchorKmers) */
    amap anchors; rcenc arc; o0model anchorDictModel; uint64_t n_anchors;
    uint64_t* anchor_kmers; uint64_t ak_cap;
    /* traces */
    int32_t* tr_pos; uint32_t* tr_addr; uint8_t* tr_flags; uint64_t tr_cap;
    bytevec events;
    /* scratch */
    kmer_t* kmers; char* seq; uint32_t* Npos; uint32_t* errPos; uint8_t* bifVal; uint8_t* bifType;
    uint32_t scratch_cap;
    int finished;
};

lo_encoder* lo_encoder_new(uint32_t k, uint32_t rpb, const lo_bloom* bloom) {
    if (!bloom || bloom->k != k || rpb == 0) return NULL;
    lo_encoder* e = (lo_encoder*)calloc(1, sizeof(*e));
    e->k = k; e->rpb = rpb; e->bloom = bloom; e->W = KWORDS(k);
    dm_init(&e->dm);
    enc_clear(&e->rc); enc_clear(&e->arc);
    m_init(&e->anchorDictModel, 5);
    amap_alloc(&e->anchors, 1024);
    return e;
}
void lo_encoder_free(lo_encoder* e) {
    if (!e) return;
    free(e->rc.buf.p); free(e->arc.buf.p); free(e->blocks.p); free(e->blk_off); free(e->blk_size);
    free(e->blk_nreads); free(e->anchors.keys); free(e->anchors.vals); free(e->anchor_kmers);
    free(e->tr_pos); free(e->tr_addr); free(e->tr_flags); free(e->events.p);
    free(e->kmers); free(e->seq); free(e->Npos); free(e->errPos); free(e->bifVal); free(e->bifType);
    free(e);
}

static void scratch_reserve(lo_encoder* e, uint32_t len) {
    if (len <= e->scratch_cap) return;
    uint32_t c = len + 64;
    e->kmers = (kmer_t*)realloc(e->kmers, (size_t)c * sizeof(kmer_t));
    e->seq = (char*)realloc(e->seq, c);
    e->Npos = (uint32_t*)realloc(e->Npos, (size_t)c * 4);
    e->errPos = (uint32_t*)realloc(e->errPos, (size_t)c * 4);
    e->bifVal = (uint8_t*)realloc(e->bifVal, c);
    e->bifType = (uint8_t*)realloc(e->bifType, c);
    e->scratch_cap = c;
}

/* DnaEncoder::writeBlock -> Leon::writeBlock */
static void write_block(lo_encoder* e) {
    if (e->processed == 0) return;
    enc_flush(&e->rc);
    if (e->n_blocks == e->blk_cap) {
        e->blk_cap = e->blk_cap ? e->blk_cap * 2 : 64;
        e->blk_off = (uint64_t*)realloc(e->blk_off, e->blk_cap * 8);
        e->blk_size = (uint64_t*)realloc(e->blk_size, e->blk_cap * 8);
        e->blk_nreads = (uint32_t*)realloc(e->blk_nreads, e->blk_cap * 4);
    }
    e->blk_off[e->n_blocks] = e->blocks.n;
    e->blk_size[e->n_blocks] = e->rc.buf.n;
    e->blk_nreads[e->n_blocks] = (uint32_t)e->processed;
    bv_append(&e->blocks, e->rc.buf.p, e->rc.buf.n);
    e->n_blocks++;
    enc_clear(&e->rc);
    dm_start_block(&e->dm);
    e->processed = 0;
}

/* Leon::encodeInsertedAnchor: kmer.toString(k), one symbol per base on _anchorDictModel */
static void encode_inserted_anchor(lo_encoder* e, kmer_t kmer) {
    for (uint32_t i = 0; i < e->k; i++) {
        int nt = (int)((kmer >> (2 * (e->k - 1 - i))) & 3);
        enc_encode(&e->arc, &e->anchorDictModel, (uint8_t)nt);
    }
    if (e->n_anchors == e->ak_cap) {
        e->ak_cap = e->ak_cap ? e->ak_cap * 2 : 1024;
        e->anchor_kmers = (uint64_t*)realloc(e->anchor_kmers, e->ak_cap * 8 * e->W);
    }
    kstore(e->anchor_kmers + e->n_anchors * e->W, e->W, kmer);          /* W words per anchor */
}

/* Leon::findAndInsertAnchor: scan [n/2, n/2+10), then [0, n/2), then [n/2+10, n) */
static int find_and_insert_anchor(lo_encoder* e, uint32_t nk, uint32_t* addr) {
    int iMin = (int)nk / 2, iMax = (int)nk / 2 + 10;
    if (iMax > (int)nk) iMax = (int)nk;
    int lo[3] = { iMin, 0, iMax }, hi[3] = { iMax, iMin, (int)nk };
    for (int s = 0; s < 3; s++)
        for (int i = lo[s]; i < hi[s]; i++) {
            kmer_t kmin = canonical_k(e->kmers[i], e->k);
            if (bloom_contains_k(e->bloom, kmin)) {
                encode_inserted_anchor(e, kmin);
                amap_put(&e->anchors, kmin, (uint32_t)e->n_anchors);
                *addr = (uint32_t)e->n_anchors;
                e->n_anchors++;
                return i;
            }
        }
    return -1;
}

/* DnaEncoder::encodeNoAnchorRead */
static void encode_no_anchor_read(lo_encoder* e, const char* orig, uint32_t len) {
    enc_encode(&e->rc, &e->dm.readType, 1);
    encode_numeric(&e->rc, &e->dm.noAnchorReadSize, len);
    for (uint32_t i = 0; i < len; i++) enc_encode(&e->rc, &e->dm.noAnchorRead, (uint8_t)nt2bin(orig[i]));
}

int lo_encoder_add_read(lo_encoder* e, const char* orig, uint32_t len) {
    if (e->finished) return -1;
    uint32_t k = e->k;
    scratch_reserve(e, len);
    if (e->n_reads == e->tr_cap) {
        e->tr_cap = e->tr_cap ? e->tr_cap * 2 : 1024;
        e->tr_pos = (int32_t*)realloc(e->tr_pos, e->tr_cap * 4);
        e->tr_addr = (uint32_t*)realloc(e->tr_addr, e->tr_cap * 4);
        e->tr_flags = (uint8_t*)realloc(e->tr_flags, e->tr_cap);
    }
    bv_reserve(&e->events, len);
    uint8_t* ev = e->events.p + e->events.n;
    memset(ev, 0, len);
    e->events.n += len;
    int32_t tpos = -1; uint32_t taddr = 0; uint8_t tflags = 0;

    if (len < k) {                                            /* DnaEncoder::execute */
        encode_no_anchor_read(e, orig, len);
        goto end_read;
    }
    /* buildKmers: N -> 'A', positions remembered */
    uint32_t nN = 0;
    for (uint32_t i = 0; i < len; i++) {
        char c = orig[i];
        if (nt2bin(c) == 4) { e->Npos[nN++] = i; c = 'A'; }
        e->seq[i] = c;
    }
    uint32_t nk = len - k + 1;
    {
        kmer_t km = 0, mask = kmask(k);
        for (uint32_t i = 0; i < len; i++) {
            km = ((km << 2) | (kmer_t)nt2bin(e->seq[i])) & mask;
            if (i + 1 >= k) e->kmers[i + 1 - k] = km;
        }
    }
    /* findExistingAnchor */
    int anchorPos = -1; uint32_t anchorAddress = 0;
    for (uint32_t i = 0; i < nk; i++)
        if (amap_get(&e->anchors, canonical_k(e->kmers[i], k), &anchorAddress)) { anchorPos = (int)i; break; }
    if (anchorPos == -1) {
        anchorPos = find_and_insert_anchor(e, nk, &anchorAddress);
        if (anchorPos != -1) tflags |= 2;
    }
    if (anchorPos == -1) { encode_no_anchor_read(e, orig, len); goto end_read; }

    /* ---- encodeAnchorRead ---- */
    {
        rcenc* rc = &e->rc; dnamodels* dm = &e->dm;
        uint64_t dv; uint8_t dt;
        enc_encode(rc, &dm->readType, 0);
        dt = get_delta(len, dm->prevReadSize, &dv);
        enc_encode(rc, &dm->readSizeDeltaType, dt); encode_numeric(rc, &dm->readSize, dv);
        dm->prevReadSize = len;
        dt = get_delta((uint64_t)anchorPos, dm->prevAnchorPos, &dv);
        enc_encode(rc, &dm->anchorPosDeltaType, dt); encode_numeric(rc, &dm->anchorPos, dv);
        dm->prevAnchorPos = (uint64_t)anchorPos;
        dt = get_delta(anchorAddress, dm->prevAnchorAddress, &dv);
        enc_encode(rc, &dm->anchorAddressDeltaType, dt); encode_numeric(rc, &dm->anchorAddress, dv);
        dm->prevAnchorAddress = anchorAddress;

        kmer_t anchor = e->kmers[anchorPos];
        int rev = anchor != canonical_k(anchor, k);
        enc_encode(rc, &dm->readAnchorRevcomp, (uint8_t)rev);
        tpos = anchorPos; taddr = anchorAddress; tflags |= (uint8_t)rev;

        uint32_t nBif = 0, nErr = 0;
        for (int dir = 0; dir < 2; dir++) {                   /* left walk, then right walk */
            kmer_t kmer = anchor;
            int pos = dir == 0 ? anchorPos - 1 : anchorPos + (int)k;
            int step = dir == 0 ? -1 : 1;
            for (; pos >= 0 && pos < (int)len; pos += step) {
                /* ---- buildBifurcationList(pos, kmer, right = dir) ---- */
                int nextBin = nt2bin(e->seq[pos]);
                int isN = 0;
                for (uint32_t j = 0; j < nN; j++) if ((int)e->Npos[j] == pos) { isN = 1; break; }
                if (isN) { kmer = code_seed(kmer, nextBin, dir, k); continue; }
                unsigned res4 = bloom_contains4_k(e->bloom, kmer, dir);
                int cnt = 0, first = -1, second = -1, solid = 0;
                for (int nt = 0; nt < 4; nt++)
                    if (res4 & (1u << nt)) {
                        cnt++;
                        if (first < 0) first = nt; else if (second < 0) second = nt;
                        if (nt == nextBin) solid = 1;
                    }
                if (solid && cnt == 1) {                      /* single path, nothing stored */
                    kmer = code_seed(kmer, nextBin, dir, k);
                } else if (solid && cnt == 2) {               /* binary bifurcation */
                    uint8_t b = (first == nextBin) ? 0 : 1;
                    e->bifType[nBif] = 1; e->bifVal[nBif] = b; nBif++;
                    ev[pos] |= (uint8_t)(1 + b);
                    kmer = code_seed(kmer, nextBin, dir, k);
                } else if (!solid && cnt >= 1) {              /* sequencing error: follow the first solid
                                                                 successor (with cnt == 2 the decoder would
                                                                 otherwise read the binary model) */
                    e->errPos[nErr++] = (uint32_t)pos;
                    e->bifType[nBif] = 0; e->bifVal[nBif] = (uint8_t)nextBin; nBif++;
                    ev[pos] |= (uint8_t)(3 + nextBin) | 8;
                    kmer = code_seed(kmer, first, dir, k);
                } else {                                      /* >2 solid successors, or none at all */
                    e->bifType[nBif] = 0; e->bifVal[nBif] = (uint8_t)nextBin; nBif++;
                    ev[pos] |= (uint8_t)(3 + nextBin);
                    kmer = code_seed(kmer, nextBin, dir, k);
                }
            }
        }
        /* N positions */
        encode_numeric(rc, &dm->numeric, nN);
        uint64_t prevN = 0;
        for (uint32_t i = 0; i < nN; i++) { encode_numeric(rc, &dm->Npos, e->Npos[i] - prevN); prevN = e->Npos[i]; }
        /* error positions, sorted ascending */
        encode_numeric(rc, &dm->leftError, nErr);
        for (uint32_t i = 1; i < nErr; i++) {                 /* insertion sort (tiny lists) */
            uint32_t v = e->errPos[i]; int j = (int)i - 1;
            while (j >= 0 && e->errPos[j] > v) { e->errPos[j + 1] = e->errPos[j]; j--; }
            e->errPos[j + 1] = v;
        }
        uint64_t prevE = 0;
        for (uint32_t i = 0; i < nErr; i++) { encode_numeric(rc, &dm->leftErrorPos, e->errPos[i] - prevE); prevE = e->errPos[i]; }
        /* bifurcations in walk order */
        for (uint32_t i = 0; i < nBif; i++) {
            if (e->bifType[i] == 0) enc_encode(rc, &dm->bifurcation, e->bifVal[i]);
            else                    enc_encode(rc, &dm->bifurcationBinary, e->bifVal[i]);
        }
    }
end_read:
    e->tr_pos[e->n_reads] = tpos; e->tr_addr[e->n_reads] = taddr; e->tr_flags[e->n_reads] = tflags;
    e->n_reads++;
    e->processed++;                                           /* endRead */
    if (e->processed >= e->rpb) write_block(e);               /* operator(): writeBlock + startBlock */
    return 0;
}

int lo_encoder_add_reads(lo_encoder* e, const char* bases, const uint64_t* off, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) {
        int rc = lo_encoder_add_read(e, bases + off[i], (uint32_t)(off[i + 1] - off[i]));
        if (rc) return rc;
    }
    return 0;
}
int lo_encoder_finish(lo_encoder* e) {
    if (e->finished) return 0;
    write_block(e);                                           /* ~DnaEncoder: last partial block */
    enc_flush(&e->arc);                                       /* Leon::endDnaCompression */
    e->finished = 1;
    return 0;
}
uint64_t lo_encoder_n_reads(const lo_encoder* e) { return e->n_reads; }
uint64_t lo_encoder_n_blocks(const lo_encoder* e) { return e->n_blocks; }
const uint8_t* lo_encoder_block(const lo_encoder* e, uint64_t i, uint64_t* size, uint32_t* n_reads) {
    if (i >= e->n_blocks) return NULL;
    *size = e->blk_size[i]; *n_reads = e->blk_nreads[i];
    return e->blocks.p + e->blk_off[i];
}
const uint8_t* lo_encoder_anchor_dict(const lo_encoder* e, uint64_t* size, uint64_t* n_anchors) {
    *size = e->arc.buf.n; *n_anchors = e->n_anchors;
    return e->arc.buf.p;
}
const uint64_t* lo_encoder_anchor_kmers(const lo_encoder* e) { return e->anchor_kmers; }
const int32_t*  lo_encoder_read_anchor_pos(const lo_encoder* e) { return e->tr_pos; }
const uint32_t* lo_encoder_read_anchor_addr(const lo_encoder* e) { return e->tr_addr; }
const uint8_t*  lo_encoder_read_flags(const lo_encoder* e) { return e->tr_flags; }
const uint8_t*  lo_encoder_events(const lo_encoder* e, uint64_t* total) { *total = e->events.n; return e->events.p; }
uint64_t lo_encoder_n_symbols(const lo_encoder* e) { return e->rc.n_sym; }

/* ------------------------------------------------------------------------------------------ */
/* DnaDecoder                                                                                  */
/* ------------------------------------------------------------------------------------------ */
int lo_decode_anchor_dict(const uint8_t* payload, uint64_t size, uint64_t n_anchors, uint32_t k, uint64_t* out) {
    rcdec d; o0model m;
    uint32_t W = KWORDS(k);
    dec_init(&d, payload, size);
    m_init(&m, 5);
    for (uint64_t a = 0; a < n_anchors; a++) {
        kmer_t km = 0;
        for (uint32_t i = 0; i < k; i++) km = (km << 2) | (kmer_t)(dec_next(&d, &m) & 3);
        kstore(out + a * W, W, km);                               /* W words per anchor */
    }
    return 0;
}

static int in_list(const uint32_t* l, uint32_t n, int pos) {
    for (uint32_t i = 0; i < n; i++) if ((int)l[i] == pos) return 1;
    return 0;
}

int64_t lo_decode_block(uint32_t k, const lo_bloom* bloom, const uint64_t* anchors, uint64_t n_anchors,
                        const uint8_t* payload, uint64_t size, uint32_t n_reads,
                        char* out, uint64_t out_cap, uint32_t* out_len) {
    rcdec d; dnamodels dm;
    dec_init(&d, payload, size);
    dm_init(&dm);
    uint64_t w = 0;
    uint32_t* Npos = NULL; uint32_t* errPos = NULL; uint32_t lcap = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        uint8_t type = dec_next(&d, &dm.readType);
        if (type == 1) {                                      /* decodeNoAnchorRead */
            uint64_t len = decode_numeric(&d, &dm.noAnchorReadSize);
            if (w + len > out_cap) { free(Npos); free(errPos); return -1; }
            for (uint64_t i = 0; i < len; i++) out[w + i] = BIN2NT[dec_next(&d, &dm.noAnchorRead) % 5];
            out_len[r] = (uint32_t)len; w += len;
            continue;
        }
        uint8_t dt; uint64_t dv;
        dt = dec_next(&d, &dm.readSizeDeltaType); dv = decode_numeric(&d, &dm.readSize);
        uint64_t len = from_delta(dt, dm.prevReadSize, dv); dm.prevReadSize = len;
        dt = dec_next(&d, &dm.anchorPosDeltaType); dv = decode_numeric(&d, &dm.anchorPos);
        uint64_t apos = from_delta(dt, dm.prevAnchorPos, dv); dm.prevAnchorPos = apos;
        dt = dec_next(&d, &dm.anchorAddressDeltaType); dv = decode_numeric(&d, &dm.anchorAddress);
        uint64_t addr = from_delta(dt, dm.prevAnchorAddress, dv); dm.prevAnchorAddress = addr;
        int rev = dec_next(&d, &dm.readAnchorRevcomp);
        if (addr >= n_anchors || apos + k > len || w + len > out_cap) { free(Npos); free(errPos); return -2; }
        if (len + 1 > lcap) {
            lcap = (uint32_t)len + 64;
            Npos = (uint32_t*)realloc(Npos, (size_t)lcap * 4); errPos = (uint32_t*)realloc(errPos, (size_t)lcap * 4);
        }
        uint64_t nN = decode_numeric(&d, &dm.numeric);
        if (nN > len) { free(Npos); free(errPos); return -3; }
        uint64_t prev = 0;
        for (uint64_t i = 0; i < nN; i++) { prev += decode_numeric(&d, &dm.Npos); Npos[i] = (uint32_t)prev; }
        uint64_t nErr = decode_numeric(&d, &dm.leftError);
        if (nErr > len) { free(Npos); free(errPos); return -3; }
        prev = 0;
        for (uint64_t i = 0; i < nErr; i++) { prev += decode_numeric(&d, &dm.leftErrorPos); errPos[i] = (uint32_t)prev; }

        kmer_t anchor = kload(anchors + addr * KWORDS(k), KWORDS(k));
        if (rev) anchor = revcomp_k(anchor, k);
        char* s = out + w;
        for (uint32_t i = 0; i < k; i++) s[apos + i] = BIN2NT[(anchor >> (2 * (k - 1 - i))) & 3];
        for (int dir = 0; dir < 2; dir++) {                   /* DnaDecoder::extendAnchor */
            kmer_t kmer = anchor;
            int pos = dir == 0 ? (int)apos - 1 : (int)apos + (int)k;
            int step = dir == 0 ? -1 : 1;
            for (; pos >= 0 && pos < (int)len; pos += step) {
                if (in_list(Npos, (uint32_t)nN, pos)) {
                    s[pos] = 'N';
                    kmer = code_seed(kmer, 0, dir, k);
                    continue;
                }
                unsigned res4 = bloom_contains4_k(bloom, kmer, dir);
                int cnt = 0, first = -1, second = -1;
                for (int nt = 0; nt < 4; nt++)
                    if (res4 & (1u << nt)) { cnt++; if (first < 0) first = nt; else if (second < 0) second = nt; }
                if (in_list(errPos, (uint32_t)nErr, pos)) {
                    int nt = dec_next(&d, &dm.bifurcation);
                    s[pos] = BIN2NT[nt % 5];
                    kmer = code_seed(kmer, first < 0 ? (nt & 3) : first, dir, k);
                } else if (cnt == 1) {
                    s[pos] = BIN2NT[first];
                    kmer = code_seed(kmer, first, dir, k);
                } else if (cnt == 2) {
                    int nt = dec_next(&d, &dm.bifurcationBinary) == 0 ? first : second;
                    s[pos] = BIN2NT[nt];
                    kmer = code_seed(kmer, nt, dir, k);
                } else {
                    int nt = dec_next(&d, &dm.bifurcation);
                    s[pos] = BIN2NT[nt % 5];
                    kmer = code_seed(kmer, nt & 3, dir, k);
                }
            }
        }
        for (uint64_t i = 0; i < nN; i++) if (Npos[i] < len) s[Npos[i]] = 'N';   /* also inside the anchor */
        out_len[r] = (uint32_t)len; w += len;
    }
    free(Npos); free(errPos);
    return (int64_t)w;
}

/* ------------------------------------------------------------------------------------------ */
/* exact canonical k-mer counting (test stand-in for DSK)                                      */
/* ------------------------------------------------------------------------------------------ */
static int cmp_kmer(const void* a, const void* b) {
    kmer_t x = *(const kmer_t*)a, y = *(const kmer_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
uint64_t lo_count_solid(const char* bases, const uint64_t* off, uint64_t n_reads, uint32_t k,
                        uint32_t min_abundance, uint64_t* out, uint64_t out_cap) {      /* out: W words per k-mer */
    uint64_t total = 0;
    uint32_t W = KWORDS(k);
    for (uint64_t r = 0; r < n_reads; r++) { uint64_t l = off[r + 1] - off[r]; if (l >= k) total += l - k + 1; }
    kmer_t* all = (kmer_t*)malloc((total ? total : 1) * sizeof(kmer_t));
    uint64_t n = 0; kmer_t mask = kmask(k);
    for (uint64_t r = 0; r < n_reads; r++) {
        const char* s = bases + off[r]; uint64_t l = off[r + 1] - off[r];
        kmer_t km = 0; uint32_t valid = 0;
        for (uint64_t i = 0; i < l; i++) {
            int c = nt2bin(s[i]);
            if (c == 4) { valid = 0; km = 0; continue; }       /* DSK skips k-mers containing N */
            km = ((km << 2) | (kmer_t)c) & mask;
            if (++valid >= k) all[n++] = canonical_k(km, k);
        }
    }
    qsort(all, n, sizeof(kmer_t), cmp_kmer);
    uint64_t ns = 0;
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i;
        while (j < n && all[j] == all[i]) j++;
        if (j - i >= min_abundance) { if (out && ns < out_cap) kstore(out + ns * W, W, all[i]); ns++; }
        i = j;
    }
    free(all);
    return ns;
}

/* ------------------------------------------------------------------------------------------ */
/* raw range-coder access                                                                      */
/* ------------------------------------------------------------------------------------------ */
struct lo_rc { rcenc e; };
lo_rc* lo_rc_new(void) { lo_rc* r = (lo_rc*)calloc(1, sizeof(*r)); enc_clear(&r->e); return r; }
void lo_rc_free(lo_rc* r) { if (r) { free(r->e.buf.p); free(r); } }
int lo_rc_encode_stream(lo_rc* r, const uint8_t* models, const uint8_t* syms, uint64_t n,
                        const uint32_t* sizes, uint32_t n_models) {
    o0model* m = (o0model*)malloc(sizeof(o0model) * n_models);
    for (uint32_t i = 0; i < n_models; i++) m_init(&m[i], sizes[i]);
    enc_clear(&r->e);
    for (uint64_t i = 0; i < n; i++) {
        if (models[i] >= n_models || syms[i] >= m[models[i]].n) { free(m); return -1; }
        enc_encode(&r->e, &m[models[i]], syms[i]);
    }
    enc_flush(&r->e);
    free(m);
    return 0;
}
const uint8_t* lo_rc_bytes(const lo_rc* r, uint64_t* size) { *size = r->e.buf.n; return r->e.buf.p; }
int lo_rc_decode_stream(const uint8_t* payload, uint64_t size, const uint8_t* models, uint8_t* out,
                        uint64_t n, const uint32_t* sizes, uint32_t n_models) {
    o0model* m = (o0model*)malloc(sizeof(o0model) * n_models);
    for (uint32_t i = 0; i < n_models; i++) m_init(&m[i], sizes[i]);
    rcdec d; dec_init(&d, payload, size);
    for (uint64_t i = 0; i < n; i++) out[i] = dec_next(&d, &m[models[i]]);
    free(m);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* HeaderCoder.cpp (AbstractHeaderCoder / HeaderEncoder / HeaderDecoder) [RECALLED lo]          */
/*                                                                                            */
/* What is recalled of upstream: headers are coded per read block through the same RangeCoder,  */
/* field by field against the previous header (the FIRST header of the file, stored in plain in */
/* the metadata, at the start of every block: AbstractHeaderCoder::startBlock); a header is    */
/* cut into fields at non-alphanumeric bytes; record types HEADER_END = 1, HEADER_END_MATCH,    */
/* FIELD_ASCII, FIELD_NUMERIC, FIELD_DELTA, FIELD_DELTA_2, FIELD_ZERO_ONLY,                    */
/* FIELD_ZERO_AND_NUMERIC; models _typeModel, _fieldIndexModel, _fieldColumnModel,             */
/* _misSizeModel, _asciiModel, _numericModels (encodeNumeric), _zeroModel.  The exact record    */
/* layout below is this restatement's own definition in that shape;     */
/* like the rest of the oracle it is pinned by round trip and by a second implementation        */
/* (tests/py_header.py), not by reference output.                                               */
/*                                                                                            */
/* Rules.  A field = a maximal (possibly empty) run of [0-9A-Za-z] plus the ONE separator byte  */
/* that follows it (the last field may end with the header); the fields concatenate to the     */
/* header.  Field i of the current header is compared with field i of the previous one; equal   */
/* bytes cost nothing.  A differing (or new) field is one record: type on _typeModel, the      */
/* field index as a count on _fieldIndexModel, then                                            */
/*   token = digits without a leading zero (1..18 digits), separator not NUL:                  */
/*       previous field of the same kind with the same separator -> FIELD_DELTA (value greater,  */
/*       numeric(value - prev)) or FIELD_DELTA_2 (numeric(prev - value));                      */
/*       otherwise FIELD_NUMERIC: numeric(value), separator on _asciiModel (0 = none);         */
/*   token = z >= 2 zeros only -> FIELD_ZERO_ONLY: count(z) on _zeroModel, separator;          */
/*   token = z >= 1 zeros then 1..18 digits -> FIELD_ZERO_AND_NUMERIC: count(z), numeric, sep;  */
/*   anything else -> FIELD_ASCII: count(col) on _fieldColumnModel, col = common prefix with    */
/*       the previous field, count(size - col) on _misSizeModel, the remaining bytes on         */
/*       _asciiModel.                                                                          */
/* count(x) on a 256-symbol model: x < 255 as itself, else 255 followed by numeric(x - 255).    */
/* End of header: HEADER_END_MATCH when it has at least as many fields as the previous one,    */
/* else HEADER_END followed by count(number of fields) on _fieldIndexModel.                     */
/* ------------------------------------------------------------------------------------------ */
enum { HEADER_END = 1, HEADER_END_MATCH, FIELD_ASCII, FIELD_NUMERIC, FIELD_DELTA, FIELD_DELTA_2, FIELD_ZERO_ONLY,
       FIELD_ZERO_AND_NUMERIC, HEADER_TYPE_COUNT };
/* model ids of the (model, symbol) trace: the id space the device range coder is given for this stream */
enum { HM_TYPE = 0, HM_FIELD_INDEX = 8, HM_FIELD_COLUMN = 9, HM_MIS_SIZE = 10, HM_ASCII = 11, HM_ZERO = 12, HM_NUMERIC0 = 13 };

typedef struct {
    o0model type, field_index, field_column, mis_size, ascii, zero;
    nummodel numeric;
} hdrmodels;
static void hm_start_block(hdrmodels* h) {                    /* AbstractHeaderCoder::startBlock */
    m_init(&h->type, HEADER_TYPE_COUNT); m_init(&h->field_index, 256); m_init(&h->field_column, 256);
    m_init(&h->mis_size, 256); m_init(&h->ascii, 256); m_init(&h->zero, 256); nm_init(&h->numeric);
}
typedef struct { const uint8_t* p; uint32_t len, tok; uint8_t sep, has_sep, kind; uint32_t zeros; uint64_t value; } hfield;
enum { HK_ASCII = 0, HK_NUM = 1, HK_ZERO_ONLY = 2, HK_ZERO_NUM = 3 };
static inline int h_isalnum(uint8_t c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }
/* the field starting at h[pos] (pos < len) */
static hfield h_field(const uint8_t* h, uint64_t len, uint64_t pos) {
    hfield f; memset(&f, 0, sizeof f);
    uint64_t e = pos;
    while (e < len && h_isalnum(h[e])) e++;
    f.p = h + pos; f.tok = (uint32_t)(e - pos);
    if (e < len) { f.has_sep = 1; f.sep = h[e]; e++; }
    f.len = (uint32_t)(e - pos);
    f.kind = HK_ASCII;
    int digits = f.tok > 0;
    for (uint32_t i = 0; i < f.tok && digits; i++) digits = f.p[i] >= '0' && f.p[i] <= '9';
    if (digits && !(f.has_sep && f.sep == 0)) {
        uint32_t z = 0;
        while (z < f.tok && f.p[z] == '0') z++;
        if (f.tok == 1 || z == 0) { if (f.tok <= 18) f.kind = HK_NUM; }
        else if (z == f.tok) { f.kind = HK_ZERO_ONLY; f.zeros = z; }
        else if (f.tok - z <= 18) { f.kind = HK_ZERO_NUM; f.zeros = z; }
        if (f.kind == HK_NUM || f.kind == HK_ZERO_NUM)
            for (uint32_t i = f.zeros; i < f.tok; i++) f.value = f.value * 10 + (uint64_t)(f.p[i] - '0');
    }
    return f;
}
typedef struct { rcenc* e; hdrmodels* m; bytevec* trace; } hsink;
static void h_put(hsink* s, o0model* m, uint32_t id, uint8_t c) {
    enc_encode(s->e, m, c);
    if (s->trace) { bv_push(s->trace, (uint8_t)id); bv_push(s->trace, c); }
}
static void h_numeric(hsink* s, uint64_t v) {
    int bc = byte_count(v);
    h_put(s, &s->m->numeric.m[0], HM_NUMERIC0, (uint8_t)bc);
    for (int i = 0; i < bc; i++) h_put(s, &s->m->numeric.m[i + 1], HM_NUMERIC0 + 1 + (uint32_t)i, (uint8_t)((v >> (8 * i)) & 0xff));
}
static void h_count(hsink* s, o0model* m, uint32_t id, uint64_t x) {
    if (x < 255) h_put(s, m, id, (uint8_t)x);
    else { h_put(s, m, id, 255); h_numeric(s, x - 255); }
}
static void h_encode_header(hsink* s, const uint8_t* cur, uint64_t lc, const uint8_t* prev, uint64_t lp) {
    uint64_t pc = 0, pp = 0, i = 0, fprev = 0;
    hdrmodels* M = s->m;
    while (pc < lc) {
        hfield c = h_field(cur, lc, pc);
        int have_p = pp < lp;
        hfield p; memset(&p, 0, sizeof p);
        if (have_p) { p = h_field(prev, lp, pp); pp += p.len; fprev++; }
        if (!(have_p && p.len == c.len && memcmp(p.p, c.p, c.len) == 0)) {
            if (c.kind == HK_NUM) {
                if (have_p && p.kind == HK_NUM && p.has_sep == c.has_sep && p.sep == c.sep && p.value != c.value) {
                    int up = c.value > p.value;
                    h_put(s, &M->type, HM_TYPE, up ? FIELD_DELTA : FIELD_DELTA_2);
                    h_count(s, &M->field_index, HM_FIELD_INDEX, i);
                    h_numeric(s, up ? c.value - p.value : p.value - c.value);
                } else {
                    h_put(s, &M->type, HM_TYPE, FIELD_NUMERIC);
                    h_count(s, &M->field_index, HM_FIELD_INDEX, i);
                    h_numeric(s, c.value);
                    h_put(s, &M->ascii, HM_ASCII, c.has_sep ? c.sep : 0);
                }
            } else if (c.kind == HK_ZERO_ONLY || c.kind == HK_ZERO_NUM) {
                h_put(s, &M->type, HM_TYPE, c.kind == HK_ZERO_ONLY ? FIELD_ZERO_ONLY : FIELD_ZERO_AND_NUMERIC);
                h_count(s, &M->field_index, HM_FIELD_INDEX, i);
                h_count(s, &M->zero, HM_ZERO, c.zeros);
                if (c.kind == HK_ZERO_NUM) h_numeric(s, c.value);
                h_put(s, &M->ascii, HM_ASCII, c.has_sep ? c.sep : 0);
            } else {
                uint32_t col = 0;
                if (have_p) while (col < c.len && col < p.len && c.p[col] == p.p[col]) col++;
                h_put(s, &M->type, HM_TYPE, FIELD_ASCII);
                h_count(s, &M->field_index, HM_FIELD_INDEX, i);
                h_count(s, &M->field_column, HM_FIELD_COLUMN, col);
                h_count(s, &M->mis_size, HM_MIS_SIZE, c.len - col);
                for (uint32_t j = col; j < c.len; j++) h_put(s, &M->ascii, HM_ASCII, c.p[j]);
            }
        }
        pc += c.len; i++;
    }
    while (pp < lp) { hfield p = h_field(prev, lp, pp); pp += p.len; fprev++; }
    if (i >= fprev) h_put(s, &M->type, HM_TYPE, HEADER_END_MATCH);
    else { h_put(s, &M->type, HM_TYPE, HEADER_END); h_count(s, &M->field_index, HM_FIELD_INDEX, i); }
}

/* one block of n headers (text without the leading '>' / '@'); first = the file's first header.
 * payload (malloc'ed, lo_free) and, when trace != NULL, the (model id, symbol) pairs in coding order */
int lo_header_encode_block(const char* headers, const uint64_t* off, uint64_t n, const char* first, uint64_t first_len,
                           uint8_t** payload, uint64_t* size, uint8_t** trace, uint64_t* trace_size) {
    rcenc e; memset(&e, 0, sizeof e); enc_clear(&e);
    hdrmodels* M = (hdrmodels*)malloc(sizeof(hdrmodels));
    hm_start_block(M);
    bytevec tr; memset(&tr, 0, sizeof tr);
    hsink s = { &e, M, trace ? &tr : NULL };
    const uint8_t* prev = (const uint8_t*)first; uint64_t lp = first_len;
    for (uint64_t r = 0; r < n; r++) {
        const uint8_t* cur = (const uint8_t*)headers + off[r]; uint64_t lc = off[r + 1] - off[r];
        h_encode_header(&s, cur, lc, prev, lp);
        prev = cur; lp = lc;
    }
    enc_flush(&e);
    free(M);
    *payload = e.buf.p; *size = e.buf.n;
    if (trace) { *trace = tr.p; *trace_size = tr.n; }
    return 0;
}
void lo_free(void* p) { free(p); }

static uint64_t hd_count(rcdec* d, hdrmodels* M, o0model* m) {
    uint64_t x = dec_next(d, m);
    return x < 255 ? x : 255 + decode_numeric(d, &M->numeric);
}
/* HeaderDecoder over one block: the n headers back to back in out, out_off[n + 1]; returns the bytes written or -1 */
int64_t lo_header_decode_block(const uint8_t* payload, uint64_t size, uint64_t n, const char* first, uint64_t first_len,
                               char* out, uint64_t out_cap, uint64_t* out_off) {
    rcdec d; dec_init(&d, payload, size);
    hdrmodels* M = (hdrmodels*)malloc(sizeof(hdrmodels));
    hm_start_block(M);
    const uint8_t* prev = (const uint8_t*)first; uint64_t lp = first_len;
    uint64_t w = 0;
    int64_t rc = 0;
    out_off[0] = 0;
    for (uint64_t r = 0; r < n && rc == 0; r++) {
        uint64_t start = w, pp = 0, nf = 0;                      /* pp: read cursor in prev, nf: fields written so far */
        #define HD_COPY_PREV_UNTIL(limit)                                                        \
            while (nf < (limit) && pp < lp) { hfield p = h_field(prev, lp, pp);                  \
                if (w + p.len > out_cap) { rc = -1; break; }                                     \
                memcpy(out + w, p.p, p.len); w += p.len; pp += p.len; nf++; }
        for (;;) {
            uint8_t t = dec_next(&d, &M->type);
            if (t == HEADER_END_MATCH) { HD_COPY_PREV_UNTIL(~0ull); break; }
            if (t == HEADER_END) { uint64_t f = hd_count(&d, M, &M->field_index); if (f < nf) rc = -1; else { HD_COPY_PREV_UNTIL(f); if (nf != f) rc = -1; } break; }
            if (t < FIELD_ASCII || t >= HEADER_TYPE_COUNT) { rc = -1; break; }
            uint64_t idx = hd_count(&d, M, &M->field_index);
            if (idx < nf) { rc = -1; break; }
            HD_COPY_PREV_UNTIL(idx);
            if (rc || nf != idx) { rc = -1; break; }
            hfield p; memset(&p, 0, sizeof p);
            int have_p = pp < lp;
            if (have_p) { p = h_field(prev, lp, pp); pp += p.len; }
            char tmp[64]; int tl = 0;
            if (t == FIELD_ASCII) {
                uint64_t col = hd_count(&d, M, &M->field_column), sz = hd_count(&d, M, &M->mis_size);
                if (col > p.len || w + col + sz > out_cap || col + sz < col) { rc = -1; break; }
                if (col) memcpy(out + w, p.p, col);
                w += col;
                for (uint64_t j = 0; j < sz; j++) out[w++] = (char)dec_next(&d, &M->ascii);
            } else {
                uint64_t v = 0, z = 0; uint8_t sep; int has_sep;
                if (t == FIELD_DELTA || t == FIELD_DELTA_2) {
                    uint64_t dv = decode_numeric(&d, &M->numeric);
                    if (!have_p || p.kind != HK_NUM) { rc = -1; break; }
                    v = t == FIELD_DELTA ? p.value + dv : p.value - dv;
                    sep = p.sep; has_sep = p.has_sep;
                } else {
                    if (t != FIELD_NUMERIC) z = hd_count(&d, M, &M->zero);
                    if (t != FIELD_ZERO_ONLY) v = decode_numeric(&d, &M->numeric);
                    sep = dec_next(&d, &M->ascii); has_sep = sep != 0;
                }
                if (t != FIELD_ZERO_ONLY) tl = snprintf(tmp, sizeof tmp, "%llu", (unsigned long long)v);
                if (w + z + (uint64_t)tl + 1 > out_cap || z > out_cap) { rc = -1; break; }
                memset(out + w, '0', z); w += z;
                memcpy(out + w, tmp, (size_t)tl); w += (uint64_t)tl;
                if (has_sep) out[w++] = (char)sep;
            }
            nf++;
        }
        #undef HD_COPY_PREV_UNTIL
        out_off[r + 1] = w;
        prev = (const uint8_t*)out + start; lp = w - start;
    }
    free(M);
    return rc ? rc : (int64_t)w;
}

/* ------------------------------------------------------------------------------------------ */
/* Quality stream, lossy form: DnaEncoder::storeSolidCoverageInfo + smoothQuals [RECALLED med]   */
/* cover[i] = number of the read's k-mers (N read as 'A', canonical) that are in the bloom and   */
/* span position i; qual[i] becomes '@' where cover[i] >= 2 or qual[i] > '@'; reads shorter      */
/* than k are left alone.  qual is rewritten in place.                                           */
/* ------------------------------------------------------------------------------------------ */
void lo_qual_smooth(const lo_bloom* bloom, uint32_t k, const char* seq, uint32_t len, uint8_t* qual) {
    if (len < k) return;
    uint32_t* cover = (uint32_t*)calloc(len, sizeof(uint32_t));
    kmer_t km = 0;
    const kmer_t mask = kmask(k);
    for (uint32_t i = 0; i < len; i++) {
        int c = nt2bin(seq[i]);
        if (c > 3) c = 0;
        km = ((km << 2) | (kmer_t)c) & mask;
        if (i + 1 >= k && bloom_contains_k(bloom, canonical_k(km, k)))
            for (uint32_t j = i + 1 - k; j <= i; j++) cover[j]++;
    }
    for (uint32_t i = 0; i < len; i++)
        if (cover[i] >= 2 || qual[i] > (uint8_t)'@') qual[i] = (uint8_t)'@';
    free(cover);
}

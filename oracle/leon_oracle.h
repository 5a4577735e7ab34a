/*
 * leon_oracle.h -- CPU restatement of Leon's DNA encode path (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED.  The reference's implementation of this path lives in the gatb-core
 * submodule, which is absent from /root/reference.  This oracle restates
 * the algorithm of upstream gatb-core >= 1.4.0 (un-pinned; /root/reference/.gitmodules:1-3,
 * /root/reference/README.md:87) from the published design (README.md:11-13) and from recall of
 * the upstream sources named below.  It is pinned by NO reference golden vector (the reference
 * ships none: SURVEY.md section 4) -- only by round-trip (its own decoder) and by self-golden
 * fixtures under tests/golden/ that are labelled as such.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (leon_amd/, include/) never links, imports or calls it.
 *
 * Upstream files restated (paths relative to gatb-core/gatb-core/src/gatb/, [RECALLED]):
 *   tools/compression/RangeCoder.{hpp,cpp}   Order0Model, RangeEncoder, RangeDecoder
 *   tools/compression/CompressionUtils.hpp   encodeNumeric / decodeNumeric / getDeltaValue
 *   tools/compression/DnaCoder.{hpp,cpp}     AbstractDnaCoder, DnaEncoder, DnaDecoder
 *   tools/compression/HeaderCoder.{hpp,cpp}  AbstractHeaderCoder, HeaderEncoder, HeaderDecoder (row f3)
 *   tools/compression/Leon.{hpp,cpp}         anchorExist, findAndInsertAnchor, encodeInsertedAnchor,
 *                                            nt2bin/bin2nt, READ_PER_BLOCK
 *   tools/collections/impl/Bloom.hpp         HashFunctors, Bloom, BloomCacheCoherent,
 *                                            BloomNeighborCoherent (insert/contains/contains4)
 *   tools/math/NativeInt64.hpp / LargeInt    hash1 (hash64), simplehash16, revcomp
 *   kmer/impl/Model.hpp                      2-bit code A0 C1 T2 G3 = (c>>1)&3
 */
#ifndef LEON_ORACLE_H
#define LEON_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- k-mer helpers.  k <= 31: one 64-bit word per k-mer; 32 <= k <= 63: two words (low, high), as upstream's
 * LargeInt<2>.  Arrays of k-mers hold W = (k >= 32 ? 2 : 1) words per k-mer. ---- */
uint64_t lo_revcomp(uint64_t kmer, uint32_t k);
uint64_t lo_canonical(uint64_t kmer, uint32_t k);
uint64_t lo_hash64(uint64_t key, uint64_t seed);      /* NativeInt64 hash1 */
uint64_t lo_hash_seed(uint32_t idx);                  /* HashFunctors::seed_tab[idx], user_seed 0 */
uint64_t lo_random_value(uint32_t idx);               /* simplehash16 table entry (see .c) */

/* ---- BloomNeighborCoherent ---- */
typedef struct lo_bloom lo_bloom;
lo_bloom* lo_bloom_new(uint64_t tai_bloom, uint32_t k, uint32_t n_hash, uint32_t block_nbits);
void      lo_bloom_free(lo_bloom* b);
void      lo_bloom_insert(lo_bloom* b, const uint64_t* kmers, uint64_t n);
int       lo_bloom_contains(const lo_bloom* b, uint64_t kmer);
unsigned  lo_bloom_contains4(const lo_bloom* b, uint64_t kmer, int right);
int       lo_bloom_contains_w(const lo_bloom* b, const uint64_t* words);
unsigned  lo_bloom_contains4_w(const lo_bloom* b, const uint64_t* words, int right);
uint8_t*  lo_bloom_bits(lo_bloom* b);
uint64_t  lo_bloom_nbytes(const lo_bloom* b);
uint64_t  lo_bloom_tai(const lo_bloom* b);            /* after the power-of-two adjustment */
uint64_t  lo_bloom_reduced_tai(const lo_bloom* b);

/* ---- sequential DNA encoder (DnaEncoder with -nb-cores 1 semantics) ---- */
typedef struct lo_encoder lo_encoder;
lo_encoder* lo_encoder_new(uint32_t k, uint32_t reads_per_block, const lo_bloom* bloom);
void        lo_encoder_free(lo_encoder* e);
/* seq: ASCII bases (A,C,G,T; any other byte is an N).  Reads must be added in file order. */
int         lo_encoder_add_read(lo_encoder* e, const char* seq, uint32_t len);
int         lo_encoder_add_reads(lo_encoder* e, const char* bases, const uint64_t* offsets, uint64_t n);
int         lo_encoder_finish(lo_encoder* e);

uint64_t       lo_encoder_n_reads(const lo_encoder* e);
uint64_t       lo_encoder_n_blocks(const lo_encoder* e);
const uint8_t* lo_encoder_block(const lo_encoder* e, uint64_t i, uint64_t* size, uint32_t* n_reads);
const uint8_t* lo_encoder_anchor_dict(const lo_encoder* e, uint64_t* size, uint64_t* n_anchors);
const uint64_t* lo_encoder_anchor_kmers(const lo_encoder* e);
/* per-read trace, for stage-wise parity checks: anchor position (-1 none), address, flags */
const int32_t*  lo_encoder_read_anchor_pos(const lo_encoder* e);
const uint32_t* lo_encoder_read_anchor_addr(const lo_encoder* e);
const uint8_t*  lo_encoder_read_flags(const lo_encoder* e);      /* bit0 revcomp, bit1 inserted */
/* per-position event bytes of every read, concatenated with the caller's offsets:
 * bits0-2: 0 none, 1 binary-0, 2 binary-1, 3+nt four-ary symbol nt (0..4); bit3: error position */
const uint8_t*  lo_encoder_events(const lo_encoder* e, uint64_t* total);
uint64_t        lo_encoder_n_symbols(const lo_encoder* e);

/* ---- decoder (DnaDecoder) ---- */
int lo_decode_anchor_dict(const uint8_t* payload, uint64_t size, uint64_t n_anchors, uint32_t k,
                          uint64_t* out_kmers);
/* decodes one block; out must hold the decoded bases, out_len[n_reads] receives the lengths */
int64_t lo_decode_block(uint32_t k, const lo_bloom* bloom, const uint64_t* anchors, uint64_t n_anchors,
                        const uint8_t* payload, uint64_t size, uint32_t n_reads,
                        char* out, uint64_t out_cap, uint32_t* out_len);

/* ---- header stream (HeaderEncoder / HeaderDecoder, one read block per call; rules in leon_oracle.c) ----
 * headers: the blocks' header texts back to back (no leading '>' / '@', no newline), off[n + 1]; first: the file's first
 * header (AbstractHeaderCoder::startBlock starts every block from it).  payload / trace are malloc'ed: lo_free. */
int     lo_header_encode_block(const char* headers, const uint64_t* off, uint64_t n, const char* first, uint64_t first_len,
                               uint8_t** payload, uint64_t* size, uint8_t** trace, uint64_t* trace_size);
int64_t lo_header_decode_block(const uint8_t* payload, uint64_t size, uint64_t n, const char* first, uint64_t first_len,
                               char* out, uint64_t out_cap, uint64_t* out_off);
void    lo_free(void* p);
/* ---- quality stream, lossy form (DnaEncoder::smoothQuals): qual rewritten in place ---- */
void    lo_qual_smooth(const lo_bloom* bloom, uint32_t k, const char* seq, uint32_t len, uint8_t* qual);

/* ---- exact k-mer counting helper for tests (stands in for DSK) ---- */
/* returns the number of distinct canonical k-mers with count >= min_abundance; fills out (may be NULL) */
uint64_t lo_count_solid(const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t k,
                        uint32_t min_abundance, uint64_t* out, uint64_t out_cap);

/* ---- raw range coder access for unit tests ---- */
typedef struct lo_rc lo_rc;
lo_rc*   lo_rc_new(void);
void     lo_rc_free(lo_rc* r);
/* encodes syms[i] with model models[i]; model_sizes[m] gives alphabet size of model m */
int      lo_rc_encode_stream(lo_rc* r, const uint8_t* models, const uint8_t* syms, uint64_t n,
                             const uint32_t* model_sizes, uint32_t n_models);
const uint8_t* lo_rc_bytes(const lo_rc* r, uint64_t* size);
int      lo_rc_decode_stream(const uint8_t* payload, uint64_t size, const uint8_t* models, uint8_t* out_syms,
                             uint64_t n, const uint32_t* model_sizes, uint32_t n_models);

#ifdef __cplusplus
}
#endif
#endif

/*
 * leon_dna.h -- C-ABI of the MI355X-native Leon DNA encode path (libleon_dna.so).
 *
 * This is the drop-in boundary for the one hot path SURVEY.md section 8 scopes: what gatb-core's
 * Leon::startDnaCompression does by running Dispatcher::iterate(bank, DnaEncoder(this)) on pthreads
 * (call shape verified at /root/reference/src/main.cpp:44 `Leon().run(argc, argv)`; everything below
 * it lives in the un-vendored gatb-core submodule, /root/reference/.gitmodules:1-3, so the upstream
 * names cited per entry point are [RECALLED] names, not file:line -- SURVEY.md section 0).
 *
 * Conventions: plain pointers and sizes, no C++ or torch types; every call returns 0 on success or a
 * negative LEON_E_* code and never throws; leon_last_error() gives the message.  One ctx = one ordered
 * read stream (one output file); calls on a ctx are serialised by the caller; HIP streams are private.
 * There is NO CPU fallback: without a HIP device ctx_create fails with LEON_E_NO_DEVICE.
 */
#ifndef LEON_DNA_H
#define LEON_DNA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LEON_DNA_ABI_VERSION 5

enum {
    LEON_OK = 0,
    LEON_E_INVALID = -1,     /* bad argument / configuration */
    LEON_E_NO_DEVICE = -2,   /* no usable HIP device (the product has no CPU path) */
    LEON_E_HIP = -3,         /* HIP runtime error, message carries hipGetErrorString */
    LEON_E_STATE = -4,       /* call order violated (e.g. encode after finish), or the stream is poisoned: a batch that
                                failed after it had begun to change the stream (HIP error, internal bound, sink) leaves the
                                dictionary and the caller's block sequence part-way through; every later call on the stream
                                returns this code until leon_dna_reset_stream.  Batches refused for their arguments
                                (LEON_E_INVALID, call order) leave the context untouched. */
    LEON_E_OVERFLOW = -5,    /* an internal device buffer bound was hit (reported, never silently truncated) */
    LEON_E_SINK = -6         /* the block sink returned non-zero */
};

typedef struct leon_dna_ctx leon_dna_ctx;

/* Replaces the state a DnaEncoder copies from its owning Leon*: _kmerSize, Leon::READ_PER_BLOCK and the
 * shared IBloom<kmer_type>* created by Leon::createBloom (BloomFactory BLOOM_NEIGHBOR, 7 hashes). */
typedef struct leon_dna_cfg {
    uint32_t struct_size;        /* sizeof(leon_dna_cfg) */
    uint32_t kmer_size;          /* 3..63.  k-mers cross this ABI as W 64-bit words each, low word first:
                                    W = 1 below 32 (upstream LargeInt<1>), W = 2 from 32 to 63 (LargeInt<2>) */
    uint32_t reads_per_block;    /* Leon::READ_PER_BLOCK, 50000 */
    uint32_t bloom_n_hash;       /* 7 */
    uint32_t bloom_block_nbits;  /* BloomCacheCoherent block_nbits, 12 */
    int32_t  device_id;          /* HIP device ordinal */
    uint64_t bloom_tai;          /* tai_bloom as handed to BloomNeighborCoherent (bits, before its padding) */
    const uint64_t* random_values; /* optional 256-entry simplehash16 table; NULL = built-in */
    uint64_t resolve_window;     /* reads per anchor-resolution window; 0 = default (1<<21) */
    uint32_t flags;              /* LEON_F_* */
    uint32_t reserved;
} leon_dna_cfg;

#define LEON_F_KEEP_TRACE 1u     /* keep per-read anchors / events of the last batch for leon_dna_trace_* */
#define LEON_F_DICT_ON_DEVICE 2u /* code the anchor-dictionary stream with the device range coder (one serial chain on one
                                    workgroup, ~50x slower than the default host thread: DESIGN.md 4.4) instead of the host thread */

/* Replaces Leon::writeBlock(buffer, size, nReads, blockId): called on the calling thread, in increasing
 * block_id; payload is owned by the library and valid until the sink returns.  Non-zero aborts. */
typedef int (*leon_block_sink)(void* user, uint64_t block_id, const uint8_t* payload, uint64_t size,
                               uint32_t n_reads);

typedef struct leon_dna_stats {
    uint64_t n_reads, n_bases, n_blocks, n_anchors, n_no_anchor, n_symbols, payload_bytes;
    uint64_t resolve_rounds, resolve_windows;
    /* HIP-event milliseconds of the last batch, measured on the library's own stream */
    float ms_pack, ms_resolve, ms_sort, ms_walk, ms_symbols, ms_rangecoder, ms_d2h, ms_total;
    uint32_t walk_launches, reserved;
    float ms_anchor_wait, ms_chain_busy;   /* host ms leon_dna_finish waited for the dictionary-stream thread; ms that thread spent coding */
    /* the walk divided by anchor (leon_dna_set_exchange), last batch: ms_walk above is then this rank's SLICE; ms_exchange = forming the
       words + the caller's exchange + scattering what came back (host wall-clock, the call waits for the device around it);
       ms_emulated = LEON_XCH_EMULATE only: the other ranks' slices walked here in their stead (not part of a real rank's time) */
    float ms_exchange, ms_exchange_call, ms_emulated, ms_emulated_lookups;   /* ms_emulated_lookups: the part of ms_emulated (and of ms_resolve) spent on the other ranks' window look-ups */
    uint64_t xch_words_sent, xch_words_received, walk_reads;
    /* anchor resolution, last batch: resolve_rounds above counts the parallel rounds; what they left unsettled (reads each waiting for the
       one before it: files in genome-position order) went through the exact sequential pass -- that many reads, in that many windows,
       ms_resolve_chain of ms_resolve (ABI 5) */
    uint64_t resolve_chain_reads, resolve_chain_windows;
    float ms_resolve_chain, ms_gather_call;   /* ms_gather_call: host ms inside the caller's gather callback (leon_dna_set_gather), part of ms_exchange_call */
} leon_dna_stats;

/* -- lifecycle (DnaEncoder ctor / dtor bracket one thread's work upstream) -- */
int  leon_dna_ctx_create(const leon_dna_cfg* cfg, leon_dna_ctx** out);
void leon_dna_ctx_destroy(leon_dna_ctx* ctx);
const char* leon_last_error(const leon_dna_ctx* ctx);      /* ctx may be NULL: last create error */
int  leon_dna_abi_version(void);

/* -- bloom (probe side is on the path; build side: Leon::createBloom's insert loop) -- */
int leon_dna_bloom_nbytes(const leon_dna_ctx* ctx, uint64_t* nbytes);          /* Bloom::getSize, nchar */
int leon_dna_bloom_upload(leon_dna_ctx* ctx, const uint8_t* bits, uint64_t nbytes);   /* StorageTools::loadBloom */
int leon_dna_bloom_download(leon_dna_ctx* ctx, uint8_t* bits, uint64_t nbytes);       /* StorageTools::saveBloom */
int leon_dna_bloom_clear(leon_dna_ctx* ctx);
int leon_dna_bloom_insert(leon_dna_ctx* ctx, const uint64_t* kmers, uint64_t n);      /* n host k-mers (n*W words), IBloom::insert */
int leon_dna_bloom_insert_device(leon_dna_ctx* ctx, const uint64_t* d_kmers, uint64_t n);
int leon_dna_bloom_device_ptr(leon_dna_ctx* ctx, void** d_bits, uint64_t* nbytes);    /* for an RCCL broadcast */
/* device-to-device forms of upload/download: the bloom travels between GPUs over xGMI, never via the host */
int leon_dna_bloom_upload_device(leon_dna_ctx* ctx, const uint8_t* d_bits, uint64_t nbytes);
int leon_dna_bloom_download_device(leon_dna_ctx* ctx, uint8_t* d_bits, uint64_t nbytes);
/* BloomNeighborCoherent::contains4 / contains over a list of k-mers (host in, host out) */
int leon_dna_bloom_contains4(leon_dna_ctx* ctx, const uint64_t* kmers, uint64_t n, int right, uint8_t* out);
int leon_dna_bloom_contains(leon_dna_ctx* ctx, const uint64_t* kmers, uint64_t n, uint8_t* out);

/* -- the hot path: DnaEncoder::operator()(Sequence&) over a batch of reads in file order --
 * bases: concatenated ASCII read data (A,C,G,T; any other byte is an N); offsets[n_reads+1].
 * first_read_index must continue the stream; every batch but the last must hold a whole number of
 * blocks (n_reads % reads_per_block == 0).  Blocks go to `sink` in increasing block_id. */
int leon_dna_encode_batch(leon_dna_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads,
                          uint64_t first_read_index, leon_block_sink sink, void* user);
/* same with bases/offsets already resident in device memory (HBM) */
int leon_dna_encode_batch_device(leon_dna_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_offsets,
                                 uint64_t n_reads, uint64_t first_read_index, leon_block_sink sink, void* user);

/* Optional: size every per-batch device buffer for batches of up to max_reads reads / max_bases bases ahead of the first
 * batch (a host calls it while it is still parsing the input), so that the first encode of a process allocates nothing
 * large.  Estimates only: whatever a batch needs beyond them is grown on demand. */
int leon_dna_reserve(leon_dna_ctx* ctx, uint64_t max_reads, uint64_t max_bases);

/* N contexts, one per GPU, working on ONE output file: every rank is fed the SAME batches; each resolves the
 * anchors of all reads (replicated, so the dictionary and every read's anchor are file-order exact on every rank
 * without any exchange), then walks and codes only its contiguous share of each batch's blocks, which its sink
 * receives with their global block ids.  Rank 0 alone produces the dictionary stream.  Default: rank 0 of 1. */
int leon_dna_set_shard(leon_dna_ctx* ctx, uint32_t rank, uint32_t world);

/* How a batch's WALK is divided among the ranks of leon_dna_set_shard(rank, world > 1).
 * LEON_XCH_OFF (default): every rank walks the reads of its own block range.  The reads of one anchor are spread over all block
 * ranges, so every rank fetches nearly every anchor group's bloom sectors: the eight ranks' walks of a 100 M-read file add up to
 * 2.8 x the one-GPU walk.
 * LEON_XCH_BY_ANCHOR: the batch's reads, sorted by anchor address, are cut into `world` contiguous slices and rank r walks slice r --
 * whole anchor groups, the sharing that makes the one-GPU walk cheap -- whatever block the reads belong to; what the walk found then
 * travels to the rank that codes each read's block in ONE exchange per batch, which the CALLER performs (the library stays free of
 * RCCL): `fn` is called once per batch, on the calling thread, with this rank's 64-bit words in DEVICE memory grouped by
 * destination rank (send_counts[d] words for rank d, in rank order) and must return, in device memory that stays valid until the
 * next call on this context, the words the other ranks (and this one) hold for this rank, concatenated in any order, with their
 * total in *recv_total -- an all-to-all (ncclAllToAllv / all_to_all_single).  Every rank must call it for every batch, also a rank
 * that codes no block of the batch.  Non-zero return = the batch fails (LEON_E_SINK-like: LEON_E_STATE, stream poisoned).
 * LEON_XCH_EMULATE: the same division with NO other rank present: the context walks every slice itself, one after the other, and
 * keeps the words meant for its own rank -- the bytes of a real run (one-process tests of an N-rank job, timing one seat of it:
 * stats.ms_walk is the own slice, stats.ms_emulated the rest).  fn is ignored.
 * The blocks' bytes are the same in all three modes. */
enum { LEON_XCH_OFF = 0, LEON_XCH_BY_ANCHOR = 1, LEON_XCH_EMULATE = 2 };
typedef int (*leon_exchange_fn)(void* user, const uint64_t* d_send, const uint64_t* send_counts, uint32_t world,
                                const uint64_t** d_recv, uint64_t* recv_total);
int leon_dna_set_exchange(leon_dna_ctx* ctx, uint32_t mode, leon_exchange_fn fn, void* user);

/* The anchor resolution's look-ups (a good half of the stage every rank otherwise repeats for ALL reads) divided among the ranks as
 * well: of every resolution window rank r looks up the r-th run of ceil(window / world) reads, and ONE all-gather per window --
 * again the caller's -- tells every rank what every read's pass found: `fn` is called on the calling thread with a device buffer of
 * world * part_bytes bytes whose part r (at d_buf + r * part_bytes) this rank has filled for r == its own rank, and must return with
 * every part filled by its rank (ncclAllGather / all_gather_into_tensor, in place or through a copy).  Each rank then makes the
 * other runs' results its own (the found key's slot, the proposals' entries in its dictionary), so the dictionaries stay identical.
 * Takes effect with LEON_XCH_BY_ANCHOR and world > 1; LEON_XCH_EMULATE computes the other ranks' runs itself
 * (stats.ms_emulated_lookups).  ~50 calls per 100 M reads, 16 MB each at the default window.  fn == NULL: every rank looks
 * everything up (the default).  Must precede the first batch of a stream.  Non-zero return of fn = the batch fails. */
typedef int (*leon_gather_fn)(void* user, void* d_buf, uint64_t part_bytes, uint32_t world);
int leon_dna_set_gather(leon_dna_ctx* ctx, leon_gather_fn fn, void* user);
/* FAILURE WITH CALLBACKS SET -- a rule for the caller.  The two callbacks put the caller's collectives INSIDE leon_dna_encode_batch*: one
 * all-gather per resolution window, one all-to-all per batch.  A rank whose call fails for a reason of its own (no memory for a buffer,
 * a HIP error, its sink, input the other ranks did not get) returns at once with its error code and does NOT enter the collectives it
 * has not reached: the library cannot "meet" them with empty parts, it does not know the caller's communicator.  The other ranks are
 * then waiting in theirs.  So a caller that sets a callback MUST treat a non-zero return of leon_dna_encode_batch* on any rank as the
 * end of the job on all of them: abort the communicator (ncclCommAbort), or run with a collective timeout and end the process non-zero
 * so that the launcher tears the job down -- bench.py does the latter: `Watch` names the collective that did not complete within
 * LEON_BENCH_COLL_TIMEOUT seconds and exits, a failed rank exits at once.  The context itself is poisoned (LEON_E_STATE) like after any
 * failure part-way through a batch; leon_dna_reset_stream starts a new stream once the communicator is whole again. */

/* Leon::endDnaCompression: flush the anchor-dictionary range coder (Leon::encodeInsertedAnchor stream).
 * payload stays owned by ctx until destroy. */
int leon_dna_finish(leon_dna_ctx* ctx, const uint8_t** dict_payload, uint64_t* dict_size, uint64_t* n_anchors);

/* -- the step BEFORE the path: solid k-mers of the reads, the set Leon::createBloom inserts (upstream DSK,
 * SortingCountAlgorithm).  Canonical k-mers occurring at least min_abundance times (0 = automatic threshold, see leon_kmer_auto_cutoff); k-mers containing an N are skipped.
 * Output: W words per k-mer (unordered across hash partitions).  histogram (optional, 256 entries): number of distinct
 * k-mers by abundance, clipped at 255.  max_keys_per_pass: k-mers sorted at once (0 = sized from the free device memory).  Errors: leon_last_error(NULL). */
int leon_kmer_solid_device(int device_id, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n_reads,
                           uint32_t kmer_size, uint32_t min_abundance, uint64_t max_keys_per_pass,
                           uint64_t** d_solid, uint64_t* n_solid, uint64_t* histogram);   /* *d_solid: leon_device_free */
int leon_kmer_solid(int device_id, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t kmer_size,
                    uint32_t min_abundance, uint64_t max_keys_per_pass, uint64_t* out, uint64_t out_cap, uint64_t* n_solid,
                    uint64_t* histogram);
void leon_device_free(void* d_ptr);
/* leon_kmer_solid_device keeps its large work buffers for its next call instead of freeing them (freed device memory is wiped by the
 * driver before it is handed out again, at the expense of whoever allocates next); leon_dna_reserve returns them once the context's
 * own buffers exist, and so does this. */
void leon_device_trim(void);
/* min_abundance = 0 in the two calls above means "automatic" (Leon's default, /root/reference/README.md:54): the threshold
 * this function derives from the abundance spectrum (first local minimum, never below 2; 2 when there is no valley).
 * histogram: 256 entries as returned above.  Host-only. */
int leon_kmer_auto_cutoff(const uint64_t* histogram, uint32_t* cutoff);
/* plain device memory for callers that keep their reads resident in HBM (the host mirror's -c does): errors through
 * leon_last_error(NULL) */
int leon_device_count(int* n_devices);
int leon_device_alloc(int device_id, uint64_t bytes, void** d_ptr);
int leon_device_upload(int device_id, void* d_dst, const void* src, uint64_t bytes);
int leon_device_copy(int device_id, void* d_dst, const void* d_src, uint64_t bytes);   /* device to device; complete on return */
int leon_device_download(int device_id, void* dst, const void* d_src, uint64_t bytes);

/* Host-only helper (no GPU, no ctx): the dictionary stream leon_dna_finish returns for a given anchor list, i.e.
 * Leon::encodeInsertedAnchor over `kmers` in address order followed by the flush.  Used by the CPU tests. */
int leon_host_anchor_dict_encode(const uint64_t* kmers, uint64_t n_anchors, uint32_t kmer_size, uint8_t* out,
                                 uint64_t out_cap, uint64_t* size);

/* -- the inverse path: DnaDecoder::execute over read blocks, blocks in parallel on the device --
 * Needs the bloom of the compressed file in ctx (leon_dna_bloom_upload).  anchors: the dictionary (n_anchors * W words,
 * from leon_host_anchor_dict_decode).  payloads: the blocks' payloads back to back, payload_off[n_blocks + 1];
 * block_n_reads / block_n_bases: reads and bases per block (the container's block table).  Output: the reads' bases back
 * to back in block order (out_cap >= the sum of block_n_bases) and every read's length (sum of block_n_reads entries).
 * LEON_E_INVALID with a message when a payload does not decode against this bloom / dictionary; out_bases and out_len are then
 * as the caller left them (the call writes them only once every block has decoded).
 * Memory: the context keeps a table of what the decoding waves have learnt from the bloom (16 or 32 bytes x 4 per solid
 * k-mer the bloom was sized for, at most 40 % of the device's free memory; LEON_DC_CACHE_MB in the environment overrides,
 * 0 = none) from call to call, for as long as the bloom's bits do not change and the context does not encode (an encode
 * call returns the table's memory first); it makes the call faster, never different.
 * n_threads of the host functions below: 0 = the CPUs this process may use (its cgroup's quota). */
int leon_dna_decode_blocks(leon_dna_ctx* ctx, const uint64_t* anchors, uint64_t n_anchors, const uint8_t* payloads,
                           const uint64_t* payload_off, const uint32_t* block_n_reads, const uint64_t* block_n_bases,
                           uint64_t n_blocks, uint8_t* out_bases, uint64_t out_cap, uint32_t* out_len);
/* Host-only (no GPU, no ctx): Leon::decodeAnchorDict, the inverse of the stream leon_dna_finish returns.
 * out_kmers: n_anchors * W words. */
int leon_host_anchor_dict_decode(const uint8_t* payload, uint64_t size, uint64_t n_anchors, uint32_t kmer_size,
                                 uint64_t* out_kmers);

/* -- the streams either side of the DNA stream --------------------------------------------------
 * Header stream, Leon::startHeaderCompression's Dispatcher::iterate(bank, HeaderEncoder(this)) [RECALLED]: headers
 * (text after '>' / '@', no newline) back to back with offsets[n_reads + 1], in file order; first_header: the file's
 * first header, which every block starts from (AbstractHeaderCoder::startBlock; upstream stores it in the metadata).
 * Same batching rules and sink contract as leon_dna_encode_batch; the header stream keeps its own read / block
 * counters on the context and follows leon_dna_set_shard.  Record layout: DESIGN.md section 1.3. */
int leon_header_encode_batch(leon_dna_ctx* ctx, const uint8_t* headers, const uint64_t* offsets, uint64_t n_reads,
                             uint64_t first_read_index, const uint8_t* first_header, uint64_t first_header_len,
                             leon_block_sink sink, void* user);
int leon_header_encode_batch_device(leon_dna_ctx* ctx, const uint8_t* d_headers, const uint64_t* d_offsets, uint64_t n_reads,
                                    uint64_t first_read_index, const uint8_t* first_header, uint64_t first_header_len,
                                    leon_block_sink sink, void* user);
/* Host-only (no GPU, no ctx): HeaderDecoder over read blocks, blocks in parallel on n_threads host threads (0 = all).
 * out: the headers back to back in block order, out_off[total reads + 1]; *out_size = bytes needed (LEON_E_OVERFLOW when
 * out_cap is smaller: call again with that capacity).  Errors: leon_last_error(NULL). */
int leon_host_header_decode_blocks(const uint8_t* payloads, const uint64_t* payload_off, const uint32_t* block_n_reads,
                                   uint64_t n_blocks, const uint8_t* first_header, uint64_t first_header_len, uint8_t* out,
                                   uint64_t out_cap, uint64_t* out_off, uint64_t* out_size, uint32_t n_threads);
/* The same with the arithmetic decoding on the device (one wave per block, every block at once; the header TEXT is rebuilt from
 * the decoded symbols on n_threads host threads): same arguments, same results and errors as leon_host_header_decode_blocks.
 * A block with more symbols than the device buffer gives it (free text in every header) sends the call to the host decoder. */
int leon_header_decode_blocks(leon_dna_ctx* ctx, const uint8_t* payloads, const uint64_t* payload_off, const uint32_t* block_n_reads,
                              uint64_t n_blocks, const uint8_t* first_header, uint64_t first_header_len, uint8_t* out,
                              uint64_t out_cap, uint64_t* out_off, uint64_t* out_size, uint32_t n_threads);
/* leon_header_decode_blocks in its two halves, for callers that decode a file in rounds: the symbols of ALL header blocks in one device
 * call (a block is one serial chain on one wave, ~1.5 s whether the call holds 200 blocks or 2 000: a call per round pays that every
 * round), then the text of any run of blocks on host threads.  The set is the library's until leon_header_symbols_free.
 * leon_header_text_from_symbols: blocks [first_block, first_block + n_blocks) of the set, block_n_reads = THOSE blocks' read counts; outputs
 * and errors as leon_host_header_decode_blocks; LEON_E_STATE when the set's symbols did not fit the device buffer (free text in every
 * header): the caller then decodes the payloads with leon_host_header_decode_blocks. */
typedef struct leon_header_symbols leon_header_symbols;
int leon_header_decode_symbols(leon_dna_ctx* ctx, const uint8_t* payloads, const uint64_t* payload_off, const uint32_t* block_n_reads,
                               uint64_t n_blocks, leon_header_symbols** set);
int leon_header_text_from_symbols(const leon_header_symbols* set, uint64_t first_block, uint64_t n_blocks, const uint32_t* block_n_reads,
                                  const uint8_t* first_header, uint64_t first_header_len, uint8_t* out, uint64_t out_cap, uint64_t* out_off,
                                  uint64_t* out_size, uint32_t n_threads);
void leon_header_symbols_free(leon_header_symbols* set);
/* Quality stream, lossy form (Leon's default, /root/reference/README.md:55): DnaEncoder::storeSolidCoverageInfo + smoothQuals
 * [RECALLED]: a quality becomes '@' where at least two of the read's solid k-mers (in the bloom of ctx) span the position, or
 * where it is above '@'; reads shorter than k are left alone.  quals: one byte per base, same offsets as the bases, rewritten
 * in place; what comes out then goes through leon_host_qual_encode_blocks like the lossless form. */
int leon_qual_smooth_batch(leon_dna_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, uint8_t* quals);
int leon_qual_smooth_batch_device(leon_dna_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n_reads,
                                  uint8_t* d_quals);   /* d_quals[j] belongs to base d_bases[d_offsets[0] + j] */
/* Quality stream, lossless form (`-lossless`): per read block the quality lines, each followed by '\n', through zlib
 * (upstream deflates the block's buffered quality lines, Leon::writeBlockLena [RECALLED]).  Host-only, blocks in parallel
 * on n_threads threads; blocks go to the sink in increasing id starting at first_block_id.  zlib_level: -1 = zlib default. */
int leon_host_qual_encode_blocks(const uint8_t* quals, const uint64_t* offsets, uint64_t n_reads, uint32_t reads_per_block,
                                 int zlib_level, uint32_t n_threads, leon_block_sink sink, void* user, uint64_t first_block_id);
/* The same blocks written on the device: d_quals = the qualities in device memory (d_quals[j] is the byte at offsets[0] + j of the
 * concatenation, as leon_qual_smooth_batch_device leaves them), offsets in HOST memory.  Each block is a zlib stream that inflates
 * to the block's quality lines, each followed by '\n' -- what leon_host_qual_decode_blocks (or upstream's inflate) reads -- but
 * not the bytes zlib's default strategy would write: deflate with matches at distance 1 only (zlib's Z_RLE) and dynamic Huffman
 * codes per 32 KB of text (deflate_kernels.hip).  Errors: leon_last_error(NULL). */
int leon_qual_deflate_blocks_device(int device_id, const uint8_t* d_quals, const uint64_t* offsets, uint64_t n_reads,
                                    uint32_t reads_per_block, leon_block_sink sink, void* user, uint64_t first_block_id);
/* It keeps its device buffers (about 1.5 GB after a large call) for the next call: this returns them. */
void leon_qual_deflate_release(void);
/* inverse: block_n_bytes = quality bytes per block without the newlines; out_off[total reads + 1] */
int leon_host_qual_decode_blocks(const uint8_t* payloads, const uint64_t* payload_off, const uint32_t* block_n_reads,
                                 const uint64_t* block_n_bytes, uint64_t n_blocks, uint8_t* out, uint64_t out_cap,
                                 uint64_t* out_off, uint32_t n_threads);

/* Start a new output file on the same context: forgets the anchor dictionary, the dictionary stream and the
 * read/block counters (a fresh Leon object upstream); keeps the bloom and the device buffers. */
int leon_dna_reset_stream(leon_dna_ctx* ctx);

int leon_dna_get_stats(const leon_dna_ctx* ctx, leon_dna_stats* out);

/* Measurement hook (profiles/scripts/walk_order.py): the NEXT batch's reads are walked in the order of d_keys (one 48-bit key
 * per read of the batch, device memory, the caller's until the call returns) instead of the order of their anchors' addresses.
 * Changes which lanes share bloom sectors, never the bytes: walk events are indexed by read position. */
int leon_dna_debug_walk_order(leon_dna_ctx* ctx, const uint64_t* d_keys);

/* -- traces of the last batch (need LEON_F_KEEP_TRACE); for stage-wise parity tests -- */
int leon_dna_trace_anchors(leon_dna_ctx* ctx, int32_t* anchor_pos, uint32_t* anchor_addr, uint8_t* flags,
                           uint64_t n_reads);
int leon_dna_trace_events(leon_dna_ctx* ctx, uint8_t* events, uint64_t n_bases);
int leon_dna_anchor_kmers(leon_dna_ctx* ctx, uint64_t* kmers, uint64_t n_anchors);

/* -- RangeEncoder::encode over independent symbol streams (one Order0Model set per stream) --
 * syms: 2 bytes per symbol (model id, value); stream i covers symbols [begin[i], begin[i+1]).
 * model ids as in DESIGN.md (8 small models, 8 numeric groups x 9).  Output concatenated, sizes[i]. */
int leon_rc_encode_streams(leon_dna_ctx* ctx, const uint8_t* syms, const uint64_t* begin, uint64_t n_streams,
                           uint8_t* out, uint64_t out_cap, uint64_t* sizes);

#ifdef __cplusplus
}
#endif
#endif

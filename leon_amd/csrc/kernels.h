// kernels.h -- host-callable launchers of the gfx950 kernels (definitions in *.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "leon_device.h"

namespace leon {

// device view of one batch of reads
struct ReadsDev {
    const uint32_t* packed;     // 2 dwords per 32-base slot
    const uint32_t* nmask;      // 1 dword per slot, bit j&31 of dword j>>5 set <=> base j is an N
    const uint64_t* slot_off;   // first slot of each read (n+1 entries)
    const uint64_t* base_off;   // caller's offsets (n+1 entries), base_off[0] may be non-zero
    const uint32_t* len;
    const uint32_t* n_count;    // number of N per read
    uint64_t n;
    uint64_t ev_origin;         // read whose first base is byte 0 of the event buffer (first read of this rank's share)
    uint32_t k, pad;
};

// anchor dictionary: open addressing, linear probing (replaces Leon::_anchorKmers, Hash16<kmer,u32>)
// A slot is ONE aligned record, so a look-up and its `fin < g` test touch one 64-byte sector:
//   one-word keys (k < 32):  16 bytes { key, fin };            tent[] beside it (only unresolved reads touch it)
//   two-word keys (k >= 32): 32 bytes { key lo, key hi, fin, tent }   (the high word doubles as the claim word)
struct DictDev {
    uint64_t* slots;    // key = canonical k-mer or KEY_EMPTY in the (high) word; fin = global index of the read that
                        // inserted the key, IDX_INF while only proposed
    uint64_t* tent;     // smallest global read index currently proposing the key (per resolution round), at
    uint64_t tstride;   // tent[slot * tstride]: its own array for one-word keys, word 3 of the slot for two-word keys
    uint32_t* addr;     // anchor address once assigned (read once per read, after the window's fixpoint)
    uint64_t mask;      // capacity - 1
    unsigned long long* n_keys;
    uint32_t* wbits;    // 2^WBITS_LOG2-bit filter of the keys made final in the current resolution window (k_final_pos)
    uint32_t* fbits;    // filter of ALL final keys (k_lookup_cand; set by k_check): 64-bit words, a key's word chosen by its MINIMIZER
                        // (consecutive k-mers of a read mostly share it), its two bits inside the word by the key's hash
    uint32_t* pbits;    // 2^WBITS_LOG2-bit filter of the keys PROPOSED in the current window (k_check)
    uint32_t fwshift;   // 32 - log2(words of fbits)
    uint32_t mm_m, mm_P, mm_c;   // the minimizer's geometry (minimizer_geometry below)
    int* err;           // set by a kernel that gave up waiting for another wave's half-written two-word key
};
constexpr uint32_t WBITS_LOG2 = 23;       // the two per-window filters: 1 MiB each (a window makes <= 2^20 keys final: <= 12 % full, ~2 % at steady state)
// The word of the final keys' filter a k-mer belongs to: the smallest hash among P of its canonical m-mers -- the P in the MIDDLE of
// the k-mer (offsets c .. c + P - 1 of its k - m + 1), P a power of two (the sliding minimum along a read is then log2(P) shifted
// minima, no remainder), m = 16 or 15 bases so that the middle is symmetric: the same m-mers whichever strand the k-mer is read on.
inline void minimizer_geometry(uint32_t k, uint32_t& m, uint32_t& P, uint32_t& c) {
    if (k <= 16) { m = k; P = 1; c = 0; return; }
    for (m = 16;; m--) {
        const uint32_t w = k - m + 1;
        P = 1;
        while (P * 2 <= w && P < 32) P *= 2;
        if (((w - P) & 1) == 0) { c = (w - P) / 2; return; }         // (m = 15 always gets here: w grows by one, P stays or w == P)
    }
}
constexpr uint32_t FBITS_LOG2_MAX = 31;
constexpr uint32_t FBITS_LOG2_DEFAULT = 28;   // 32 MiB (the sweep that chose it)
// (Rounds 2-3, for the record: one bit per key at a place of the key's own hash, 4 MiB = an XCD's L2 -- 2^22 bits 336 ms for the resolve
// stage at 100 M reads, 2^24 300, 2^25 283, 2^26 287, 2^27 285; non-temporal dictionary probes, a second-level filter, windows of
// 2^19 / 2^21 / 2^22 reads: nothing better than 217 ms, profiles/r3_resolve_sweep.txt.  Addressed by minimizer: 166-184 ms.)

enum : uint8_t { ST_NOANCHOR = 0, ST_HIT = 1, ST_UNRESOLVED = 2, ST_INSERTER = 3, ST_HITNEW = 4 };

struct ResolveDev {
    uint8_t* status;
    uint32_t* hit_pos; uint32_t* hit_slot;
    uint32_t* cand_pos; uint32_t* cand_slot;
    int32_t* anchor_pos; uint32_t* anchor_addr; uint8_t* flags; uint64_t* sort_key;
    uint32_t* ins_flag;          // per read of the current window
};

// ---- bloom ----
void launch_bloom_insert(hipStream_t s, BloomDev B, const uint16_t* rv16, const uint64_t* kmers, uint64_t n);
void launch_bloom_query(hipStream_t s, BloomDev B, const uint16_t* rv16, const uint64_t* kmers, uint64_t n,
                        int mode /*0 contains, 1 contains4 left, 2 contains4 right*/, uint8_t* out);
// ---- pack ----
void launch_rebase_offsets(hipStream_t s, uint64_t* off, uint64_t n, uint64_t base);
void launch_read_slots(hipStream_t s, const uint64_t* base_off, uint64_t n, uint64_t* slots, uint32_t* bad /*set when an offset pair is not a read*/);
void launch_pack(hipStream_t s, const uint8_t* bases, const uint64_t* base_off, const uint64_t* slot_off, uint64_t n,
                 uint32_t* packed, uint32_t* nmask, uint32_t* len, uint32_t* n_count);
// ---- anchor resolution ----
void launch_dict_init(hipStream_t s, DictDev D, uint64_t cap, uint32_t W);
void launch_dict_rehash(hipStream_t s, DictDev from, uint64_t from_cap, DictDev to, uint32_t k);
void launch_lookup_cand(hipStream_t s, ReadsDev R, BloomDev B, const uint16_t* rv16, DictDev D, ResolveDev V,
                        uint64_t w0, uint64_t w1, uint64_t first_global, uint32_t* ulist, uint32_t* ucount,
                        unsigned long long* trace = nullptr /* 12 counters, zeroed by the caller: the traced instantiation (LEON_TRACE_RESOLVE) */,
                        uint64_t* xres = nullptr /* per read of the pass, at [i - xbase]: (position << 8) | status, for the other ranks */,
                        uint64_t xbase = 0, bool dry = false /* xres only: nothing of the pass stays in this context */);
// the reads of [w0, w1) outside [s0, s1), from what the ranks that looked them up found (xres, indexed from w0)
void launch_lookup_apply(hipStream_t s, ReadsDev R, DictDev D, ResolveDev V, uint64_t w0, uint64_t w1, uint64_t s0, uint64_t s1,
                         uint64_t first_global, const uint64_t* xres, uint32_t* ulist, uint32_t* ucount);
void launch_check(hipStream_t s, ReadsDev R, DictDev D, ResolveDev V, uint64_t first_global,
                  const uint32_t* ulist, const uint32_t* ucount, uint32_t max_count, uint32_t* next_list, uint32_t* next_count);
void launch_reset_tent(hipStream_t s, DictDev D, ResolveDev V, const uint32_t* list, const uint32_t* count, uint32_t max_count);
void launch_propose(hipStream_t s, DictDev D, ResolveDev V, uint64_t first_global, const uint32_t* list,
                    const uint32_t* count, uint32_t max_count);
void launch_final_pos(hipStream_t s, ReadsDev R, DictDev D, ResolveDev V, uint64_t w0, uint64_t w1, uint64_t first_global);
void launch_ins_flags(hipStream_t s, ResolveDev V, uint64_t w0, uint64_t w1);
void launch_assign_addr(hipStream_t s, DictDev D, ResolveDev V, uint64_t w0, uint64_t w1, const uint32_t* rank,
                        uint64_t addr_base, uint64_t* anchor_kmers, uint32_t k);
// the exact sequential pass behind the rounds (dna_kernels.hip, "k_chain_*"): what the rounds leave, in read order
constexpr uint32_t CHAIN_LOG2 = 19;      // reads per k_chain_seq: a bit per read in LDS (64 KB)
constexpr uint32_t CHAIN_EL = 96;        // entry rows of a step (64 reads) staged in LDS; longer lists read the rest from global memory
constexpr uint32_t CHAIN_DEPTH = 3;      // steps staged ahead of the consumer wave
void launch_chain_flags(hipStream_t s, ResolveDev V, uint64_t w0, uint64_t w1, uint32_t* flag);
void launch_chain_compact(hipStream_t s, ResolveDev V, uint64_t w0, uint64_t w1, const uint32_t* rank, uint32_t* list);
void launch_chain_repropose(hipStream_t s, DictDev D, ResolveDev V, uint64_t first_global, const uint32_t* reset_list, uint32_t n_reset,
                            const uint32_t* list, uint32_t n);
void launch_chain_prep(hipStream_t s, bool fill, ReadsDev R, DictDev D, ResolveDev V, uint64_t first_global, uint64_t w0, const uint32_t* list,
                       uint32_t n, uint32_t c0, const uint32_t* rank, uint32_t* cnt, uint32_t* own, const uint64_t* gbase, uint32_t* ent,
                       const unsigned long long* om, const unsigned long long* late, unsigned long long* dep, unsigned long long* xdep);
void launch_chain_tables(hipStream_t s, const uint32_t* cnt, const uint32_t* own, uint32_t n, uint64_t* rows /* steps + 1 */,
                         unsigned long long* om /* 64 per step */, unsigned long long* late /* per step */);
int launch_chain_seq(hipStream_t s, uint32_t n, const uint32_t* cnt, const uint32_t* own, const unsigned long long* dep, const unsigned long long* xdep,
                     const uint64_t* gbase, const uint32_t* ent, unsigned long long* ins /* per step of 64 reads: its inserters */, unsigned long long* trace /* nullptr or 8 counters */);
void launch_chain_apply(hipStream_t s, DictDev D, ResolveDev V, uint64_t first_global, const uint32_t* list, uint32_t n, const unsigned long long* ins, uint32_t k);
void launch_anchor_symbols(hipStream_t s, const uint64_t* kmers, uint64_t n_anchors, uint32_t k, uint8_t* syms);
void launch_finalize_reads(hipStream_t s, ReadsDev R, DictDev D, ResolveDev V, uint64_t w0, uint64_t w1);
// ---- walk ----
// what earlier walkers learnt from the bloom, shared in HBM (dna_kernels.hip, "the walk's path cache"): 64-byte buckets of write-once slots
struct WalkCache { uint64_t* slots; uint64_t bucket_mask; uint32_t hop_shift, pad; };          // slots == nullptr: off; a k-mer is a hop point when its 32-bit hash >> hop_shift is 0
void launch_walk_cache_init(hipStream_t s, WalkCache C, uint32_t k);
void launch_walk(hipStream_t s, ReadsDev R, BloomDev B, const uint16_t* rv16, const int32_t* anchor_pos, const uint8_t* flags,
                 const uint32_t* perm, uint64_t n_walk, uint8_t* events, const uint64_t* ev_off = nullptr /* per walked read: its place in `events` */,
                 WalkCache wc = WalkCache{nullptr, 0, 28, 0});
// ---- the walk divided by anchor among the ranks of a job (leon_dna_set_exchange) ----
void launch_lower_bound(hipStream_t s, const uint64_t* sorted, uint64_t n, uint64_t bound, unsigned long long* out);
void launch_slice_weights(hipStream_t s, const uint64_t* sorted_keys, uint64_t n, uint64_t* w /* n + 1 */);
void launch_slice_splits(hipStream_t s, const uint64_t* cum /* exclusive sums of the weights, n + 1 */, uint64_t n, uint32_t world, unsigned long long* split /* world + 1 */);
void launch_slice_reads(hipStream_t s, ReadsDev R, const uint32_t* perm /* the slice */, uint64_t n_slice, uint64_t* len_out /* n_slice + 1 */,
                        uint32_t* slot_of /* n, filled with 0xFF by the caller */);
void launch_ev_words(hipStream_t s, ReadsDev R, const uint32_t* slot_of, const uint64_t* ev_off, const uint8_t* events, uint64_t n, uint32_t rpb,
                     uint64_t n_blocks, uint32_t world, uint64_t* cnt_or_off /* n + 1 */, uint64_t* send /* nullptr = count pass */);
void launch_ev_scatter(hipStream_t s, const uint64_t* words, uint64_t n_words, uint8_t* events, uint64_t n_bytes, int* err);
// ---- symbols ----
void launch_prev_anchored(hipStream_t s, const int32_t* anchor_pos, uint64_t n, uint32_t rpb, uint64_t first_block,
                          uint64_t n_blocks, int64_t* prev);
void launch_symbols(hipStream_t s, ReadsDev R, const int32_t* anchor_pos, const uint32_t* anchor_addr,
                    const uint8_t* flags, const int64_t* prev, const uint8_t* events, uint64_t r0, uint64_t n_local,
                    uint64_t* sym_off /*count or offsets, indexed from r0*/, uint32_t* n_err /*per read, from the count pass*/,
                    uint8_t* syms /*nullptr = count pass*/);
void launch_block_ranges(hipStream_t s, const uint64_t* sym_off, uint64_t n_reads, uint32_t rpb, uint64_t n_blocks,
                         uint64_t* blk_begin /*n_blocks+1*/, uint64_t* out_off /*n_blocks+1*/);
void launch_max_block_syms(hipStream_t s, const uint64_t* sym_off /*offsets, n_reads+1*/, uint64_t n_reads, uint32_t rpb, uint64_t n_blocks,
                           unsigned long long* out_max /*zeroed by the caller*/);
// ---- range coder ----
// small_sizes: alphabet sizes of the small models, a nibble per model id (SMALL_SIZES_DNA / SMALL_SIZES_HEADER)
// max_block_syms: the longest block's symbol count (an upper bound will do): picks the instantiation that can divide exactly
void launch_rc_encode(hipStream_t s, const uint8_t* syms, const uint64_t* blk_begin, uint64_t n_blocks,
                      uint8_t* out, const uint64_t* out_off, uint64_t* out_size, uint32_t* model_scratch, int* err,
                      uint64_t max_block_syms, uint32_t small_sizes = SMALL_SIZES_DNA, bool counts_apart = false);
// counts_apart: the caller vouches that every numeric group's first model (CompressionUtils::encodeNumeric's byte count) only sees symbols
// 0..8 -- true of k_symbols' stream -- so those models can live apart from the 256-symbol slots; a symbol above 8 there sets *err = 3
// the modelers alone, tiles [tile0, tile1) of every block: a 64-bit record per symbol (cumLow | freq << 22 | model << 44) at
// recs + rec_off[block] (the launch's tiles of the block, in stream order); `state` carries a block's models from one launch to the next
size_t rc_records_state_bytes(uint64_t n_blocks);
void launch_rc_records(hipStream_t s, const uint8_t* syms, const uint64_t* blk_begin, uint64_t n_blocks, uint32_t tile0, uint32_t tile1, uint64_t* recs,
                       const uint64_t* rec_off, uint32_t* state, uint32_t* model_scratch, int* err, uint32_t small_sizes);
// ---- header stream (HeaderEncoder): records of every header against the previous one ----
void launch_hdr_symbols(hipStream_t s, const uint8_t* hdr, const uint64_t* off, uint64_t n, uint32_t rpb, const uint8_t* first,
                        uint32_t first_len, uint64_t* sym_off /*count or offsets*/, uint8_t* syms /*nullptr = count pass*/);
size_t rc_model_scratch_bytes(uint64_t n_blocks);
void launch_gather_payload(hipStream_t s, const uint8_t* out, const uint64_t* out_off, const uint64_t* dst_off,
                           const uint64_t* sizes, uint64_t n_blocks, uint8_t* dst);

// ---- quality stream, lossy form: DnaEncoder::smoothQuals over packed reads, quals in place (indexed like the bases) ----
void launch_qual_smooth(hipStream_t s, ReadsDev R, BloomDev B, const uint16_t* rv16, uint8_t* quals);     // file order, every read probes for itself
// the same with the probes shared between the reads of a locus: minimizer per read -> (caller sorts) -> solid flags -> rewrite
void launch_read_minimizer(hipStream_t s, ReadsDev R, uint32_t* key /*its hash, 31 bits; 0xFFFFFFFF for reads shorter than k*/, uint32_t* mpos /*position << 1 | forward*/);
// ... or, for reads whose anchors are known, the anchor in the minimizer's place (key = the anchor's address)
void launch_mpos_from_anchors(hipStream_t s, ReadsDev R, const int32_t* anchor_pos, const uint32_t* anchor_addr, const uint8_t* flags, uint32_t* key, uint32_t* mpos);
void launch_solid_flags(hipStream_t s, ReadsDev R, BloomDev B, const uint16_t* rv16, const uint32_t* perm, const uint32_t* mpos,
                        uint32_t* flags /*a word per 32-base slot like nmask, zeroed*/);
void launch_qual_rewrite(hipStream_t s, ReadsDev R, const uint32_t* flags, uint8_t* quals);
// ---- decoder (DnaDecoder) ----
// what the decoding waves have learnt from the bloom, shared in HBM (decode_kernels.hip): 64-byte buckets of write-once slots
struct PathCache { uint64_t* slots; uint64_t bucket_mask; };          // slots == nullptr: off
size_t path_cache_slot_bytes(uint32_t k);
void launch_path_cache_init(hipStream_t s, PathCache C, uint32_t k);
void launch_path_cache_prewalk(hipStream_t s, BloomDev B, PathCache C, const uint16_t* rv16, const uint64_t* anchors, uint64_t n_anchors, uint32_t max_steps);
void launch_bloom_fingerprint(hipStream_t s, const uint8_t* bits, uint64_t n_bytes, uint64_t* d_sum /* zeroed by the caller */);
size_t decode_scratch_bytes(uint64_t n_blocks);
// header blocks: the stream's symbols as bytes, block b's in syms[sym_begin[b] .. sym_begin[b + 1]) (its share), sym_count[b] of them
void launch_hdr_decode_symbols(hipStream_t s, const uint8_t* payloads, const uint64_t* pay_off, const uint32_t* blk_reads, uint64_t n_blocks,
                               uint8_t* syms, const uint64_t* sym_begin, unsigned long long* sym_count, int* err);
void launch_decode_blocks(hipStream_t s, BloomDev B, PathCache C, const uint16_t* rv16, const uint64_t* anchors, uint64_t n_anchors,
                          const uint8_t* payloads, const uint64_t* pay_off, const uint32_t* blk_reads, const uint64_t* blk_read0,
                          const uint64_t* blk_out0, uint64_t n_blocks, uint8_t* out, uint32_t* out_len, uint32_t* scratch,
                          uint32_t* pool, unsigned long long* pool_cursor, uint64_t pool_words, int* err,
                          unsigned long long* stats /* nullptr, or 16 counters: rounds by kind, time by part (LEON_TRACE_DECODE) */);

}  // namespace leon

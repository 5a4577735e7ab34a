// capi.hip -- the C-ABI of include/leon_dna.h: context, device memory, and the stage pipeline
//   pack -> anchor resolution (windows, fixpoint rounds) -> anchor sort -> walk -> symbols -> range coder -> D2H.
#include "../../include/leon_dna.h"
#include "kernels.h"
#include "host_rc.h"
#include "staging.h"
#include "host_blocks.h"

#include "prim.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace leon;

namespace {

thread_local std::string g_create_error;
}
namespace leon { void set_create_error(const std::string& msg) { g_create_error = msg; } }
namespace {

const bool g_trace_alloc = getenv("LEON_TRACE_ALLOC") != nullptr;   // measurement aid: device allocations of 1 ms or more on stderr

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        auto t0 = std::chrono::steady_clock::now();
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { (void)hipGetLastError(); leon_device_trim(); e = hipMalloc(&p, want); }   // (the k-mer counter's parked buffers: kmer_kernels.hip)
        if (e == hipSuccess) cap = want;
        if (g_trace_alloc) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms >= 1.0) fprintf(stderr, "[leon alloc] %.1f MB in %.1f ms\n", want / 1e6, ms);
        }
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T* as() const { return (T*)p; }
};
// a device buffer that lives for one call: released on every way out of the scope
struct TmpBuf : DevBuf {
    TmpBuf() = default;
    TmpBuf(const TmpBuf&) = delete;
    TmpBuf& operator=(const TmpBuf&) = delete;
    ~TmpBuf() { release(); }
};

uint64_t splitmix_rv(uint32_t idx) {     // built-in simplehash16 table, see DESIGN.md "recalled constants"
    uint64_t s = 0x4C454F4EULL + (uint64_t)(idx + 1) * 0x9E3779B97F4A7C15ULL;
    s = (s ^ (s >> 30)) * 0xBF58476D1CE4E5B9ULL;
    s = (s ^ (s >> 27)) * 0x94D049BB133111EBULL;
    return s ^ (s >> 31);
}
uint64_t hash_seed0() {                  // HashFunctors::generate_hash_seed, seed_tab[0], user_seed 0
    const uint64_t r0 = 0xAAAAAAAA55555555ULL, r3 = 0xB5B5B5B54B4B4B4BULL;
    return r0 * r3;
}

__global__ void k_iota(uint32_t* v, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) v[i] = (uint32_t)i;
}

}  // namespace

// The short waits of a batch -- a counter read back after a kernel of a millisecond, hundreds of them in the anchor resolution, and
// the ones before the first anchors reach the dictionary chain -- POLL the stream instead of calling hipStreamSynchronize: a wait
// that goes to sleep comes back on the host's 10 ms tick here (steps of chain + 30-40 ms, leon_dna_reset_stream taking 17 ms instead
// of 0.4), and the read-backs land in pinned host words so that the copies themselves do not wait inside the runtime.  Long waits
// (the walk, the range coder) keep sleeping: a spinning core would only take clock from the chain's.
// LEON_SPIN_SYNC=0: always sleep in the runtime instead (hosts with few cores, or many contexts whose spinning waits would compete
// with one another and with the chain thread the spinning was meant to help).
inline hipError_t spin_sync(hipStream_t s) {
    static const bool spin = [] { const char* e = getenv("LEON_SPIN_SYNC"); return !(e && e[0] == '0'); }();
    if (!spin) return hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0;; i++) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        for (int k = 0; k < 32; k++) __builtin_ia32_pause();
        if ((i & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
    }
    (void)hipGetLastError();
    return hipStreamSynchronize(s);
}

struct leon_dna_ctx {
    leon_dna_cfg cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    uint64_t* h_rb = nullptr;                                    // pinned host words the short read-backs of a batch land in (spin_sync)
    std::string err;
    // bloom
    uint8_t* d_bloom = nullptr;
    uint64_t bloom_nchar = 0;
    BloomDev B{};
    uint16_t* d_rv16 = nullptr;
    // dictionary
    DictDev D{};
    uint64_t dict_cap = 0;
    uint64_t n_keys = 0;
    unsigned long long* d_nkeys = nullptr;
    uint64_t n_anchors = 0;
    DevBuf anchor_kmers;
    AnchorDictWorker* anchor_worker = nullptr;   // host thread coding the dictionary stream, fed per window
    double anchor_wait_ms = 0;
    std::vector<uint8_t> dict_device_out;        // LEON_F_DICT_ON_DEVICE: the stream coded by k_rc_encode
    // stream state
    uint64_t next_read = 0, next_block = 0;
    uint32_t shard_rank = 0, shard_world = 1;    // leon_dna_set_shard
    uint32_t xch_mode = LEON_XCH_OFF;            // leon_dna_set_exchange: how the walk is divided among the ranks
    leon_exchange_fn xch_fn = nullptr; void* xch_user = nullptr;
    leon_gather_fn gather_fn = nullptr; void* gather_user = nullptr;   // leon_dna_set_gather: the look-ups of a window divided as well
    DevBuf xch_slot, xch_off, xch_evoff, xch_events, xch_send, xch_split, xch_res;
    DevBuf resolve_trace;
    DevBuf round_hist;                           // the counts of a window's fixpoint rounds, read back together
    DevBuf wcache;                               // the walk's path cache (dna_kernels.hip): cleared at every batch
    DevBuf chain_cnt, chain_own, chain_ins, chain_rows, chain_ent, chain_trace, chain_dep, chain_xdep, chain_om, chain_late;   // the sequential pass behind the rounds (k_chain_*)
    std::vector<hipEvent_t> chain_ev;            // pairs around the sequential passes of a batch
    // small launches: the blocks' chains on host cores (host_blocks.h), fed by the device's modelers chunk by chunk
    DevBuf hb_recs[2], hb_recoff, hb_state;
    uint64_t* h_recs = nullptr; size_t h_recs_cap = 0;          // pinned: the records of a launch
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> hb_ev;                              // per chunk: modelled, copied
    std::vector<HostBlockCoder> hb_coders;
    uint64_t* h_anchor[2] = {nullptr, nullptr}; size_t h_anchor_cap = 0;   // pinned: a window's new anchors on their way to the dictionary chain
    bool partial_seen = false, finished = false;
    uint64_t hdr_next_read = 0, hdr_next_block = 0;   // the header stream's own counters (leon_header_encode_batch)
    bool hdr_partial_seen = false;
    DevBuf hdr_first;
    const uint64_t* walk_keys = nullptr;         // leon_dna_debug_walk_order: the next batch's walk order (measurement hook)
    uint32_t fbits_log2 = FBITS_LOG2_DEFAULT;
    bool poisoned = false;                       // a batch failed after it had started to change the stream: LEON_E_STATE until reset_stream
    // batch buffers
    DevBuf in_bases, in_off, slot_off, packed, nmask, rlen, ncount;
    DevBuf status, hit_pos, hit_slot, cand_pos, cand_slot, anchor_pos, anchor_addr, flags, sort_key, ins_flag, rank;
    DevBuf ulist0, ulist1, counters, cub_tmp, sort_key2, perm, perm2, events, prev, sym_off, syms;
    DevBuf blk_begin, out_off, out_size, rc_out, rc_scratch, dst_off, payload, errflag, nerr, wbits, fbits, pbits;
    void* h_payload = nullptr; size_t h_payload_cap = 0;
    uint64_t last_n = 0, last_bases = 0;
    const void* last_d_bases = nullptr; const void* last_d_off = nullptr;      // the last encoded batch: its anchors can serve the smoothing of the same reads
    // decoder: the path cache (decode_kernels.hip) lives as long as the bloom it was learnt from
    DevBuf dc_cache;
    DevBuf dc_out, dc_pay, dc_len, dc_pool, dc_scr;   // a decode call's large buffers, kept from call to call (a fresh 15 GB allocation waits 0.3 s for the driver to wipe it)
    PathCache dc_pc{};
    uint64_t dc_bloom_fp = 0;                    // fingerprint of the bloom bits the cache's entries were derived from
    bool dc_filled = false;
    leon_dna_stats stats{};
    hipEvent_t ev[12]{};
    std::vector<hipEvent_t> pack_ev;             // pairs around the pack launches of a batch
};

namespace {

int fail(leon_dna_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}
#define HIPCHK(c, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail((c), LEON_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));    \
    } while (0)

void dict_free(DictDev& D) {
    if (D.slots) (void)hipFree(D.slots);
    if (D.tent && D.tstride == 1) (void)hipFree(D.tent);     // two-word keys keep tent inside the slots
    if (D.addr) (void)hipFree(D.addr);
    D = DictDev{};
}
int dict_alloc(leon_dna_ctx* c, DictDev& D, uint64_t cap) {
    const uint32_t W = kmer_words(c->cfg.kmer_size);
    D = DictDev{};
    hipError_t e = hipMalloc((void**)&D.slots, cap * 16 * W);
    if (e == hipSuccess && W == 1) { D.tstride = 1; e = hipMalloc((void**)&D.tent, cap * 8); }
    if (e == hipSuccess) e = hipMalloc((void**)&D.addr, cap * 4);
    if (e != hipSuccess) {                                    // nothing of a half-built table is kept
        dict_free(D);
        return fail(c, LEON_E_HIP, std::string("anchor dictionary allocation: ") + hipGetErrorString(e));
    }
    if (W == 2) { D.tent = D.slots + 3; D.tstride = 4; }
    D.mask = cap - 1;
    D.n_keys = c->d_nkeys;
    D.wbits = c->wbits.as<uint32_t>();
    D.fbits = c->fbits.as<uint32_t>();
    D.pbits = c->pbits.as<uint32_t>();
    D.fwshift = 32 - (c->fbits_log2 - 6);
    minimizer_geometry(c->cfg.kmer_size, D.mm_m, D.mm_P, D.mm_c);
    D.err = c->errflag.as<int>() + 2;
    launch_dict_init(c->stream, D, cap, W);
    return LEON_OK;
}
// capacity >= 4 * keys: the look-up kernels are bound by chains of dependent loads, not by the table footprint
int dict_reserve(leon_dna_ctx* c, uint64_t keys) {
    uint64_t need = 1024;
    while (need < 4 * keys) need <<= 1;                   // load factor <= 1/4: a miss costs 1.2 dependent probes, not 2.5
    if (need <= c->dict_cap) return LEON_OK;
    if (need > (1ull << 32)) return fail(c, LEON_E_OVERFLOW, "anchor dictionary would exceed 2^32 slots");
    DictDev nd{};
    HIPCHK(c, hipMemsetAsync(c->d_nkeys, 0, 8, c->stream));
    int rc = dict_alloc(c, nd, need);
    if (rc) return rc;
    if (c->dict_cap) {
        launch_dict_rehash(c->stream, c->D, c->dict_cap, nd, c->cfg.kmer_size);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        dict_free(c->D);
    }
    c->D = nd;
    c->dict_cap = need;
    return LEON_OK;
}

ReadsDev reads_view(leon_dna_ctx* c, const uint64_t* d_off, uint64_t n) {
    ReadsDev R{};
    R.packed = c->packed.as<uint32_t>(); R.nmask = c->nmask.as<uint32_t>(); R.slot_off = c->slot_off.as<uint64_t>();
    R.base_off = d_off; R.len = c->rlen.as<uint32_t>(); R.n_count = c->ncount.as<uint32_t>();
    R.n = n; R.k = c->cfg.kmer_size;
    return R;
}

// leon_dna_encode_batch's upload of the caller's bases while the device already packs and resolves the groups that have
// arrived.  The caller's memory is pageable: one hipMemcpy from it runs at 11-12 GB/s here (the runtime stages it through
// ONE thread), a fifth of what PCIe carries.  So the staging is done here, by a few threads: each takes the next 16 MiB
// piece of the byte range, copies it into one of its two pinned buffers and sends that to the device on its own stream.
struct Upload {
    static constexpr uint64_t kPiece = kStagePiece;             // (staging.h; 3 threads reach PCIe's 52-56 GB/s, more only take CPU time from the dictionary
                                                                //  chain: 6 threads 1 079 ms per step, 3 threads 910)
    std::atomic<uint64_t> bytes_done{0};                       // bases [0, bytes_done) of the batch are in HBM
    std::atomic<int> failed{0};                                // 1: a copy failed
    std::atomic<int> cancel{0};                                // set by the caller when the batch has failed: stop copying
    std::atomic<uint64_t> next_piece{0};
    std::vector<std::atomic<uint8_t>> piece_done;
    std::mutex mu;
    uint64_t prefix = 0, n_bytes = 0;
    std::vector<std::thread> th;
    explicit Upload(uint64_t bytes) : piece_done((bytes + kPiece - 1) / kPiece), n_bytes(bytes) { for (auto& f : piece_done) f.store(0); }
    void publish(uint64_t piece) {
        piece_done[piece].store(1, std::memory_order_release);
        std::lock_guard<std::mutex> g(mu);
        while (prefix < piece_done.size() && piece_done[prefix].load(std::memory_order_acquire)) prefix++;
        bytes_done.store(std::min(n_bytes, prefix * kPiece), std::memory_order_release);
    }
    ~Upload() { for (auto& t : th) if (t.joinable()) t.join(); }
};
constexpr uint32_t kRbHostChainsErr = 500;   // slot of the pinned read-back words (h_rb, 512 of them) the host chains' first chunk brings the modelers' error flag to
int ensure_cub(leon_dna_ctx* c, size_t bytes) { HIPCHK(c, c->cub_tmp.ensure(bytes)); return LEON_OK; }

}  // namespace


namespace leon { uint32_t usable_cpus(); }

namespace {
// Launches of at most this many blocks have their chains coded on host cores (0 = never): LEON_RC_HOST_BLOCKS, default 400.
// With 2 000 blocks (8 per CU) the device's chains all run at once and win: 126 ms for 2.3 G symbols is 18 G symbols/s, 16 host cores at ~2.3 ns do 7.
uint64_t rc_host_blocks() {
    if (const char* e = getenv("LEON_RC_HOST_BLOCKS")) return (uint64_t)std::max<long long>(0, atoll(e));
    return 400;
}
uint32_t rc_host_threads(uint64_t n_blocks, uint32_t world) {
    // (all but one of the process's CPUs: the dictionary chain and its helpers are busy for the first part of a small file's step only --
    // configuration #2, 200 blocks: 88.9 ms with 12 threads and 4 chunks, 73.4 with 15 and 8, 63.9 with 15 and 16; profiles/r4_hostchains_sweep_config2.txt.
    // A rank of a job of `world` (leon_dna_set_shard: the GPUs of ONE node) shares the host, and its CPU quota, with the others: an equal share of
    // what rank 0's dictionary chain and its helpers leave.  Eight ranks taking 32 threads each would run the cgroup out of its quota, and the
    // throttling that follows stops the chain -- the one thing the job's step waits for -- along with everything else.)
    const uint32_t cpus = usable_cpus();
    uint32_t n_thr = world > 1 ? std::max<uint32_t>(2, (cpus > 6 ? cpus - 6 : 0) / world) : (cpus > 2 ? cpus - 1 : 2);
    n_thr = std::max<uint32_t>(2, std::min<uint32_t>(32, n_thr));
    if (const char* e = getenv("LEON_RC_HOST_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 256) n_thr = (uint32_t)v; }
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_thr, n_blocks));
}
// Who codes a launch's chains.  The device's coder waves take about as long as the LONGEST block's chain whatever the number of
// blocks (~100 ns per symbol of it, up to 8 blocks per CU); the host's way takes the slowest of three things that run side by side --
// the modelers' pass over the longest block (four waves per block since round 5), the records' 8 bytes per symbol over PCIe, the host
// threads' chains (two side by side per thread) -- a chunk behind one another.  10 M reads of 150 bp in 200 blocks: 116 against ~42 ms,
// the host (measured 134 and 45); 20 M reads of 250 bp in 400 blocks of 1.9 M symbols: 190 against ~140 (measured 174 and 142): left to the device, see below.
bool rc_on_host(uint64_t n_blocks, uint64_t n_syms, uint64_t max_block_syms, uint32_t world) {
    if (!n_blocks || n_blocks > rc_host_blocks() || max_block_syms + 1024 >= (1ull << HB_COUNT_BITS) || n_syms * 8 > (12ull << 30)) return false;
    if (getenv("LEON_RC_HOST_BLOCKS")) return true;              // (asked for by name: the tests, measurements)
    const double device_ns = 100.0 * (double)max_block_syms * (double)((n_blocks + 2047) / 2048);
    // (round 5: four modeler waves per block -- ~12 ns per symbol of the longest block --, the records' 8 bytes per symbol over PCIe at ~52 GB/s,
    // the chains two side by side per thread at ~2.6 ns per symbol; whichever is slowest, a sixteenth later for the first chunk's way in)
    const double thr = (double)rc_host_threads(n_blocks, world);
    const double host_ns = 1.0625 * std::max(12.0 * (double)max_block_syms, std::max(0.16 * (double)n_syms, 2.6 * (double)n_syms / thr));
    // (a near tie goes to the device: the host's cores have other work -- the dictionary chain, which is what a step waits for: at the k = 63 / 250 bp
    // shape, 400 blocks of 1.9 M symbols, the host's way takes 142 ms against the device's 174 and the step 554 ms against 542)
    return host_ns < 0.7 * device_ns;
}

// The range coder stage of a small launch: the device's modeler waves turn the symbols of every block into records, chunk of tiles
// by chunk; a chunk crosses PCIe while the next is modelled, and host threads -- each owning every n-th block -- code it while the
// one after that crosses.  On return c->hb_coders[b] holds block b's payload.  Bytes: exactly k_rc_encode's (the tests run both).
int rc_blocks_on_host(leon_dna_ctx* c, const uint8_t* d_syms, const uint64_t* d_blk_begin, uint64_t nbl, uint64_t n_syms, uint32_t small_sizes,
                      uint32_t n_small) {
    hipStream_t s = c->stream;
    if (!c->copy_stream) {
        // A stream of HIGH priority: the runtime maps streams onto a few hardware queues, and in a process that also holds a
        // communicator's streams (RCCL: every rank of an N-rank job) this one landed on the queue of `s` -- every chunk's copy then
        // waited behind all sixteen modeler launches and the chains started when the modelers had finished (a seat of 8: 110 ms for
        // the stage instead of ~75).  Streams of another priority get queues of their own.
        int lo = 0, hi = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(c, hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, hi));
    }
    std::vector<uint64_t> bb(nbl + 1);
    HIPCHK(c, hipMemcpyAsync(bb.data(), d_blk_begin, (nbl + 1) * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    uint64_t max_tiles = 0;
    for (uint64_t b = 0; b < nbl; b++) max_tiles = std::max<uint64_t>(max_tiles, (bb[b + 1] - bb[b] + 63) / 64);
    uint32_t n_chunks = 16;                                      // (a chunk = a pipeline stage: the first must be modelled and cross before a chain starts, the last is coded after everything)
    if (const char* e = getenv("LEON_RC_HOST_CHUNKS")) { const int v = atoi(e); if (v >= 1 && v <= 64) n_chunks = (uint32_t)v; }
    n_chunks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_chunks, max_tiles));
    const uint64_t Tc = (max_tiles + n_chunks - 1) / std::max<uint32_t>(n_chunks, 1);
    // where block b's records of chunk ch lie inside the chunk, and the chunks inside the pinned buffer
    std::vector<uint64_t> rec_off((size_t)n_chunks * nbl), rec_cnt((size_t)n_chunks * nbl), chunk_base(n_chunks + 1, 0);
    uint64_t chunk_max = 0;
    for (uint32_t ch = 0; ch < n_chunks; ch++) {
        uint64_t at = 0;
        for (uint64_t b = 0; b < nbl; b++) {
            const uint64_t len = bb[b + 1] - bb[b], a0 = std::min<uint64_t>(len, ch * Tc * 64), a1 = std::min<uint64_t>(len, (ch + 1) * Tc * 64);
            rec_off[(size_t)ch * nbl + b] = at; rec_cnt[(size_t)ch * nbl + b] = a1 - a0;
            at += a1 - a0;
        }
        chunk_base[ch + 1] = chunk_base[ch] + at;
        chunk_max = std::max(chunk_max, at);
    }
    if (chunk_base[n_chunks] != n_syms) return fail(c, LEON_E_STATE, "block symbol ranges do not add up (internal error)");
    // (memory this way needs and the device's coder does not: when it is not to be had, the caller takes the device's coder -- return 1)
    auto no_room = [&](hipError_t e) { if (e == hipSuccess) return false; (void)hipGetLastError(); return true; };
    for (auto& b : c->hb_recs) if (no_room(b.ensure(chunk_max * 8 + 64))) return 1;
    if (no_room(c->hb_recoff.ensure((size_t)n_chunks * nbl * 8)) || no_room(c->hb_state.ensure(rc_records_state_bytes(nbl)))) return 1;
    HIPCHK(c, c->rc_scratch.ensure(rc_model_scratch_bytes(nbl)));
    if (n_syms * 8 + 64 > c->h_recs_cap) {
        if (c->h_recs) HIPCHK(c, hipHostFree(c->h_recs));
        c->h_recs = nullptr; c->h_recs_cap = 0;
        const size_t want = n_syms * 8 + n_syms + 4096;
        if (no_room(hipHostMalloc((void**)&c->h_recs, want, hipHostMallocDefault))) { c->h_recs = nullptr; return 1; }
        c->h_recs_cap = want;
    }
    while (c->hb_ev.size() < 2 * (size_t)n_chunks) { hipEvent_t e; HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->hb_ev.push_back(e); }
    HIPCHK(c, hipMemcpyAsync(c->hb_recoff.p, rec_off.data(), rec_off.size() * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemsetAsync(c->errflag.p, 0, 4, s));
    if (c->hb_coders.size() < nbl) c->hb_coders.resize(nbl);
    // the host threads
    const uint32_t n_thr = rc_host_threads(nbl, c->shard_world);
    static const bool trace = getenv("LEON_TRACE_RC_HOST") != nullptr;     // measurement aid: when each chunk was modelled / had crossed / was coded, on stderr
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_now = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    std::vector<double> t_coded(n_chunks, 0.0), t_copied(n_chunks, 0.0);
    std::mutex trace_mu;
    (void)hb_recip_table();                                      // (built before the threads ask for it)
    std::atomic<uint32_t> chunks_ready{0};
    std::atomic<int> abort_flag{0};
    std::vector<std::thread> workers;
    const uint64_t* const h_recs = c->h_recs;
    HostBlockCoder* const coders = c->hb_coders.data();
    for (uint32_t w = 0; w < n_thr; w++)
        workers.emplace_back([&, w] {
            try {
            for (uint64_t b = w; b < nbl; b += n_thr) coders[b].start(small_sizes, n_small);
            for (uint32_t ch = 0; ch < n_chunks; ch++) {
                for (uint32_t spin = 0; chunks_ready.load(std::memory_order_acquire) <= ch; spin++) {
                    if (abort_flag.load()) return;
                    if (spin < 2000) __builtin_ia32_pause(); else std::this_thread::yield();
                }
                // two of the thread's blocks side by side (HostBlockCoder::code2: two dependent chains fill the core one leaves half idle)
                static const bool pairs = [] { const char* e = getenv("LEON_RC_HOST_PAIRS"); return !(e && e[0] == '0'); }();
                for (uint64_t b = w; b < nbl; b += 2 * (uint64_t)n_thr) {
                    const uint64_t b2 = b + n_thr;
                    const uint64_t* ra = h_recs + chunk_base[ch] + rec_off[(size_t)ch * nbl + b];
                    if (b2 < nbl && pairs) HostBlockCoder::code2(coders[b], ra, rec_cnt[(size_t)ch * nbl + b], coders[b2], h_recs + chunk_base[ch] + rec_off[(size_t)ch * nbl + b2], rec_cnt[(size_t)ch * nbl + b2]);
                    else {
                        coders[b].code(ra, rec_cnt[(size_t)ch * nbl + b]);
                        if (b2 < nbl) coders[b2].code(h_recs + chunk_base[ch] + rec_off[(size_t)ch * nbl + b2], rec_cnt[(size_t)ch * nbl + b2]);
                    }
                }
                if (trace) { const double t = ms_now(); std::lock_guard<std::mutex> g(trace_mu); t_coded[ch] = std::max(t_coded[ch], t); }
            }
            for (uint64_t b = w; b < nbl; b += n_thr) coders[b].flush();
            } catch (...) { abort_flag.store(2); }               // (an output buffer that could not grow: the call fails, nothing is thrown across threads)
        });
    struct Join { std::vector<std::thread>& t; std::atomic<int>& a; bool ok = false; ~Join() { if (!ok) a.store(1); for (auto& x : t) if (x.joinable()) x.join(); } } join{workers, abort_flag};
    for (uint32_t ch = 0; ch < n_chunks; ch++) {
        if (ch >= 2) HIPCHK(c, hipStreamWaitEvent(s, c->hb_ev[2 * (ch - 2) + 1], 0));            // the buffer's previous chunk has left it
        launch_rc_records(s, d_syms, d_blk_begin, nbl, (uint32_t)(ch * Tc), (uint32_t)((ch + 1) * Tc), c->hb_recs[ch & 1].as<uint64_t>(),
                          c->hb_recoff.as<uint64_t>() + (size_t)ch * nbl, c->hb_state.as<uint32_t>(), c->rc_scratch.as<uint32_t>(), c->errflag.as<int>(), small_sizes);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(c->hb_ev[2 * ch], s));
        HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->hb_ev[2 * ch], 0));
        const uint64_t bytes = (chunk_base[ch + 1] - chunk_base[ch]) * 8;
        if (bytes) HIPCHK(c, hipMemcpyAsync(c->h_recs + chunk_base[ch], c->hb_recs[ch & 1].p, bytes, hipMemcpyDeviceToHost, c->copy_stream));
        // (the modelers' one refusal travels with the first chunk, on this stream: a blocking hipMemcpy of the flag waited for EVERY launch
        // queued on `s` in a process that holds a communicator -- each rank of an N-rank job --, and the chains began when the modelers had finished)
        if (ch == 0) HIPCHK(c, hipMemcpyAsync(c->h_rb + kRbHostChainsErr, c->errflag.p, 4, hipMemcpyDeviceToHost, c->copy_stream));
        HIPCHK(c, hipEventRecord(c->hb_ev[2 * ch + 1], c->copy_stream));
    }
    const double t_enqueued = ms_now();
    for (uint32_t ch = 0; ch < n_chunks; ch++) {
        HIPCHK(c, hipEventSynchronize(c->hb_ev[2 * ch + 1]));
        if (ch == 0 && (int)(uint32_t)c->h_rb[kRbHostChainsErr]) return fail(c, LEON_E_OVERFLOW, "a block has more symbols than the host chains' records can count");
        chunks_ready.store(ch + 1, std::memory_order_release);
        t_copied[ch] = ms_now();
    }
    for (auto& t : workers) t.join();
    join.ok = true;
    if (abort_flag.load() == 2) return fail(c, LEON_E_OVERFLOW, "a host chain ran out of memory for its output");
    if (trace) {
        fprintf(stderr, "[leon rc host] %llu blocks, %llu symbols, %u chunks, %u threads, launches enqueued at %.1f ms:", (unsigned long long)nbl, (unsigned long long)n_syms, n_chunks, n_thr, t_enqueued);
        for (uint32_t ch = 0; ch < n_chunks; ch++) fprintf(stderr, " chunk %u in host memory at %.1f ms, coded at %.1f;", ch, t_copied[ch], t_coded[ch]);
        fprintf(stderr, " done at %.1f ms\n", ms_now());
    }
    return LEON_OK;
}
}  // namespace

extern "C" {

int leon_dna_abi_version(void) { return LEON_DNA_ABI_VERSION; }

const char* leon_last_error(const leon_dna_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int leon_dna_ctx_create(const leon_dna_cfg* cfg, leon_dna_ctx** out) {
    if (!cfg || !out) return fail(nullptr, LEON_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(leon_dna_cfg)) return fail(nullptr, LEON_E_INVALID, "leon_dna_cfg.struct_size mismatch");
    if (cfg->kmer_size < 3 || cfg->kmer_size > 63)
        return fail(nullptr, LEON_E_INVALID, "kmer_size must be in 3..63 (one 64-bit word below 32, two words from 32 to 63)");
    if (cfg->reads_per_block == 0) return fail(nullptr, LEON_E_INVALID, "reads_per_block must be > 0");
    if (cfg->bloom_n_hash < 1 || cfg->bloom_n_hash > 10) return fail(nullptr, LEON_E_INVALID, "bloom_n_hash must be in 1..10");
    if (cfg->bloom_block_nbits < 4 || cfg->bloom_block_nbits > 16)
        return fail(nullptr, LEON_E_INVALID, "bloom_block_nbits must be in 4..16");
    {   // BloomNeighborCoherent's modulus tai' - 2 * blk must stay positive (bloom_tai = 0 makes tai a power of two, tai' = tai - 1
        // < 2 * blk), and an absurd size is refused here rather than by hipMalloc
        const uint64_t blk0 = 1ull << cfg->bloom_block_nbits;
        uint64_t tai0 = cfg->bloom_tai + 2 * blk0;
        if ((tai0 & (tai0 - 1)) == 0) tai0--;
        if (tai0 <= 2 * blk0 || cfg->bloom_tai > (1ull << 46)) return fail(nullptr, LEON_E_INVALID, "bloom_tai out of range (1 .. 2^46 bits)");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, LEON_E_NO_DEVICE, "no HIP device: the DNA encode path has no CPU fallback");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, LEON_E_INVALID, "device_id out of range");
    leon_dna_ctx* c = new leon_dna_ctx();
    c->cfg = *cfg;
    c->cfg.random_values = nullptr;
    if (c->cfg.resolve_window == 0) c->cfg.resolve_window = 1ull << 21;     // (2^20 until the look-ups got cheaper: 174 -> 168 ms per 100 M reads, profiles/r4_minimizer_filter.txt)
    c->device = cfg->device_id;
#define CREATE_CHK(call)                                                                                         \
    do {                                                                                                         \
        hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            int r_ = fail(nullptr, LEON_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));               \
            leon_dna_ctx_destroy(c);                                                                             \
            return r_;                                                                                           \
        }                                                                                                        \
    } while (0)
    CREATE_CHK(hipSetDevice(c->device));
    CREATE_CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CREATE_CHK(hipHostMalloc((void**)&c->h_rb, 4096, hipHostMallocDefault));
    for (auto& e : c->ev) CREATE_CHK(hipEventCreate(&e));
    // BloomCacheCoherent / BloomContainer geometry
    uint64_t blk = 1ull << cfg->bloom_block_nbits;
    uint64_t tai = cfg->bloom_tai + 2 * blk;
    c->bloom_nchar = 1 + tai / 8;
    if ((tai & (tai - 1)) == 0) tai--;
    uint64_t reduced = tai - 2 * blk;
    CREATE_CHK(hipMalloc((void**)&c->d_bloom, c->bloom_nchar + 16));
    CREATE_CHK(hipMemsetAsync(c->d_bloom, 0, c->bloom_nchar + 16, c->stream));
    uint16_t rv16[256];
    for (uint32_t i = 0; i < 256; i++) rv16[i] = (uint16_t)((cfg->random_values ? cfg->random_values[i] : splitmix_rv(i)) & 0xFFFF);
    CREATE_CHK(hipMalloc((void**)&c->d_rv16, sizeof(rv16)));
    CREATE_CHK(hipMemcpy(c->d_rv16, rv16, sizeof(rv16), hipMemcpyHostToDevice));
    uint32_t k = cfg->kmer_size;
    c->B.bits = c->d_bloom; c->B.reduced_tai = reduced; c->B.mod_magic = ~0ull / reduced; c->B.seed0 = hash_seed0();
    c->B.k = k; c->B.n_hash = cfg->bloom_n_hash; c->B.block_mask = (uint32_t)(blk - 1);
    CREATE_CHK(hipMalloc((void**)&c->d_nkeys, 8));
    CREATE_CHK(hipMemset(c->d_nkeys, 0, 8));
    CREATE_CHK(c->counters.ensure(64));
    CREATE_CHK(c->errflag.ensure(16));
    CREATE_CHK(hipMemsetAsync(c->errflag.p, 0, 16, c->stream));
    CREATE_CHK(c->wbits.ensure((1ull << WBITS_LOG2) / 8));
    CREATE_CHK(c->pbits.ensure((1ull << WBITS_LOG2) / 8));
    if (const char* e = getenv("LEON_FBITS_LOG2")) {          // measurement override: size of the final-key filter
        const int v = atoi(e);
        if (v >= 12 && v <= (int)FBITS_LOG2_MAX) c->fbits_log2 = (uint32_t)v;
    }
    CREATE_CHK(c->fbits.ensure((1ull << c->fbits_log2) / 8));
    CREATE_CHK(hipMemsetAsync(c->fbits.p, 0, (1ull << c->fbits_log2) / 8, c->stream));
    CREATE_CHK(hipStreamSynchronize(c->stream));
#undef CREATE_CHK
    c->anchor_worker = new AnchorDictWorker(cfg->kmer_size);
    *out = c;
    return LEON_OK;
}

void leon_dna_ctx_destroy(leon_dna_ctx* c) {
    if (!c) return;
    delete c->anchor_worker;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    dict_free(c->D);
    DevBuf* bufs[] = { &c->anchor_kmers, &c->in_bases, &c->in_off, &c->slot_off, &c->packed, &c->nmask, &c->rlen, &c->ncount,
                       &c->status, &c->hit_pos, &c->hit_slot, &c->cand_pos, &c->cand_slot, &c->anchor_pos, &c->anchor_addr,
                       &c->flags, &c->sort_key, &c->ins_flag, &c->rank, &c->ulist0, &c->ulist1, &c->counters, &c->cub_tmp,
                       &c->sort_key2, &c->perm, &c->perm2, &c->events, &c->prev, &c->sym_off, &c->syms, &c->blk_begin,
                       &c->out_off, &c->out_size, &c->rc_out, &c->rc_scratch, &c->dst_off, &c->payload, &c->errflag, &c->nerr, &c->wbits, &c->fbits, &c->pbits, &c->hdr_first, &c->dc_cache, &c->dc_out, &c->dc_pay, &c->dc_len, &c->dc_pool, &c->dc_scr,
                       &c->xch_slot, &c->xch_off, &c->xch_evoff, &c->xch_events, &c->xch_send, &c->xch_split, &c->xch_res, &c->round_hist, &c->resolve_trace, &c->hb_recs[0], &c->hb_recs[1], &c->hb_recoff, &c->hb_state };
    for (DevBuf* b : bufs) b->release();
    if (c->d_bloom) (void)hipFree(c->d_bloom);
    if (c->d_rv16) (void)hipFree(c->d_rv16);
    if (c->d_nkeys) (void)hipFree(c->d_nkeys);
    if (c->h_payload) (void)hipHostFree(c->h_payload);
    if (c->h_rb) (void)hipHostFree(c->h_rb);
    for (auto& b : c->h_anchor) if (b) (void)hipHostFree(b);
    if (c->h_recs) (void)hipHostFree(c->h_recs);
    for (auto& e : c->hb_ev) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->pack_ev) (void)hipEventDestroy(e);
    for (auto& e : c->chain_ev) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// ------------------------------------------------------------------------------------------------ bloom
int leon_dna_bloom_nbytes(const leon_dna_ctx* c, uint64_t* n) {
    if (!c || !n) return LEON_E_INVALID;
    *n = c->bloom_nchar;
    return LEON_OK;
}
int leon_dna_bloom_upload(leon_dna_ctx* c, const uint8_t* bits, uint64_t n) {
    if (!c || !bits) return LEON_E_INVALID;
    if (n != c->bloom_nchar) return fail(c, LEON_E_INVALID, "bloom_upload: size differs from leon_dna_bloom_nbytes");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, staged_h2d(c->device, c->d_bloom, bits, n));
    return LEON_OK;
}
int leon_dna_bloom_download(leon_dna_ctx* c, uint8_t* bits, uint64_t n) {
    if (!c || !bits) return LEON_E_INVALID;
    if (n != c->bloom_nchar) return fail(c, LEON_E_INVALID, "bloom_download: size differs from leon_dna_bloom_nbytes");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, staged_d2h(c->device, bits, c->d_bloom, n));
    return LEON_OK;
}
int leon_dna_bloom_clear(leon_dna_ctx* c) {
    if (!c) return LEON_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->d_bloom, 0, c->bloom_nchar + 16, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LEON_OK;
}
int leon_dna_bloom_insert_device(leon_dna_ctx* c, const uint64_t* d_kmers, uint64_t n) {
    if (!c || (!d_kmers && n)) return LEON_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    launch_bloom_insert(c->stream, c->B, c->d_rv16, d_kmers, n);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LEON_OK;
}
int leon_dna_bloom_insert(leon_dna_ctx* c, const uint64_t* kmers, uint64_t n) {
    if (!c || (!kmers && n)) return LEON_E_INVALID;
    if (!n) return LEON_OK;
    HIPCHK(c, hipSetDevice(c->device));
    TmpBuf tmp;
    const uint64_t bytes = n * 8 * kmer_words(c->cfg.kmer_size);
    HIPCHK(c, tmp.ensure(bytes));
    HIPCHK(c, hipMemcpy(tmp.p, kmers, bytes, hipMemcpyHostToDevice));
    return leon_dna_bloom_insert_device(c, tmp.as<uint64_t>(), n);
}
int leon_dna_bloom_device_ptr(leon_dna_ctx* c, void** p, uint64_t* n) {
    if (!c || !p || !n) return LEON_E_INVALID;
    *p = c->d_bloom; *n = c->bloom_nchar;
    return LEON_OK;
}
int leon_dna_bloom_upload_device(leon_dna_ctx* c, const uint8_t* d_bits, uint64_t n) {
    if (!c || !d_bits) return LEON_E_INVALID;
    if (n != c->bloom_nchar) return fail(c, LEON_E_INVALID, "bloom_upload_device: size differs from leon_dna_bloom_nbytes");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->d_bloom, d_bits, n, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LEON_OK;
}
int leon_dna_bloom_download_device(leon_dna_ctx* c, uint8_t* d_bits, uint64_t n) {
    if (!c || !d_bits) return LEON_E_INVALID;
    if (n != c->bloom_nchar) return fail(c, LEON_E_INVALID, "bloom_download_device: size differs from leon_dna_bloom_nbytes");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(d_bits, c->d_bloom, n, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LEON_OK;
}
static int bloom_query(leon_dna_ctx* c, const uint64_t* kmers, uint64_t n, int mode, uint8_t* out) {
    if (!c || ((!kmers || !out) && n)) return LEON_E_INVALID;
    if (!n) return LEON_OK;
    HIPCHK(c, hipSetDevice(c->device));
    TmpBuf dk, dout;
    const uint64_t bytes = n * 8 * kmer_words(c->cfg.kmer_size);
    HIPCHK(c, dk.ensure(bytes));
    HIPCHK(c, dout.ensure(n));
    HIPCHK(c, hipMemcpy(dk.p, kmers, bytes, hipMemcpyHostToDevice));
    launch_bloom_query(c->stream, c->B, c->d_rv16, dk.as<uint64_t>(), n, mode, dout.as<uint8_t>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, dout.p, n, hipMemcpyDeviceToHost));
    return LEON_OK;
}
int leon_dna_bloom_contains4(leon_dna_ctx* c, const uint64_t* kmers, uint64_t n, int right, uint8_t* out) {
    return bloom_query(c, kmers, n, right ? 2 : 1, out);
}
int leon_dna_bloom_contains(leon_dna_ctx* c, const uint64_t* kmers, uint64_t n, uint8_t* out) {
    return bloom_query(c, kmers, n, 0, out);
}

// ------------------------------------------------------------------------------------------------ encode
static int encode_batch_impl(leon_dna_ctx* c, const uint8_t* d_bases, const uint64_t* d_off, uint64_t n,
                             uint64_t first_read_index, leon_block_sink sink, void* user, Upload* up, const uint64_t* up_off);

// A batch that fails before it has touched the stream (bad arguments, bad offsets, call order) leaves the context as it was.
// One that fails later -- a HIP error, an internal bound, the sink -- leaves the dictionary, the dictionary-stream thread
// and the caller's block sequence part-way through the batch: the context is poisoned and every call on the stream
// returns LEON_E_STATE until leon_dna_reset_stream starts a new one.
static int encode_batch_guarded(leon_dna_ctx* c, const uint8_t* d_bases, const uint64_t* d_off, uint64_t n,
                                uint64_t first_read_index, leon_block_sink sink, void* user, Upload* up, const uint64_t* up_off = nullptr) {
    if (!c) return LEON_E_INVALID;
    if (c->poisoned) return fail(c, LEON_E_STATE, "an earlier batch failed part-way: the stream is unusable until leon_dna_reset_stream");
    if (c->dc_pc.slots) {                                     // a context that goes back to encoding gives the decoder's table (up to 40 % of the HBM) back first
        (void)hipStreamSynchronize(c->stream);
        c->dc_cache.release(); c->dc_pc = PathCache{}; c->dc_filled = false;
    }
    for (DevBuf* b : { &c->dc_out, &c->dc_pay, &c->dc_len, &c->dc_pool, &c->dc_scr }) b->release();
    const int rc = encode_batch_impl(c, d_bases, d_off, n, first_read_index, sink, user, up, up_off);
    if (rc != LEON_OK && c->poisoned) c->err += " (stream poisoned: leon_dna_reset_stream to go on)";
    return rc;
}

int leon_dna_encode_batch_device(leon_dna_ctx* c, const uint8_t* d_bases, const uint64_t* d_off, uint64_t n,
                                 uint64_t first_read_index, leon_block_sink sink, void* user) {
    return encode_batch_guarded(c, d_bases, d_off, n, first_read_index, sink, user, nullptr);
}

// reads [0, group_end(a)) are packed (and, through the host entry point, uploaded) together: the first resolution window
// alone, so that the device and the dictionary chain start at once, then groups of 8 windows
// (the first window itself is short -- first_window() reads -- so that the first anchors reach the host chain, the longest
// single piece of a step, a few milliseconds after the call starts; the result does not depend on where windows end)
static uint64_t first_window(uint64_t window) { return std::min<uint64_t>(window, 1ull << 17); }
static uint64_t group_end(uint64_t a, uint64_t n, uint64_t window, bool /*streamed*/) {
    if (a == 0) return std::min(n, first_window(window));
    // (resident input used to pack everything that was left in one go: 9.6 ms at 100 M reads between the first window and the
    // second, during which the dictionary chain ran out of the first window's anchors and idled)
    return std::min(n, a + 8 * std::min<uint64_t>(window, 1ull << 20));
}

static int encode_batch_impl(leon_dna_ctx* c, const uint8_t* d_bases, const uint64_t* d_off, uint64_t n,
                             uint64_t first_read_index, leon_block_sink sink, void* user, Upload* up, const uint64_t* up_off) {
    if (!c) return LEON_E_INVALID;
    if (c->finished) return fail(c, LEON_E_STATE, "encode_batch after finish");
    if (first_read_index != c->next_read) return fail(c, LEON_E_STATE, "first_read_index does not continue the stream");
    if (c->partial_seen && n) return fail(c, LEON_E_STATE, "a batch with a partial block must be the last one");
    if (n == 0) return LEON_OK;
    if (!d_bases || !d_off || !sink) return fail(c, LEON_E_INVALID, "null argument");
    if (n > 0xFFFFFFF0ull) return fail(c, LEON_E_INVALID, "more than 2^32 reads in one batch");
    const uint32_t rpb = c->cfg.reads_per_block, k = c->cfg.kmer_size;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    c->stats = leon_dna_stats{};
    const uint64_t* const walk_keys = c->walk_keys;              // (leon_dna_debug_walk_order applies to THIS batch only, whatever becomes of it)
    c->walk_keys = nullptr;
    static const bool trace_step = getenv("LEON_TRACE_STEP") != nullptr;   // measurement aid: host time of a batch's first milestones, on stderr
    const auto t_enter = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (trace_step) fprintf(stderr, "[leon step] %-34s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count());
    };

    // ---- sizes ----
    uint64_t off_first = 0, off_last = 0;
    HIPCHK(c, hipMemcpyAsync(c->h_rb + 0, d_off, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(c->h_rb + 1, d_off + n, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, spin_sync(s));
    off_first = c->h_rb[0]; off_last = c->h_rb[1];
    if (off_last < off_first) return fail(c, LEON_E_INVALID, "offsets are not monotonic");
    const uint64_t n_bases = off_last - off_first;
    (void)n_bases;
    const uint64_t n_blocks = (n + rpb - 1) / rpb;

    HIPCHK(c, hipEventRecord(c->ev[0], s));
    // ---- pack ----
    HIPCHK(c, c->slot_off.ensure((n + 1) * 8));
    HIPCHK(c, hipMemsetAsync(c->counters.as<uint32_t>() + 4, 0, 4, s));
    launch_read_slots(s, d_off, n, c->slot_off.as<uint64_t>(), c->counters.as<uint32_t>() + 4);
    size_t tmp_bytes = 0;
    HIPCHK(c, prim::ExclusiveSum(nullptr, tmp_bytes, c->slot_off.as<uint64_t>(), c->slot_off.as<uint64_t>(), n + 1, s));
    if (int rc = ensure_cub(c, tmp_bytes)) return rc;
    HIPCHK(c, prim::ExclusiveSum(c->cub_tmp.p, tmp_bytes, c->slot_off.as<uint64_t>(), c->slot_off.as<uint64_t>(), n + 1, s));
    uint64_t n_slots = 0;
    uint32_t bad_offsets = 0;
    HIPCHK(c, hipMemcpyAsync(c->h_rb + 0, c->slot_off.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(c->h_rb + 1, c->counters.as<uint32_t>() + 4, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, spin_sync(s));
    n_slots = c->h_rb[0]; bad_offsets = (uint32_t)c->h_rb[1];
    if (bad_offsets) return fail(c, LEON_E_INVALID, "offsets are not monotonic (or a read is longer than 2^31 bases)");
    mark("offsets checked, slots scanned");
    HIPCHK(c, c->packed.ensure((n_slots * 2 + 16) * 4));     // wave loads reach 12 dwords past a pass start
    HIPCHK(c, c->nmask.ensure((n_slots + 4) * 4));
    HIPCHK(c, c->rlen.ensure(n * 4));
    HIPCHK(c, c->ncount.ensure(n * 4));
    HIPCHK(c, hipMemsetAsync(c->packed.as<uint32_t>() + n_slots * 2, 0, 64, s));
    // the first resolution window's reads are packed first, later groups right before their first window: the host thread
    // that codes the dictionary stream (the longest single piece of a step) gets its first anchors ~25 ms earlier, and
    // with host input the upload of a group overlaps the resolution of the groups before it
    uint64_t packed_upto = 0;
    uint32_t n_pack_ev = 0;
    auto pack_group = [&]() -> int {
        const uint64_t a = packed_upto, b = group_end(a, n, c->cfg.resolve_window, up != nullptr);
        if (up) {
            // the group's last base, from the caller's offsets: checked before it is waited for (whatever the entries between
            // the group boundaries are, the device checks them one by one and refuses the batch)
            const uint64_t want = up_off[b];
            if (want < up_off[0] || want > up_off[n] || want < up_off[a]) return fail(c, LEON_E_INVALID, "offsets are not monotonic");
            while (up->bytes_done.load(std::memory_order_acquire) < want - up_off[0] && !up->failed.load()) std::this_thread::yield();
            if (up->failed.load()) return fail(c, LEON_E_HIP, "upload of the read bases failed");
        }
        if (c->pack_ev.size() < 2 * (size_t)(n_pack_ev + 1)) {
            hipEvent_t e0, e1;
            HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
            c->pack_ev.push_back(e0); c->pack_ev.push_back(e1);
        }
        HIPCHK(c, hipEventRecord(c->pack_ev[2 * n_pack_ev], s));
        launch_pack(s, d_bases, d_off + a, c->slot_off.as<uint64_t>() + a, b - a, c->packed.as<uint32_t>(), c->nmask.as<uint32_t>(),
                    c->rlen.as<uint32_t>() + a, c->ncount.as<uint32_t>() + a);
        HIPCHK(c, hipEventRecord(c->pack_ev[2 * n_pack_ev + 1], s));
        n_pack_ev++;
        packed_upto = b;
        return LEON_OK;
    };
    if (int rc = pack_group()) return rc;
    HIPCHK(c, hipEventRecord(c->ev[1], s));
    ReadsDev R = reads_view(c, d_off, n);

    // ---- anchor resolution ----
    HIPCHK(c, c->status.ensure(n));
    HIPCHK(c, c->hit_pos.ensure(n * 4)); HIPCHK(c, c->hit_slot.ensure(n * 4));
    HIPCHK(c, c->cand_pos.ensure(n * 4)); HIPCHK(c, c->cand_slot.ensure(n * 4));
    HIPCHK(c, c->anchor_pos.ensure(n * 4)); HIPCHK(c, c->anchor_addr.ensure(n * 4));
    HIPCHK(c, c->flags.ensure(n)); HIPCHK(c, c->sort_key.ensure(n * 8));
    const uint64_t W = std::min<uint64_t>(c->cfg.resolve_window, n);
    HIPCHK(c, c->ins_flag.ensure(W * 4)); HIPCHK(c, c->rank.ensure(W * 4));
    HIPCHK(c, c->ulist0.ensure(W * 4)); HIPCHK(c, c->ulist1.ensure(W * 4));
    ResolveDev V{};
    V.status = c->status.as<uint8_t>(); V.hit_pos = c->hit_pos.as<uint32_t>(); V.hit_slot = c->hit_slot.as<uint32_t>();
    V.cand_pos = c->cand_pos.as<uint32_t>(); V.cand_slot = c->cand_slot.as<uint32_t>();
    V.anchor_pos = c->anchor_pos.as<int32_t>(); V.anchor_addr = c->anchor_addr.as<uint32_t>(); V.flags = c->flags.as<uint8_t>();
    V.sort_key = c->sort_key.as<uint64_t>(); V.ins_flag = c->ins_flag.as<uint32_t>();
    uint32_t* counters = c->counters.as<uint32_t>();            // [0],[1]: list counts
    uint32_t* lists[2] = { c->ulist0.as<uint32_t>(), c->ulist1.as<uint32_t>() };
    size_t scan_tmp = 0;
    HIPCHK(c, prim::ExclusiveSum(nullptr, scan_tmp, V.ins_flag, c->rank.as<uint32_t>(), W, s));
    if (int rc = ensure_cub(c, scan_tmp)) return rc;
    c->poisoned = true;             // from here on the dictionary and the dictionary stream change: cleared on success
    // Host round trips: a wait costs 20-40 us and the stage used to make ~6 per window (580 per 100 M reads: the window's first count,
    // one per fixpoint round, the insert count, the new anchors' copy).  Now two: the rounds are launched AHEAD of their counts -- the
    // kernels take the list lengths from device memory, an empty list costs a launch that finds nothing to do -- three at once, then two
    // at a time for the few windows that need more, with every round's count copied to a small history that comes back with the
    // next wait; and a window's new anchors travel to pinned memory behind the kernels and are handed to the dictionary chain at
    // the NEXT window's first wait (the first window's at once: the chain, the longest piece of a step, starts with them).
    const uint64_t KW = kmer_words(k);                              // 64-bit words per anchor k-mer
    // The rounds are NOT asked to finish: their count is the longest chain of reads each waiting for the one before it, which no valid
    // input bounds (reads in genome-position order make one chain of a whole window).  After kRoundsAhead rounds -- more while the
    // list keeps halving, kRoundsMax at most -- what is left goes, in read order, through the exact sequential pass (chain_tail below).
    // LEON_RESOLVE_ROUNDS=a[:m]: rounds launched ahead / at most (measurement aid; any values give the same bytes).
    const uint32_t kRoundsAhead = [] { const char* e = getenv("LEON_RESOLVE_ROUNDS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 32 ? (uint32_t)v : 3u; }();
    const uint32_t kRoundsMax = [&] { const char* e = getenv("LEON_RESOLVE_ROUNDS"); const char* q = e ? strchr(e, ':') : nullptr; const int v = q ? atoi(q + 1) : 0;
                                      return std::max<uint32_t>(kRoundsAhead, v >= 1 && v <= 62 ? (uint32_t)v : 9u); }();
    constexpr uint32_t kRoundsMore = 2, kHist = 64;
    HIPCHK(c, c->round_hist.ensure(kHist * 4));
    uint32_t* d_hist = c->round_hist.as<uint32_t>();
    if (c->h_anchor_cap < W * 8 * KW) {
        for (auto& b : c->h_anchor) { if (b) HIPCHK(c, hipHostFree(b)); b = nullptr; }
        c->h_anchor_cap = 0;
        for (auto& b : c->h_anchor) HIPCHK(c, hipHostMalloc((void**)&b, W * 8 * KW + 64, hipHostMallocDefault));
        c->h_anchor_cap = W * 8 * KW;
    }
    struct Pending { int buf = -1; uint64_t n = 0; } pending;      // a window's new anchors on their way to pinned memory
    auto hand_over = [&]() {                                       // (called right after a wait: the copy has landed)
        if (pending.buf < 0) return;
        std::vector<uint64_t> fresh(c->h_anchor[pending.buf], c->h_anchor[pending.buf] + pending.n * KW);
        c->anchor_worker->push(std::move(fresh));
        pending.buf = -1;
    };
    // LEON_TRACE_RESOLVE=1 (measurement aid): what a read costs k_lookup_cand by its outcome, on stderr at the end of the stage
    static const bool trace_resolve = getenv("LEON_TRACE_RESOLVE") != nullptr;
    unsigned long long* d_trace = nullptr;
    if (trace_resolve) { HIPCHK(c, c->resolve_trace.ensure(12 * 8)); d_trace = c->resolve_trace.as<unsigned long long>(); HIPCHK(c, hipMemsetAsync(d_trace, 0, 12 * 8, s)); }
    // the look-ups shared out among the ranks of a job (leon_dna_set_gather; LEON_XCH_EMULATE plays the other ranks.  LEON_XCH_LOOKUPS=0: off, a measurement aid)
    static const bool lookups_env_off = [] { const char* e = getenv("LEON_XCH_LOOKUPS"); return e && atoi(e) == 0; }();
    const uint32_t Wn = c->shard_world;
    const bool share_lookups = Wn > 1 && !lookups_env_off && ((c->xch_mode == LEON_XCH_BY_ANCHOR && c->gather_fn) || c->xch_mode == LEON_XCH_EMULATE);
    float lk_call_ms = 0, lk_emul_ms = 0;
    int anchor_buf = 0;
    uint32_t hint = (uint32_t)std::min<uint64_t>(W, 1u << 20);      // grid-size hint of a window's first round (any size is correct: grid-stride loops)
    // ---- the exact sequential pass behind the rounds (dna_kernels.hip, k_chain_*): the `left` reads the rounds did not settle ----
    static const bool trace_chain = getenv("LEON_TRACE_CHAIN") != nullptr;
    uint32_t n_chain_ev = 0;
    auto chain_tail = [&](uint32_t left, uint64_t w0, uint64_t w1, uint32_t* clist) -> int {
        if (c->chain_ev.size() < 2 * (size_t)(n_chain_ev + 1)) {
            hipEvent_t e0, e1;
            HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
            c->chain_ev.push_back(e0); c->chain_ev.push_back(e1);
        }
        HIPCHK(c, hipEventRecord(c->chain_ev[2 * n_chain_ev], s));
        // in read order: flags over the window, their ranks (a read's chain index), the compacted list
        uint32_t* rank = c->rank.as<uint32_t>();
        launch_chain_flags(s, V, w0, w1, V.ins_flag);
        HIPCHK(c, prim::ExclusiveSum(c->cub_tmp.p, scan_tmp, V.ins_flag, rank, w1 - w0, s));
        launch_chain_compact(s, V, w0, w1, rank, clist);
        // (LEON_CHAIN_CHUNK: reads per k_chain_seq, at most the 2^CHAIN_LOG2 its LDS holds a bit for -- a test hook: the path a window of more than
        // half a million unsettled reads takes, chunk after chunk with tent re-proposed in between, on inputs the oracle codes in seconds;
        // read at every call: the tests change it inside one process)
        const uint32_t CH = [] { const char* e = getenv("LEON_CHAIN_CHUNK"); const long v = e ? atol(e) : 0; return v >= 1 && v <= (1l << CHAIN_LOG2) ? (uint32_t)v : 1u << CHAIN_LOG2; }();
        unsigned long long* d_ctrace = nullptr;
        if (trace_chain) { HIPCHK(c, c->chain_trace.ensure(8 * 8)); d_ctrace = c->chain_trace.as<unsigned long long>(); HIPCHK(c, hipMemsetAsync(d_ctrace, 0, 8 * 8, s)); }
        for (uint32_t c0 = 0; c0 < left; c0 += CH) {
            const uint32_t nc = std::min(CH, left - c0), nG = (nc + 63) / 64;
            // a later chunk: tent still names reads of the chunk before (settled now) -- cleared, and what is left proposes again
            if (c0) launch_chain_repropose(s, c->D, V, first_read_index, clist + (c0 - CH), left - (c0 - CH), clist + c0, left - c0);
            HIPCHK(c, c->chain_cnt.ensure((uint64_t)nc * 4)); HIPCHK(c, c->chain_own.ensure((uint64_t)nc * 4));
            HIPCHK(c, c->chain_ins.ensure(((uint64_t)nG + 1) * 8)); HIPCHK(c, c->chain_rows.ensure(((uint64_t)nG + 1) * 8));
            HIPCHK(c, c->chain_dep.ensure((uint64_t)nc * 8)); HIPCHK(c, c->chain_xdep.ensure((uint64_t)nc * 8)); HIPCHK(c, c->chain_om.ensure((uint64_t)nG * 64 * 8)); HIPCHK(c, c->chain_late.ensure((uint64_t)nG * 8));
            uint64_t* rows = c->chain_rows.as<uint64_t>();
            unsigned long long* om = c->chain_om.as<unsigned long long>(); unsigned long long* late = c->chain_late.as<unsigned long long>();
            unsigned long long* dep = c->chain_dep.as<unsigned long long>(); unsigned long long* xdep = c->chain_xdep.as<unsigned long long>();
            size_t rows_tmp = 0;
            HIPCHK(c, prim::ExclusiveSum(nullptr, rows_tmp, rows, rows, nG + 1, s));
            if (rows_tmp > c->cub_tmp.cap) { HIPCHK(c, hipStreamSynchronize(s)); if (int rc = ensure_cub(c, std::max(rows_tmp, scan_tmp))) return rc; }
            launch_chain_prep(s, false, R, c->D, V, first_read_index, w0, clist + c0, nc, c0, rank, c->chain_cnt.as<uint32_t>(), c->chain_own.as<uint32_t>(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
            launch_chain_tables(s, c->chain_cnt.as<uint32_t>(), c->chain_own.as<uint32_t>(), nc, rows, om, late);
            HIPCHK(c, prim::ExclusiveSum(c->cub_tmp.p, rows_tmp, rows, rows, nG + 1, s));
            HIPCHK(c, hipMemcpyAsync(c->h_rb + 0, rows + nG, 8, hipMemcpyDeviceToHost, s));
            HIPCHK(c, spin_sync(s));
            const uint64_t total_rows = c->h_rb[0];
            HIPCHK(c, c->chain_ent.ensure(std::max<uint64_t>(total_rows, 1) * 64 * 4));
            launch_chain_prep(s, true, R, c->D, V, first_read_index, w0, clist + c0, nc, c0, rank, c->chain_cnt.as<uint32_t>(), c->chain_own.as<uint32_t>(), rows, c->chain_ent.as<uint32_t>(), om, late, dep, xdep);
            if (launch_chain_seq(s, nc, c->chain_cnt.as<uint32_t>(), c->chain_own.as<uint32_t>(), dep, xdep, rows, c->chain_ent.as<uint32_t>(), c->chain_ins.as<unsigned long long>(), d_ctrace))
                return fail(c, LEON_E_HIP, "the sequential resolution pass could not be launched (its LDS request was refused)");
            launch_chain_apply(s, c->D, V, first_read_index, clist + c0, nc, c->chain_ins.as<unsigned long long>(), k);
            HIPCHK(c, hipGetLastError());
        }
        // (the rounds end with every tent they touched cleared; so must this: a tent that still named a settled read would block whoever
        // proposes the key in a later window)
        { const uint32_t c_last = (left - 1) / CH * CH; launch_chain_repropose(s, c->D, V, first_read_index, clist + c_last, left - c_last, nullptr, 0); }
        HIPCHK(c, hipEventRecord(c->chain_ev[2 * n_chain_ev + 1], s));
        n_chain_ev++;
        c->stats.resolve_chain_reads += left; c->stats.resolve_chain_windows++;
        if (trace_chain) {
            unsigned long long t[8];
            HIPCHK(c, hipStreamSynchronize(s));
            HIPCHK(c, hipMemcpy(t, d_ctrace, sizeof t, hipMemcpyDeviceToHost));
            float cms = 0; (void)hipEventElapsedTime(&cms, c->chain_ev[2 * n_chain_ev - 2], c->chain_ev[2 * n_chain_ev - 1]);
            fprintf(stderr, "[leon chain] window [%llu, %llu): %u reads left by the rounds; %llu steps of 64, %.2f ballot iterations and %.1f entry rows per step, %llu inserters; %.2f ms; per step the settler waited %.0f (tester: %.0f), settled + published %.0f ticks of s_memtime\n",
                    (unsigned long long)w0, (unsigned long long)w1, left, t[0], t[0] ? (double)t[1] / t[0] : 0.0, t[0] ? (double)t[2] / t[0] : 0.0, t[3], cms,
                    t[0] ? (double)t[4] / t[0] : 0.0, t[0] ? (double)t[5] / t[0] : 0.0, t[0] ? (double)t[6] / t[0] : 0.0);
        }
        return LEON_OK;
    };
    for (uint64_t w0 = 0, w1 = 0; w0 < n; w0 = w1) {
        w1 = std::min(n, w0 + (w0 == 0 ? first_window(W) : W));
        if (int rc = dict_reserve(c, c->n_keys + (w1 - w0))) return rc;
        HIPCHK(c, hipMemsetAsync(counters, 0, 8, s));
        HIPCHK(c, hipMemsetAsync(c->wbits.p, 0, (1ull << WBITS_LOG2) / 8, s));
        HIPCHK(c, hipMemsetAsync(c->pbits.p, 0, (1ull << WBITS_LOG2) / 8, s));
        if (!share_lookups) launch_lookup_cand(s, R, c->B, c->d_rv16, c->D, V, w0, w1, first_read_index, lists[0], counters, d_trace);
        else {
            // The window's look-ups divided among the ranks (leon_dna_set_gather): rank r takes the r-th run of P reads, everybody
            // learns what everybody found from ONE all-gather of a word per read -- the caller's -- and makes the other runs' results
            // its own (k_lookup_apply): every rank's dictionary goes on holding every proposal, as if it had looked everything up.
            const uint64_t P = (w1 - w0 + Wn - 1) / Wn;
            auto run_of = [&](uint32_t r, uint64_t& a, uint64_t& b) { a = std::min(w1, w0 + (uint64_t)r * P); b = std::min(w1, a + P); };
            HIPCHK(c, c->xch_res.ensure((uint64_t)Wn * P * 8));
            uint64_t* xres = c->xch_res.as<uint64_t>();
            uint64_t s0 = 0, s1 = 0;
            run_of(c->shard_rank, s0, s1);
            launch_lookup_cand(s, R, c->B, c->d_rv16, c->D, V, s0, s1, first_read_index, lists[0], counters, d_trace, xres, w0, false);
            HIPCHK(c, hipStreamSynchronize(s));
            const auto t_l0 = std::chrono::steady_clock::now();
            if (c->xch_mode == LEON_XCH_BY_ANCHOR) {
                if (c->gather_fn(c->gather_user, xres, P * 8, Wn)) return fail(c, LEON_E_STATE, "the gather callback returned non-zero");
                HIPCHK(c, hipSetDevice(c->device));               // (the callback may have changed the thread's device)
                lk_call_ms += (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_l0).count();
            } else {                                              // no other rank present: their runs computed here, leaving nothing but their words
                for (uint32_t r = 0; r < Wn; r++) {
                    if (r == c->shard_rank) continue;
                    uint64_t a = 0, b = 0;
                    run_of(r, a, b);
                    launch_lookup_cand(s, R, c->B, c->d_rv16, c->D, V, a, b, first_read_index, lists[0], counters, nullptr, xres, w0, true);
                }
                HIPCHK(c, hipStreamSynchronize(s));
                lk_emul_ms += (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_l0).count();
            }
            launch_lookup_apply(s, R, c->D, V, w0, w1, s0, s1, first_read_index, xres, lists[0], counters);
        }
        HIPCHK(c, hipMemcpyAsync(d_hist, counters, 4, hipMemcpyDeviceToDevice, s));         // hist[0]: the window's unresolved reads
        int cur = 0;
        uint32_t n_hist = 1, cnt = 0, cnt0 = 0;
        bool first_wait = true;
        // (Fewer rounds ahead for a file whose last window left the sequential pass most of its list -- position-sorted reads -- were
        // measured and are slower: rounds 2 and 3 settle the reads that contain round 1's inserters, 15 M of a sorted 100 M-read
        // file's, in 180 ms; the sequential pass takes 18 ns for each of them.  1 929 ms against 1 838 for the stage.)
        for (uint32_t ahead = kRoundsAhead;; ahead = kRoundsMore) {
            for (uint32_t r = 0; r < ahead && n_hist < kHist; r++) {
                const int nxt = cur ^ 1;
                const uint32_t h = std::max<uint32_t>(hint >> (2 * (n_hist - 1) < 31 ? 2 * (n_hist - 1) : 31), 4096);
                HIPCHK(c, hipMemsetAsync(counters + nxt, 0, 4, s));
                launch_check(s, R, c->D, V, first_read_index, lists[cur], counters + cur, h, lists[nxt], counters + nxt);
                HIPCHK(c, hipMemcpyAsync(d_hist + n_hist, counters + nxt, 4, hipMemcpyDeviceToDevice, s));
                launch_reset_tent(s, c->D, V, lists[cur], counters + cur, h);
                launch_propose(s, c->D, V, first_read_index, lists[nxt], counters + nxt, h);
                cur = nxt; n_hist++;
            }
            HIPCHK(c, hipMemcpyAsync(c->h_rb + 8, d_hist, n_hist * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(c->h_rb + 1, c->D.err, 4, hipMemcpyDeviceToHost, s));
            HIPCHK(c, spin_sync(s));
            if (first_wait) { hand_over(); first_wait = false; }
            if ((int)(uint32_t)c->h_rb[1]) {
                HIPCHK(c, hipMemsetAsync(c->D.err, 0, 4, s));
                if ((int)(uint32_t)c->h_rb[1] == 2)
                    return fail(c, LEON_E_STATE, "the look-ups gathered from the other ranks do not fit this rank's reads or dictionary (were all ranks fed the same batches?)");
                return fail(c, LEON_E_STATE, "anchor dictionary: a two-word key stayed half-written (a stalled wave); batch abandoned");
            }
            const uint32_t* hist = reinterpret_cast<const uint32_t*>(c->h_rb + 8);
            cnt0 = hist[0];
            for (uint32_t r = 1; r < n_hist; r++)
                if (hist[r - 1] > 0 && hist[r] >= hist[r - 1]) return fail(c, LEON_E_STATE, "anchor resolution made no progress (internal error)");
            cnt = hist[n_hist - 1];
            if (cnt == 0) {
                for (uint32_t r = 1; r < n_hist; r++) if (hist[r - 1] > 0) c->stats.resolve_rounds++;
                break;
            }
            // more rounds only while they pay: the list at least halved in the last round and the budget is not spent
            const bool halving = n_hist >= 2 && 2ull * hist[n_hist - 1] <= hist[n_hist - 2];
            if (!halving || n_hist - 1 >= kRoundsMax || n_hist + kRoundsMore > kHist) {
                for (uint32_t r = 1; r < n_hist; r++) if (hist[r - 1] > 0) c->stats.resolve_rounds++;
                if (int rc = chain_tail(cnt, w0, w1, lists[cur ^ 1])) return rc;
                break;
            }
        }
        hint = std::max<uint32_t>(2 * cnt0, 4096);
        if (cnt0 > 0) {
            launch_final_pos(s, R, c->D, V, w0, w1, first_read_index);
            launch_ins_flags(s, V, w0, w1);
            HIPCHK(c, prim::ExclusiveSum(c->cub_tmp.p, scan_tmp, V.ins_flag, c->rank.as<uint32_t>(), w1 - w0, s));
            uint32_t last_rank = 0, last_flag = 0;
            HIPCHK(c, hipMemcpyAsync(c->h_rb + 0, c->rank.as<uint32_t>() + (w1 - w0 - 1), 4, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(c->h_rb + 1, V.ins_flag + (w1 - w0 - 1), 4, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(c->h_rb + 2, c->d_nkeys, 8, hipMemcpyDeviceToHost, s));
            HIPCHK(c, spin_sync(s));
            last_rank = (uint32_t)c->h_rb[0]; last_flag = (uint32_t)c->h_rb[1]; c->n_keys = c->h_rb[2];
            uint64_t n_new = (uint64_t)last_rank + last_flag;
            if (c->n_anchors + n_new > 0xFFFFFFFFull) return fail(c, LEON_E_OVERFLOW, "more than 2^32 anchors");
            if ((c->n_anchors + n_new) * 8 * KW > c->anchor_kmers.cap) {      // grow, keeping what is there
                TmpBuf nb;
                HIPCHK(c, nb.ensure(std::max<uint64_t>((c->n_anchors + n_new) * 2, 1024) * 8 * KW));
                if (c->n_anchors) HIPCHK(c, hipMemcpyAsync(nb.p, c->anchor_kmers.p, c->n_anchors * 8 * KW, hipMemcpyDeviceToDevice, s));
                HIPCHK(c, hipStreamSynchronize(s));
                std::swap(static_cast<DevBuf&>(nb).p, c->anchor_kmers.p);        // nb now holds the old buffer and releases it
                std::swap(static_cast<DevBuf&>(nb).cap, c->anchor_kmers.cap);
            }
            launch_assign_addr(s, c->D, V, w0, w1, c->rank.as<uint32_t>(), c->n_anchors, c->anchor_kmers.as<uint64_t>(), k);
            if (n_new && c->shard_rank == 0 && !(c->cfg.flags & LEON_F_DICT_ON_DEVICE)) {   // the window's new anchors, for the host thread coding the dictionary stream
                HIPCHK(c, hipMemcpyAsync(c->h_anchor[anchor_buf], c->anchor_kmers.as<uint64_t>() + c->n_anchors * KW, n_new * 8 * KW, hipMemcpyDeviceToHost, s));
                pending.buf = anchor_buf; pending.n = n_new;
                anchor_buf ^= 1;
                if (w0 == 0) {
                    HIPCHK(c, spin_sync(s));
                    hand_over();
                    mark("first window's anchors to the chain");
                }
            }
            c->n_anchors += n_new;
        }
        launch_finalize_reads(s, R, c->D, V, w0, w1);
        while (w1 < n && packed_upto < std::min(n, w1 + W)) { if (int rc = pack_group()) return rc; }   // the next window's reads
        c->stats.resolve_windows++;
    }
    if (pending.buf >= 0) { HIPCHK(c, spin_sync(s)); hand_over(); }
    if (trace_resolve) {
        unsigned long long t[12];
        HIPCHK(c, hipMemcpy(t, d_trace, sizeof t, hipMemcpyDeviceToHost));
        const char* cls[3] = {"found an anchor of the dictionary", "went on to propose its own", "no anchor at all"};
        for (int j = 0; j < 3; j++)
            if (t[4 * j]) fprintf(stderr, "[leon resolve] k_lookup_cand, reads that %-34s: %10llu reads, per read %.1f filter probes, %.1f dictionary probes behind a filter maybe, %.1f k-mers through the bloom\n",
                                  cls[j], t[4 * j], (double)t[4 * j + 1] / t[4 * j], (double)t[4 * j + 2] / t[4 * j], (double)t[4 * j + 3] / t[4 * j]);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev[2], s));
    mark("resolution launched to its end");
    // ---- this rank's share of the batch: a contiguous range of whole blocks (all of it when not sharded) ----
    uint64_t lb0 = 0, lb1 = n_blocks;
    if (c->shard_world > 1) {
        const uint64_t q = n_blocks / c->shard_world, rm = n_blocks % c->shard_world, rk = c->shard_rank;
        lb0 = rk * q + std::min<uint64_t>(rk, rm);
        lb1 = lb0 + q + (rk < rm ? 1 : 0);
    }
    const uint64_t nbl = lb1 - lb0;
    const uint64_t r0 = std::min<uint64_t>(n, lb0 * rpb), r1 = std::min<uint64_t>(n, lb1 * rpb), nl = r1 - r0;
    uint64_t off_r0 = off_first, off_r1 = off_last;
    if (c->shard_world > 1) {
        HIPCHK(c, hipMemcpy(&off_r0, d_off + r0, 8, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(&off_r1, d_off + r1, 8, hipMemcpyDeviceToHost));
    }
    const uint64_t nl_bases = off_r1 - off_r0;
    R.ev_origin = r0;
    auto ms = [&](int a, int b) { float v = 0; (void)hipEventElapsedTime(&v, c->ev[a], c->ev[b]); return v; };
    auto chain_ms = [&]() { float t = 0; for (uint32_t e = 0; e < n_chain_ev; e++) { float v = 0; (void)hipEventElapsedTime(&v, c->chain_ev[2 * e], c->chain_ev[2 * e + 1]); t += v; } return t; };
    c->stats.n_reads = nl; c->stats.n_bases = nl_bases; c->stats.n_blocks = nbl; c->stats.n_anchors = c->n_anchors;
    c->last_n = n; c->last_bases = nl_bases;
    c->last_d_bases = d_bases; c->last_d_off = d_off;
    const bool by_anchor = c->shard_world > 1 && c->xch_mode != LEON_XCH_OFF;
    if (nl == 0 && !(by_anchor && c->xch_mode == LEON_XCH_BY_ANCHOR)) {   // nothing of this batch is ours to encode (and nobody waits for our slice)
        HIPCHK(c, hipStreamSynchronize(s));
        float pack2 = 0; for (uint32_t e = 1; e < n_pack_ev; e++) { float v = 0; (void)hipEventElapsedTime(&v, c->pack_ev[2 * e], c->pack_ev[2 * e + 1]); pack2 += v; }
        c->stats.ms_pack = ms(0, 1) + pack2; c->stats.ms_resolve = ms(1, 2) - pack2; c->stats.ms_resolve_chain = chain_ms(); c->stats.ms_total = ms(0, 2);
        c->next_read += n; c->next_block += n_blocks;
        if (n % rpb) c->partial_seen = true;
        c->poisoned = false;
        return LEON_OK;
    }
    HIPCHK(c, c->events.ensure(nl_bases + 16));
    HIPCHK(c, hipMemsetAsync(c->events.p, 0, nl_bases + 16, s));
    // the walk's path cache: a 64-byte bucket per ~8 solid k-mers (the bloom's size says how many there are), at most an eighth of the
    // free device memory; EMPTY again at every batch -- what a batch's walkers learn from the bloom is shared among THEM, a later
    // batch (or a bench step) starts cold.  LEON_WALK_CACHE=0: off; LEON_WALK_CACHE_LOG2: log2 of the buckets (measurement)
    WalkCache wc{nullptr, 0, 28, 0};
    {
        static const int wc_env = [] { const char* e = getenv("LEON_WALK_CACHE"); return e ? atoi(e) : 1; }();
        static const int wc_log2 = [] { const char* e = getenv("LEON_WALK_CACHE_LOG2"); return e ? atoi(e) : 0; }();
        // (not for a rank of four or more: the walkers that cover one genome region are spread over all ranks' slices, a rank's own cache
        // would answer a fifth of its look-ups and cost as much as it saves)
        if (wc_env && c->B.n_hash == 7 && c->shard_world <= 2) {
            uint64_t want = c->cfg.bloom_tai / 12 / 8, buckets = 1024;
            while (buckets < want && buckets < (1ull << 27)) buckets <<= 1;          // at most 2^27 buckets = 8 GiB
            if (wc_log2 >= 10 && wc_log2 <= 31) buckets = 1ull << wc_log2;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) while (buckets > 1024 && buckets * 64 > c->wcache.cap && buckets * 64 > free_b / 8) buckets >>= 1;
            if (c->wcache.ensure(buckets * 64) == hipSuccess) {
                wc.slots = c->wcache.as<uint64_t>(); wc.bucket_mask = buckets - 1;
                static const int hop_log2 = [] { const char* e = getenv("LEON_WALK_HOP_LOG2"); const int v = e ? atoi(e) : 4; return v >= 1 && v <= 8 ? v : 4; }();
                wc.hop_shift = 32 - hop_log2;

                launch_walk_cache_init(s, wc, k);

            } else (void)hipGetLastError();
        }
    }
    HIPCHK(c, c->perm.ensure(n * 4));
    hipLaunchKernelGGL(k_iota, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 8192)), dim3(256), 0, s, c->perm.as<uint32_t>(), n);
    size_t sort_tmp = 0;
    if (!by_anchor) {
        // ---- sort the share's reads by (anchor address, strand) ----
        HIPCHK(c, c->sort_key2.ensure(nl * 8)); HIPCHK(c, c->perm2.ensure(nl * 4));
        // (measurement hook: another order of the reads in the walk changes which lanes share bloom sectors, never the bytes --
        // events are indexed by read position)
        const uint64_t* walk_key = walk_keys ? walk_keys + r0 : V.sort_key + r0;
        const unsigned key_bits = walk_keys ? 48u : 33u;
        HIPCHK(c, prim::SortPairs(nullptr, sort_tmp, walk_key, c->sort_key2.as<uint64_t>(), c->perm.as<uint32_t>() + r0,
                                                     c->perm2.as<uint32_t>(), nl, 0, key_bits, s));
        if (int rc = ensure_cub(c, sort_tmp)) return rc;
        HIPCHK(c, prim::SortPairs(c->cub_tmp.p, sort_tmp, walk_key, c->sort_key2.as<uint64_t>(), c->perm.as<uint32_t>() + r0,
                                                     c->perm2.as<uint32_t>(), nl, 0, key_bits, s));
        HIPCHK(c, hipEventRecord(c->ev[3], s));
        // ---- walk ----
        HIPCHK(c, hipEventRecord(c->ev[4], s));
        launch_walk(s, R, c->B, c->d_rv16, V.anchor_pos, V.flags, c->perm2.as<uint32_t>(), nl, c->events.as<uint8_t>(), nullptr, wc);
        HIPCHK(c, hipEventRecord(c->ev[5], s));
        c->stats.walk_launches = 1; c->stats.walk_reads = nl;
    } else {
        // ---- the walk divided by anchor (leon_dna_set_exchange): ALL of the batch's reads sorted by anchor address, cut into `world`
        // slices of equal size; this rank walks its slice into a buffer of its own and what it found goes to the ranks that code
        // the reads' blocks, as (place in that rank's event buffer, byte) words grouped by destination ----
        const uint32_t Wd = c->shard_world, me = c->shard_rank;
        if (Wd + 2 > 4096 / 8) return fail(c, LEON_E_INVALID, "set_exchange: world too large");
        HIPCHK(c, c->sort_key2.ensure(n * 8)); HIPCHK(c, c->perm2.ensure(n * 4));
        HIPCHK(c, prim::SortPairs(nullptr, sort_tmp, V.sort_key, c->sort_key2.as<uint64_t>(), c->perm.as<uint32_t>(), c->perm2.as<uint32_t>(), n, 0, 33, s));
        if (int rc = ensure_cub(c, sort_tmp)) return rc;
        HIPCHK(c, prim::SortPairs(c->cub_tmp.p, sort_tmp, V.sort_key, c->sort_key2.as<uint64_t>(), c->perm.as<uint32_t>(), c->perm2.as<uint32_t>(), n, 0, 33, s));
        unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(c->counters.as<uint32_t>() + 10);
        launch_lower_bound(s, c->sort_key2.as<uint64_t>(), n, 1ull << 32, d_cnt);            // reads without an anchor sort last: nothing to walk
        HIPCHK(c, hipMemcpyAsync(c->h_rb + 0, d_cnt, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, spin_sync(s));
        const uint64_t n_anch = c->h_rb[0];
        // the slices: equal WEIGHT (a read 1, an anchor group's first read SLICE_GROUP_WEIGHT more), the same cuts on every rank
        HIPCHK(c, c->xch_slot.ensure(n * 4)); HIPCHK(c, c->xch_off.ensure((n + 1) * 8));
        size_t scan_n = 0;
        HIPCHK(c, prim::ExclusiveSum(nullptr, scan_n, c->xch_off.as<uint64_t>(), c->xch_off.as<uint64_t>(), n + 1, s));
        if (int rc = ensure_cub(c, scan_n)) return rc;
        HIPCHK(c, c->xch_split.ensure((Wd + 1) * 8));
        launch_slice_weights(s, c->sort_key2.as<uint64_t>(), n_anch, c->xch_off.as<uint64_t>());
        HIPCHK(c, prim::ExclusiveSum(c->cub_tmp.p, scan_n, c->xch_off.as<uint64_t>(), c->xch_off.as<uint64_t>(), n_anch + 1, s));
        launch_slice_splits(s, c->xch_off.as<uint64_t>(), n_anch, Wd, c->xch_split.as<unsigned long long>());
        HIPCHK(c, hipMemcpyAsync(c->h_rb + 0, c->xch_split.p, (Wd + 1) * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipEventRecord(c->ev[3], s));
        HIPCHK(c, spin_sync(s));
        std::vector<uint64_t> slice_at(c->h_rb, c->h_rb + Wd + 1);
        // where each rank's reads begin in file order (block_range of every rank), for the words' grouping
        std::vector<uint64_t> rank_r0(Wd + 1);
        for (uint32_t d = 0; d <= Wd; d++) {
            const uint64_t q = n_blocks / Wd, rm = n_blocks % Wd;
            rank_r0[d] = std::min<uint64_t>(n, (d * q + std::min<uint64_t>(d, rm)) * rpb);
        }
        std::vector<uint64_t> send_counts(Wd, 0), send_at(Wd + 1, 0);
        // one slice: walked into xch_events, its words formed in xch_send (grouped by destination; send_at[d] = where rank d's begin)
        auto do_slice = [&](uint32_t sl, bool timed) -> int {
            const uint64_t s0 = slice_at[sl], s1 = slice_at[sl + 1], ns = s1 - s0;
            const uint32_t* slice = c->perm2.as<uint32_t>() + s0;
            HIPCHK(c, c->xch_evoff.ensure((ns + 1) * 8));
            HIPCHK(c, hipMemsetAsync(c->xch_slot.p, 0xFF, n * 4, s));
            launch_slice_reads(s, R, slice, ns, c->xch_evoff.as<uint64_t>(), c->xch_slot.as<uint32_t>());
            size_t tb = 0;
            HIPCHK(c, prim::ExclusiveSum(nullptr, tb, c->xch_evoff.as<uint64_t>(), c->xch_evoff.as<uint64_t>(), ns + 1, s));
            if (int rc = ensure_cub(c, std::max(tb, scan_n))) return rc;
            HIPCHK(c, prim::ExclusiveSum(c->cub_tmp.p, tb, c->xch_evoff.as<uint64_t>(), c->xch_evoff.as<uint64_t>(), ns + 1, s));
            HIPCHK(c, hipMemcpyAsync(c->h_rb + 0, c->xch_evoff.as<uint64_t>() + ns, 8, hipMemcpyDeviceToHost, s));
            HIPCHK(c, spin_sync(s));
            const uint64_t slice_bases = c->h_rb[0];
            HIPCHK(c, c->xch_events.ensure(slice_bases + 16));
            HIPCHK(c, hipMemsetAsync(c->xch_events.p, 0, slice_bases + 16, s));
            if (timed) HIPCHK(c, hipEventRecord(c->ev[4], s));
            launch_walk(s, R, c->B, c->d_rv16, V.anchor_pos, V.flags, slice, ns, c->xch_events.as<uint8_t>(), c->xch_evoff.as<uint64_t>(), wc);
            if (timed) { HIPCHK(c, hipEventRecord(c->ev[5], s)); c->stats.walk_launches = 1; c->stats.walk_reads = ns; }
            launch_ev_words(s, R, c->xch_slot.as<uint32_t>(), c->xch_evoff.as<uint64_t>(), c->xch_events.as<uint8_t>(), n, rpb, n_blocks, Wd,
                            c->xch_off.as<uint64_t>(), nullptr);
            HIPCHK(c, prim::ExclusiveSum(c->cub_tmp.p, scan_n, c->xch_off.as<uint64_t>(), c->xch_off.as<uint64_t>(), n + 1, s));
            for (uint32_t d = 0; d <= Wd; d++) HIPCHK(c, hipMemcpyAsync(c->h_rb + 1 + d, c->xch_off.as<uint64_t>() + rank_r0[d], 8, hipMemcpyDeviceToHost, s));
            HIPCHK(c, spin_sync(s));
            for (uint32_t d = 0; d <= Wd; d++) send_at[d] = c->h_rb[1 + d];
            for (uint32_t d = 0; d < Wd; d++) send_counts[d] = send_at[d + 1] - send_at[d];
            HIPCHK(c, c->xch_send.ensure(send_at[Wd] * 8 + 64));
            launch_ev_words(s, R, c->xch_slot.as<uint32_t>(), c->xch_evoff.as<uint64_t>(), c->xch_events.as<uint8_t>(), n, rpb, n_blocks, Wd,
                            c->xch_off.as<uint64_t>(), c->xch_send.as<uint64_t>());
            HIPCHK(c, hipGetLastError());
            return LEON_OK;
        };
        HIPCHK(c, hipMemsetAsync(c->errflag.as<int>() + 1, 0, 4, s));
        const auto t_x0 = std::chrono::steady_clock::now();     // (ms_exchange: everything of the division that is not the slice's walk itself)
        if (int rc = do_slice(me, true)) return rc;
        HIPCHK(c, hipStreamSynchronize(s));
        float walk_own = 0; (void)hipEventElapsedTime(&walk_own, c->ev[4], c->ev[5]);
        auto since = [](std::chrono::steady_clock::time_point t) { return (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
        c->stats.xch_words_sent = send_at[Wd];
        if (c->xch_mode == LEON_XCH_BY_ANCHOR) {
            const uint64_t* d_recv = nullptr; uint64_t recv_total = 0;
            const auto t_call = std::chrono::steady_clock::now();
            if (c->xch_fn(c->xch_user, c->xch_send.as<uint64_t>(), send_counts.data(), Wd, &d_recv, &recv_total))
                return fail(c, LEON_E_STATE, "the exchange callback returned non-zero");
            c->stats.ms_exchange_call = since(t_call);
            if (recv_total && !d_recv) return fail(c, LEON_E_STATE, "the exchange callback returned no buffer");
            c->stats.xch_words_received = recv_total;
            HIPCHK(c, hipSetDevice(c->device));                   // (the callback may have changed the thread's device)
            launch_ev_scatter(s, d_recv, recv_total, c->events.as<uint8_t>(), nl_bases, c->errflag.as<int>() + 1);
            HIPCHK(c, hipStreamSynchronize(s));                   // the caller's buffer is free again when this call returns
            c->stats.ms_exchange = since(t_x0) - walk_own;
        } else {
            // no other rank present: this context plays them all, one slice after the other, and keeps what is meant for its own rank
            launch_ev_scatter(s, c->xch_send.as<uint64_t>() + send_at[me], send_counts[me], c->events.as<uint8_t>(), nl_bases, c->errflag.as<int>() + 1);
            c->stats.xch_words_received = send_counts[me];
            HIPCHK(c, hipStreamSynchronize(s));
            c->stats.ms_exchange = since(t_x0) - walk_own;
            const auto t_e0 = std::chrono::steady_clock::now();
            for (uint32_t sl = 0; sl < Wd; sl++) {
                if (sl == me) continue;
                if (int rc = do_slice(sl, false)) return rc;
                launch_ev_scatter(s, c->xch_send.as<uint64_t>() + send_at[me], send_counts[me], c->events.as<uint8_t>(), nl_bases, c->errflag.as<int>() + 1);
                c->stats.xch_words_received += send_counts[me];
            }
            HIPCHK(c, hipStreamSynchronize(s));
            c->stats.ms_emulated = since(t_e0);
        }
        // (the window look-ups shared out among the ranks, earlier in this call: their gathers and, emulated, the other ranks' runs)
        c->stats.ms_exchange_call += lk_call_ms; c->stats.ms_exchange += lk_call_ms; c->stats.ms_gather_call = lk_call_ms;
        c->stats.ms_emulated += lk_emul_ms; c->stats.ms_emulated_lookups = lk_emul_ms;
        HIPCHK(c, hipEventRecord(c->ev[9], s));                  // the symbols stage begins here
        int xerr = 0;
        HIPCHK(c, hipMemcpy(&xerr, c->errflag.as<int>() + 1, 4, hipMemcpyDeviceToHost));
        if (xerr) return fail(c, LEON_E_STATE, "the exchange delivered a word that lies outside this rank's blocks");
        if (nl == 0) {                                           // our slice is delivered; no block of the batch is ours to code
            float pack2 = 0; for (uint32_t e = 1; e < n_pack_ev; e++) { float v = 0; (void)hipEventElapsedTime(&v, c->pack_ev[2 * e], c->pack_ev[2 * e + 1]); pack2 += v; }
            c->stats.ms_pack = ms(0, 1) + pack2; c->stats.ms_resolve = ms(1, 2) - pack2; c->stats.ms_resolve_chain = chain_ms(); c->stats.ms_sort = ms(2, 3); c->stats.ms_walk = ms(4, 5); c->stats.ms_total = ms(0, 5);
            c->next_read += n; c->next_block += n_blocks;
            if (n % rpb) c->partial_seen = true;
            c->poisoned = false;
            return LEON_OK;
        }
    }

    // ---- symbols ----
    HIPCHK(c, c->prev.ensure(n * 8));
    HIPCHK(c, c->sym_off.ensure((nl + 1) * 8));
    HIPCHK(c, c->nerr.ensure(nl * 8));                         // (two words per read: error positions, chunks that hold events)
    launch_prev_anchored(s, V.anchor_pos, n, rpb, lb0, nbl, c->prev.as<int64_t>());
    HIPCHK(c, hipMemsetAsync(c->sym_off.as<uint64_t>() + nl, 0, 8, s));
    launch_symbols(s, R, V.anchor_pos, V.anchor_addr, V.flags, c->prev.as<int64_t>(), c->events.as<uint8_t>(), r0, nl,
                   c->sym_off.as<uint64_t>(), c->nerr.as<uint32_t>(), nullptr);
    HIPCHK(c, prim::ExclusiveSum(nullptr, tmp_bytes, c->sym_off.as<uint64_t>(), c->sym_off.as<uint64_t>(), nl + 1, s));
    if (int rc = ensure_cub(c, tmp_bytes)) return rc;
    HIPCHK(c, prim::ExclusiveSum(c->cub_tmp.p, tmp_bytes, c->sym_off.as<uint64_t>(), c->sym_off.as<uint64_t>(), nl + 1, s));
    uint64_t n_syms = 0, max_block_syms = 0;
    unsigned long long* d_max = reinterpret_cast<unsigned long long*>(c->counters.as<uint32_t>() + 8);
    HIPCHK(c, hipMemsetAsync(d_max, 0, 8, s));
    launch_max_block_syms(s, c->sym_off.as<uint64_t>(), nl, rpb, nbl, d_max);
    HIPCHK(c, hipMemcpyAsync(&n_syms, c->sym_off.as<uint64_t>() + nl, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&max_block_syms, d_max, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    HIPCHK(c, c->syms.ensure(n_syms * 2 + 256));
    launch_symbols(s, R, V.anchor_pos, V.anchor_addr, V.flags, c->prev.as<int64_t>(), c->events.as<uint8_t>(), r0, nl,
                   c->sym_off.as<uint64_t>(), c->nerr.as<uint32_t>(), c->syms.as<uint8_t>());
    HIPCHK(c, c->blk_begin.ensure((nbl + 1) * 8)); HIPCHK(c, c->out_off.ensure((nbl + 1) * 8));
    HIPCHK(c, c->out_size.ensure(nbl * 8)); HIPCHK(c, c->dst_off.ensure((nbl + 1) * 8));
    launch_block_ranges(s, c->sym_off.as<uint64_t>(), nl, rpb, nbl, c->blk_begin.as<uint64_t>(), c->out_off.as<uint64_t>());
    HIPCHK(c, hipEventRecord(c->ev[6], s));

    // ---- range coder ----
    bool host_chains = rc_on_host(nbl, n_syms, max_block_syms, c->shard_world);
    std::vector<uint64_t> sizes(nbl), dst(nbl + 1, 0);
    uint64_t payload_bytes = 0;
    if (host_chains) {
        // a small launch: the chains run on host cores, from the modelers' records (host_blocks.h)
        const int rc = rc_blocks_on_host(c, c->syms.as<uint8_t>(), c->blk_begin.as<uint64_t>(), nbl, n_syms, SMALL_SIZES_DNA, N_SMALL_MODELS);
        if (rc == 1) host_chains = false;                        // (no room for the records: the device's coder)
        else if (rc) return rc;
    }
    if (host_chains) {
        HIPCHK(c, hipEventRecord(c->ev[7], s));
        for (uint64_t b = 0; b < nbl; b++) { sizes[b] = c->hb_coders[b].size(); dst[b + 1] = dst[b] + sizes[b]; }
        payload_bytes = dst[nbl];
        HIPCHK(c, hipEventRecord(c->ev[8], s));
        HIPCHK(c, hipStreamSynchronize(s));
    } else {
    const uint64_t rc_cap = 3 * n_syms + 72 * (nbl + 1);
    HIPCHK(c, c->rc_out.ensure(rc_cap));
    HIPCHK(c, c->rc_scratch.ensure(rc_model_scratch_bytes(nbl)));
    HIPCHK(c, hipMemsetAsync(c->errflag.p, 0, 4, s));
    launch_rc_encode(s, c->syms.as<uint8_t>(), c->blk_begin.as<uint64_t>(), nbl, c->rc_out.as<uint8_t>(),
                     c->out_off.as<uint64_t>(), c->out_size.as<uint64_t>(), c->rc_scratch.as<uint32_t>(), c->errflag.as<int>(), max_block_syms,
                     SMALL_SIZES_DNA, true);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev[7], s));

    // ---- gather + D2H ----
    int errflag = 0;
    HIPCHK(c, hipMemcpyAsync(sizes.data(), c->out_size.p, nbl * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&errflag, c->errflag.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (errflag == 3) return fail(c, LEON_E_STATE, "a numeric value's byte count above 8 in the symbol stream (internal error)");
    if (errflag) return fail(c, LEON_E_OVERFLOW, errflag == 2 ? "a read block has 2^32 symbols or more"
                                                              : "range coder output exceeded its 3 bytes/symbol bound");
    for (uint64_t b = 0; b < nbl; b++) dst[b + 1] = dst[b] + sizes[b];
    payload_bytes = dst[nbl];
    HIPCHK(c, c->payload.ensure(payload_bytes + 16));
    HIPCHK(c, hipMemcpyAsync(c->dst_off.p, dst.data(), (nbl + 1) * 8, hipMemcpyHostToDevice, s));
    launch_gather_payload(s, c->rc_out.as<uint8_t>(), c->out_off.as<uint64_t>(), c->dst_off.as<uint64_t>(), c->out_size.as<uint64_t>(),
                          nbl, c->payload.as<uint8_t>());
    if (payload_bytes + 16 > c->h_payload_cap) {
        if (c->h_payload) HIPCHK(c, hipHostFree(c->h_payload));
        c->h_payload = nullptr; c->h_payload_cap = 0;
        size_t want = payload_bytes + payload_bytes / 4 + 4096;
        HIPCHK(c, hipHostMalloc(&c->h_payload, want, hipHostMallocDefault));
        c->h_payload_cap = want;
    }
    HIPCHK(c, hipMemcpyAsync(c->h_payload, c->payload.p, payload_bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipEventRecord(c->ev[8], s));
    HIPCHK(c, hipStreamSynchronize(s));
    }

    // ---- stats ----
    c->stats.n_symbols = n_syms; c->stats.payload_bytes = payload_bytes;
    float pack2 = 0;                                                   // the part of the pack stage that ran inside the resolution loop
    for (uint32_t e = 1; e < n_pack_ev; e++) { float v = 0; (void)hipEventElapsedTime(&v, c->pack_ev[2 * e], c->pack_ev[2 * e + 1]); pack2 += v; }
    c->stats.ms_pack = ms(0, 1) + pack2; c->stats.ms_resolve = ms(1, 2) - pack2; c->stats.ms_resolve_chain = chain_ms(); c->stats.ms_sort = ms(2, 3); c->stats.ms_walk = ms(4, 5);
    c->stats.ms_symbols = by_anchor ? ms(9, 6) : ms(5, 6); c->stats.ms_rangecoder = ms(6, 7); c->stats.ms_d2h = ms(7, 8); c->stats.ms_total = ms(0, 8);

    // ---- Leon::writeBlock, in block order ----
    const uint8_t* hp = (const uint8_t*)c->h_payload;
    for (uint64_t b = 0; b < nbl; b++) {
        uint32_t nr = (uint32_t)std::min<uint64_t>(rpb, n - (lb0 + b) * rpb);
        const uint8_t* pay = host_chains ? c->hb_coders[b].data() : hp + dst[b];
        if (sink(user, c->next_block + lb0 + b, pay, sizes[b], nr)) return fail(c, LEON_E_SINK, "block sink returned non-zero");
    }
    c->next_read += n;
    c->next_block += n_blocks;
    if (n % rpb) c->partial_seen = true;
    c->poisoned = false;
    (void)k;
    return LEON_OK;
}

int leon_dna_encode_batch(leon_dna_ctx* c, const uint8_t* bases, const uint64_t* off, uint64_t n, uint64_t first_read_index,
                          leon_block_sink sink, void* user) {
    if (!c) return LEON_E_INVALID;
    if (n == 0) return leon_dna_encode_batch_device(c, nullptr, nullptr, 0, first_read_index, sink, user);
    if (!bases || !off) return fail(c, LEON_E_INVALID, "null argument");
    if (off[n] < off[0]) return fail(c, LEON_E_INVALID, "offsets are not monotonic");
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t nb = off[n] - off[0];
    HIPCHK(c, c->in_bases.ensure(nb + 64));
    HIPCHK(c, c->in_off.ensure((n + 1) * 8));
    // offsets first (rebased on the device so that they index the device copy of the bases), then the bases group by
    // group on an uploader thread while the device already works on the groups that have arrived
    HIPCHK(c, staged_h2d(c->device, c->in_off.p, off, (n + 1) * 8));      // (800 MB at 100 M reads: 73 ms as one hipMemcpy from pageable memory, 15 staged)
    if (off[0]) {
        launch_rebase_offsets(c->stream, c->in_off.as<uint64_t>(), n + 1, off[0]);
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    Upload up(nb);
    const int dev = c->device;
    uint8_t* dst = c->in_bases.as<uint8_t>();
    const uint8_t* src = bases + off[0];
    const uint32_t n_workers = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(stage_threads(), up.piece_done.size()));
    for (uint32_t w = 0; w < n_workers; w++)
        up.th.emplace_back([&up, dev, dst, src, nb] {
            StageLane L(dev);
            uint64_t held[2] = {~0ull, ~0ull};                  // the piece a buffer's copy in flight belongs to
            bool ok = L.ok;
            for (uint32_t turn = 0; ok && !up.cancel.load(); turn ^= 1) {
                if (held[turn] != ~0ull) {                     // the buffer's previous piece has landed: publish it before refilling
                    if (hipEventSynchronize(L.ev[turn]) != hipSuccess) { ok = false; break; }
                    up.publish(held[turn]); held[turn] = ~0ull;
                }
                const uint64_t piece = up.next_piece.fetch_add(1);
                if (piece >= up.piece_done.size()) break;
                const uint64_t a = piece * Upload::kPiece, m = std::min<uint64_t>(Upload::kPiece, nb - a);
                memcpy(L.buf[turn], src + a, m);
                if (hipMemcpyAsync(dst + a, L.buf[turn], m, hipMemcpyHostToDevice, L.st) != hipSuccess || hipEventRecord(L.ev[turn], L.st) != hipSuccess) { ok = false; break; }
                held[turn] = piece;
            }
            for (int t = 0; t < 2 && ok; t++)
                if (held[t] != ~0ull) { if (hipEventSynchronize(L.ev[t]) != hipSuccess) ok = false; else up.publish(held[t]); }
            if (!ok && !up.cancel.load()) up.failed.store(1);
        });
    static const bool trace_up = getenv("LEON_TRACE_UPLOAD") != nullptr;      // measurement aid: when the last base reached HBM
    const auto t_up0 = std::chrono::steady_clock::now();
    std::thread watcher;
    if (trace_up) watcher = std::thread([&up, nb, t_up0] {
        while (up.bytes_done.load() < nb && !up.failed.load() && !up.cancel.load()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up0).count();
        fprintf(stderr, "[leon upload] %.1f MB in %.1f ms = %.1f GB/s\n", nb / 1e6, ms, nb / 1e6 / ms);
    });
    const int rc = encode_batch_guarded(c, dst, c->in_off.as<uint64_t>(), n, first_read_index, sink, user, &up, off);
    if (rc != LEON_OK) up.cancel.store(1);
    for (auto& t : up.th) t.join();
    if (watcher.joinable()) watcher.join();
    return rc;
}


// Sizes every per-batch device buffer for a batch of up to max_reads reads / max_bases bases before the first batch arrives
// (a host calls it while it is still parsing): the first leon_dna_encode_batch of a process then allocates nothing large, and
// does not wait for the driver to wipe memory another stage has just returned.  Symbol counts are an
// estimate (40 per read); whatever turns out larger is grown on demand as before.
int leon_dna_reserve(leon_dna_ctx* c, uint64_t max_reads, uint64_t max_bases) {
    if (!c) return LEON_E_INVALID;
    if (max_reads == 0) return LEON_OK;
    if (max_reads > 0xFFFFFFF0ull) return fail(c, LEON_E_INVALID, "more than 2^32 reads in one batch");
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t n = max_reads, rpb = c->cfg.reads_per_block, nbl = (n + rpb - 1) / rpb;
    const uint64_t n_slots = max_bases / 32 + n, W = std::min<uint64_t>(c->cfg.resolve_window, n), est_syms = 40 * n;
    HIPCHK(c, c->slot_off.ensure((n + 1) * 8));
    HIPCHK(c, c->packed.ensure((n_slots * 2 + 16) * 4)); HIPCHK(c, c->nmask.ensure((n_slots + 4) * 4));
    HIPCHK(c, c->rlen.ensure(n * 4)); HIPCHK(c, c->ncount.ensure(n * 4));
    HIPCHK(c, c->status.ensure(n));
    HIPCHK(c, c->hit_pos.ensure(n * 4)); HIPCHK(c, c->hit_slot.ensure(n * 4));
    HIPCHK(c, c->cand_pos.ensure(n * 4)); HIPCHK(c, c->cand_slot.ensure(n * 4));
    HIPCHK(c, c->anchor_pos.ensure(n * 4)); HIPCHK(c, c->anchor_addr.ensure(n * 4));
    HIPCHK(c, c->flags.ensure(n)); HIPCHK(c, c->sort_key.ensure(n * 8));
    HIPCHK(c, c->ins_flag.ensure(W * 4)); HIPCHK(c, c->rank.ensure(W * 4));
    HIPCHK(c, c->ulist0.ensure(W * 4)); HIPCHK(c, c->ulist1.ensure(W * 4));
    HIPCHK(c, c->sort_key2.ensure(n * 8)); HIPCHK(c, c->perm.ensure(n * 4)); HIPCHK(c, c->perm2.ensure(n * 4));
    HIPCHK(c, c->events.ensure(max_bases + 16));
    HIPCHK(c, c->prev.ensure(n * 8)); HIPCHK(c, c->sym_off.ensure((n + 1) * 8)); HIPCHK(c, c->nerr.ensure(n * 8));
    HIPCHK(c, c->syms.ensure(est_syms * 2 + 256));
    HIPCHK(c, c->blk_begin.ensure((nbl + 1) * 8)); HIPCHK(c, c->out_off.ensure((nbl + 1) * 8));
    HIPCHK(c, c->out_size.ensure(nbl * 8)); HIPCHK(c, c->dst_off.ensure((nbl + 1) * 8));
    HIPCHK(c, c->rc_out.ensure(3 * est_syms + 72 * (nbl + 1)));
    HIPCHK(c, c->rc_scratch.ensure(rc_model_scratch_bytes(nbl)));
    HIPCHK(c, c->payload.ensure(max_bases / 12 + 4096));
    size_t t1 = 0, t2 = 0, t3 = 0;                               // the scans' and the sort's work space
    HIPCHK(c, prim::ExclusiveSum(nullptr, t1, c->slot_off.as<uint64_t>(), c->slot_off.as<uint64_t>(), n + 1, c->stream));
    HIPCHK(c, prim::SortPairs(nullptr, t2, c->sort_key.as<uint64_t>(), c->sort_key2.as<uint64_t>(), c->perm.as<uint32_t>(), c->perm2.as<uint32_t>(),
                                                 n, 0, 33, c->stream));
    HIPCHK(c, prim::ExclusiveSum(nullptr, t3, c->ins_flag.as<uint32_t>(), c->rank.as<uint32_t>(), W, c->stream));
    if (int rc = ensure_cub(c, std::max(t1, std::max(t2, t3)))) return rc;
    if (max_bases / 12 + 4096 > c->h_payload_cap) {
        if (c->h_payload) HIPCHK(c, hipHostFree(c->h_payload));
        c->h_payload = nullptr; c->h_payload_cap = 0;
        const size_t want = max_bases / 12 + 4096;
        HIPCHK(c, hipHostMalloc(&c->h_payload, want, hipHostMallocDefault));
        c->h_payload_cap = want;
    }
    // the walk's path cache (sized as encode_batch sizes it: a bucket per ~8 solid k-mers, at most 2^27; best effort)
    if (c->B.n_hash == 7 && c->shard_world <= 2) {
        uint64_t want = c->cfg.bloom_tai / 12 / 8, buckets = 1024;
        while (buckets < want && buckets < (1ull << 27)) buckets <<= 1;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && buckets * 64 <= free_b / 8 && c->wcache.ensure(buckets * 64) != hipSuccess) (void)hipGetLastError();
    }
    // the dictionary: one anchor per ~6 reads of a 30x read set; it grows (rehash) if the data want more
    if (int rc = dict_reserve(c, c->n_keys + n / 6 + 1024)) return rc;
    if ((n / 6) * 8 * kmer_words(c->cfg.kmer_size) > c->anchor_kmers.cap && c->n_anchors == 0) HIPCHK(c, c->anchor_kmers.ensure((n / 6) * 8 * kmer_words(c->cfg.kmer_size)));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    leon_device_trim();                                          // (the k-mer counter's parked buffers go back now that this context's exist: kmer_kernels.hip)
    return LEON_OK;
}

// ------------------------------------------------------------------------------------------------ header stream
// Leon::startHeaderCompression: Dispatcher::iterate(bank, HeaderEncoder(this)) [RECALLED].  Records of every header
// against the previous one in parallel (hdr_kernels.hip), then the per-block chains on k_rc_encode with the header
// stream's model set; blocks leave through the sink in increasing id, like the DNA stream's.
static int header_batch_impl(leon_dna_ctx* c, const uint8_t* d_hdr, const uint64_t* d_off, uint64_t n, uint64_t first_read_index,
                             const uint8_t* first_header, uint64_t first_len, leon_block_sink sink, void* user) {
    if (!c) return LEON_E_INVALID;
    if (first_read_index != c->hdr_next_read) return fail(c, LEON_E_STATE, "first_read_index does not continue the header stream");
    if (c->hdr_partial_seen && n) return fail(c, LEON_E_STATE, "a batch with a partial block must be the last one");
    if (n == 0) return LEON_OK;
    if (!d_hdr || !d_off || !sink || (!first_header && first_len)) return fail(c, LEON_E_INVALID, "null argument");
    if (n > 0xFFFFFFF0ull) return fail(c, LEON_E_INVALID, "more than 2^32 reads in one batch");
    if (first_len >= (1ull << 31)) return fail(c, LEON_E_INVALID, "first header longer than 2^31 bytes");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const uint32_t rpb = c->cfg.reads_per_block;
    const uint64_t n_blocks = (n + rpb - 1) / rpb;
    // offsets must be an offsets array before anything indexes with them (headers below 2^31 bytes each)
    HIPCHK(c, c->slot_off.ensure((n + 1) * 8));
    HIPCHK(c, hipMemsetAsync(c->counters.as<uint32_t>() + 4, 0, 4, s));
    launch_read_slots(s, d_off, n, c->slot_off.as<uint64_t>(), c->counters.as<uint32_t>() + 4);
    uint32_t bad_offsets = 0;
    HIPCHK(c, hipMemcpyAsync(&bad_offsets, c->counters.as<uint32_t>() + 4, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (bad_offsets) return fail(c, LEON_E_INVALID, "offsets are not monotonic (or a header is longer than 2^31 bytes)");
    // this rank's share: the same contiguous block range as the DNA stream's
    uint64_t lb0 = 0, lb1 = n_blocks;
    if (c->shard_world > 1) {
        const uint64_t q = n_blocks / c->shard_world, rm = n_blocks % c->shard_world, rk = c->shard_rank;
        lb0 = rk * q + std::min<uint64_t>(rk, rm);
        lb1 = lb0 + q + (rk < rm ? 1 : 0);
    }
    const uint64_t nbl = lb1 - lb0;
    const uint64_t r0 = std::min<uint64_t>(n, lb0 * rpb), r1 = std::min<uint64_t>(n, lb1 * rpb), nl = r1 - r0;
    if (nl) {
        HIPCHK(c, c->hdr_first.ensure(first_len + 16));
        if (first_len) HIPCHK(c, hipMemcpyAsync(c->hdr_first.p, first_header, first_len, hipMemcpyHostToDevice, s));
        HIPCHK(c, c->sym_off.ensure((nl + 1) * 8));
        HIPCHK(c, hipMemsetAsync(c->sym_off.as<uint64_t>() + nl, 0, 8, s));
        launch_hdr_symbols(s, d_hdr, d_off + r0, nl, rpb, c->hdr_first.as<uint8_t>(), (uint32_t)first_len, c->sym_off.as<uint64_t>(), nullptr);
        size_t tmp_bytes = 0;
        HIPCHK(c, prim::ExclusiveSum(nullptr, tmp_bytes, c->sym_off.as<uint64_t>(), c->sym_off.as<uint64_t>(), nl + 1, s));
        if (int rc = ensure_cub(c, tmp_bytes)) return rc;
        HIPCHK(c, prim::ExclusiveSum(c->cub_tmp.p, tmp_bytes, c->sym_off.as<uint64_t>(), c->sym_off.as<uint64_t>(), nl + 1, s));
        uint64_t n_syms = 0, max_block_syms = 0;
        unsigned long long* d_max = reinterpret_cast<unsigned long long*>(c->counters.as<uint32_t>() + 8);
        HIPCHK(c, hipMemsetAsync(d_max, 0, 8, s));
        launch_max_block_syms(s, c->sym_off.as<uint64_t>(), nl, rpb, nbl, d_max);
        HIPCHK(c, hipMemcpyAsync(&n_syms, c->sym_off.as<uint64_t>() + nl, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(&max_block_syms, d_max, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        HIPCHK(c, c->syms.ensure(n_syms * 2 + 256));
        launch_hdr_symbols(s, d_hdr, d_off + r0, nl, rpb, c->hdr_first.as<uint8_t>(), (uint32_t)first_len, c->sym_off.as<uint64_t>(), c->syms.as<uint8_t>());
        HIPCHK(c, c->blk_begin.ensure((nbl + 1) * 8)); HIPCHK(c, c->out_off.ensure((nbl + 1) * 8));
        HIPCHK(c, c->out_size.ensure(nbl * 8)); HIPCHK(c, c->dst_off.ensure((nbl + 1) * 8));
        launch_block_ranges(s, c->sym_off.as<uint64_t>(), nl, rpb, nbl, c->blk_begin.as<uint64_t>(), c->out_off.as<uint64_t>());
        bool host_chains = rc_on_host(nbl, n_syms, max_block_syms, c->shard_world);
        std::vector<uint64_t> sizes(nbl), dst(nbl + 1, 0);
        if (host_chains) {                                       // a small launch: the blocks' chains on host cores (host_blocks.h)
            const int rc = rc_blocks_on_host(c, c->syms.as<uint8_t>(), c->blk_begin.as<uint64_t>(), nbl, n_syms, SMALL_SIZES_HEADER, N_SMALL_MODELS);
            if (rc == 1) host_chains = false;
            else if (rc) return rc;
        }
        if (host_chains) {
            for (uint64_t b = 0; b < nbl; b++) { sizes[b] = c->hb_coders[b].size(); dst[b + 1] = dst[b] + sizes[b]; }
        } else {
        HIPCHK(c, c->rc_out.ensure(3 * n_syms + 72 * (nbl + 1)));
        HIPCHK(c, c->rc_scratch.ensure(rc_model_scratch_bytes(nbl)));
        HIPCHK(c, hipMemsetAsync(c->errflag.p, 0, 4, s));
        launch_rc_encode(s, c->syms.as<uint8_t>(), c->blk_begin.as<uint64_t>(), nbl, c->rc_out.as<uint8_t>(), c->out_off.as<uint64_t>(),
                         c->out_size.as<uint64_t>(), c->rc_scratch.as<uint32_t>(), c->errflag.as<int>(), max_block_syms, SMALL_SIZES_HEADER);
        HIPCHK(c, hipGetLastError());
        int errflag = 0;
        HIPCHK(c, hipMemcpyAsync(sizes.data(), c->out_size.p, nbl * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(&errflag, c->errflag.p, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (errflag) return fail(c, LEON_E_OVERFLOW, errflag == 2 ? "a header block has 2^32 symbols or more"
                                                                  : "range coder output exceeded its 3 bytes/symbol bound");
        for (uint64_t b = 0; b < nbl; b++) dst[b + 1] = dst[b] + sizes[b];
        const uint64_t payload_bytes = dst[nbl];
        HIPCHK(c, c->payload.ensure(payload_bytes + 16));
        HIPCHK(c, hipMemcpyAsync(c->dst_off.p, dst.data(), (nbl + 1) * 8, hipMemcpyHostToDevice, s));
        launch_gather_payload(s, c->rc_out.as<uint8_t>(), c->out_off.as<uint64_t>(), c->dst_off.as<uint64_t>(), c->out_size.as<uint64_t>(),
                              nbl, c->payload.as<uint8_t>());
        if (payload_bytes + 16 > c->h_payload_cap) {
            if (c->h_payload) HIPCHK(c, hipHostFree(c->h_payload));
            c->h_payload = nullptr; c->h_payload_cap = 0;
            size_t want = payload_bytes + payload_bytes / 4 + 4096;
            HIPCHK(c, hipHostMalloc(&c->h_payload, want, hipHostMallocDefault));
            c->h_payload_cap = want;
        }
        HIPCHK(c, hipMemcpyAsync(c->h_payload, c->payload.p, payload_bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        }
        const uint8_t* hp = (const uint8_t*)c->h_payload;
        for (uint64_t b = 0; b < nbl; b++) {
            const uint32_t nr = (uint32_t)std::min<uint64_t>(rpb, n - (lb0 + b) * rpb);
            if (sink(user, c->hdr_next_block + lb0 + b, host_chains ? c->hb_coders[b].data() : hp + dst[b], sizes[b], nr)) {
                c->poisoned = true;                             // the caller holds part of the batch's blocks
                return fail(c, LEON_E_SINK, "block sink returned non-zero (stream poisoned: leon_dna_reset_stream to go on)");
            }
        }
    }
    c->hdr_next_read += n;
    c->hdr_next_block += n_blocks;
    if (n % rpb) c->hdr_partial_seen = true;
    return LEON_OK;
}

int leon_header_encode_batch_device(leon_dna_ctx* c, const uint8_t* d_headers, const uint64_t* d_off, uint64_t n, uint64_t first_read_index,
                                    const uint8_t* first_header, uint64_t first_header_len, leon_block_sink sink, void* user) {
    if (!c) return LEON_E_INVALID;
    if (c->poisoned) return fail(c, LEON_E_STATE, "an earlier batch failed part-way: the stream is unusable until leon_dna_reset_stream");
    return header_batch_impl(c, d_headers, d_off, n, first_read_index, first_header, first_header_len, sink, user);
}

int leon_header_encode_batch(leon_dna_ctx* c, const uint8_t* headers, const uint64_t* off, uint64_t n, uint64_t first_read_index,
                             const uint8_t* first_header, uint64_t first_header_len, leon_block_sink sink, void* user) {
    if (!c) return LEON_E_INVALID;
    if (n == 0) return leon_header_encode_batch_device(c, nullptr, nullptr, 0, first_read_index, first_header, first_header_len, sink, user);
    if (!headers || !off) return fail(c, LEON_E_INVALID, "null argument");
    if (off[n] < off[0]) return fail(c, LEON_E_INVALID, "offsets are not monotonic");
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t nb = off[n] - off[0];
    HIPCHK(c, c->in_bases.ensure(nb + 64));
    HIPCHK(c, c->in_off.ensure((n + 1) * 8));
    HIPCHK(c, hipMemcpy(c->in_off.p, off, (n + 1) * 8, hipMemcpyHostToDevice));
    if (off[0]) {
        launch_rebase_offsets(c->stream, c->in_off.as<uint64_t>(), n + 1, off[0]);
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (nb) HIPCHK(c, hipMemcpy(c->in_bases.p, headers + off[0], nb, hipMemcpyHostToDevice));
    return leon_header_encode_batch_device(c, c->in_bases.as<uint8_t>(), c->in_off.as<uint64_t>(), n, first_read_index, first_header,
                                           first_header_len, sink, user);
}

// ------------------------------------------------------------------------------------------------ quality stream (lossy)
// DnaEncoder::smoothQuals over a batch: the reads are packed as for the encode path, then one wave per read rewrites the
// qualities in place (hdr_kernels.hip k_qual_smooth).  Needs the file's bloom in the context.
int leon_qual_smooth_batch_device(leon_dna_ctx* c, const uint8_t* d_bases, const uint64_t* d_off, uint64_t n, uint8_t* d_quals) {
    if (!c) return LEON_E_INVALID;
    if (n == 0) return LEON_OK;
    if (!d_bases || !d_off || !d_quals) return fail(c, LEON_E_INVALID, "null argument");
    if (n > 0xFFFFFFF0ull) return fail(c, LEON_E_INVALID, "more than 2^32 reads in one batch");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    HIPCHK(c, c->slot_off.ensure((n + 1) * 8));
    HIPCHK(c, hipMemsetAsync(c->counters.as<uint32_t>() + 4, 0, 4, s));
    launch_read_slots(s, d_off, n, c->slot_off.as<uint64_t>(), c->counters.as<uint32_t>() + 4);
    size_t tmp_bytes = 0;
    HIPCHK(c, prim::ExclusiveSum(nullptr, tmp_bytes, c->slot_off.as<uint64_t>(), c->slot_off.as<uint64_t>(), n + 1, s));
    if (int rc = ensure_cub(c, tmp_bytes)) return rc;
    HIPCHK(c, prim::ExclusiveSum(c->cub_tmp.p, tmp_bytes, c->slot_off.as<uint64_t>(), c->slot_off.as<uint64_t>(), n + 1, s));
    uint64_t n_slots = 0;
    uint32_t bad_offsets = 0;
    HIPCHK(c, hipMemcpyAsync(&n_slots, c->slot_off.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&bad_offsets, c->counters.as<uint32_t>() + 4, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (bad_offsets) return fail(c, LEON_E_INVALID, "offsets are not monotonic (or a read is longer than 2^31 bases)");
    HIPCHK(c, c->packed.ensure((n_slots * 2 + 16) * 4));
    HIPCHK(c, c->nmask.ensure((n_slots + 4) * 4));
    HIPCHK(c, c->rlen.ensure(n * 4));
    HIPCHK(c, c->ncount.ensure(n * 4));
    HIPCHK(c, hipMemsetAsync(c->packed.as<uint32_t>() + n_slots * 2, 0, 64, s));
    launch_pack(s, d_bases, d_off, c->slot_off.as<uint64_t>(), n, c->packed.as<uint32_t>(), c->nmask.as<uint32_t>(), c->rlen.as<uint32_t>(),
                c->ncount.as<uint32_t>());
    // large batches: the probes shared between the reads of a locus (hdr_kernels.hip) -- a minimizer per read, one sort, one lane per
    // read in that order, then the rewrite; small ones (and LEON_QUAL_ORDER=0, the measurement reference): every read for itself
    const char* qo = getenv("LEON_QUAL_ORDER");
    const bool in_file_order = qo && atoi(qo) == 0;
    const ReadsDev R = reads_view(c, d_off, n);
    if (!in_file_order && n >= 256) {
        HIPCHK(c, c->sort_key.ensure(n * 4)); HIPCHK(c, c->sort_key2.ensure(n * 4));
        HIPCHK(c, c->perm.ensure(n * 4)); HIPCHK(c, c->perm2.ensure(n * 4));
        HIPCHK(c, c->hit_pos.ensure(n * 4));                    // (the minimizers' positions; the encode's per-read arrays are free between batches)
        HIPCHK(c, c->events.ensure((n_slots + 4) * 4));         // (the flags)
        HIPCHK(c, hipMemsetAsync(c->events.p, 0, (n_slots + 4) * 4, s));
        // the reads of the batch this context encoded last (the CLI smooths a file right after its DNA stream): their ANCHORS are
        // at hand and serve better than minimizers -- a read with a sequencing error in its minimizer lands in a cluster of its
        // own, one with an error in an anchor k-mer simply has another anchor.  (Which k-mer a read is aligned on only decides
        // who shares probes with whom, never a flag: stale anchors would cost time, not bytes; positions are range-checked.)
        const bool anchors = n == c->last_n && d_bases == c->last_d_bases && d_off == c->last_d_off && c->shard_world == 1 && !c->poisoned;
        if (anchors) launch_mpos_from_anchors(s, R, c->anchor_pos.as<int32_t>(), c->anchor_addr.as<uint32_t>(), c->flags.as<uint8_t>(),
                                              c->sort_key.as<uint32_t>(), c->hit_pos.as<uint32_t>());
        else launch_read_minimizer(s, R, c->sort_key.as<uint32_t>(), c->hit_pos.as<uint32_t>());
        hipLaunchKernelGGL(k_iota, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 8192)), dim3(256), 0, s, c->perm.as<uint32_t>(), n);
        size_t sort_tmp = 0;
        HIPCHK(c, prim::SortPairs(nullptr, sort_tmp, c->sort_key.as<uint32_t>(), c->sort_key2.as<uint32_t>(), c->perm.as<uint32_t>(), c->perm2.as<uint32_t>(), n, 0, 32, s));
        if (int rc = ensure_cub(c, sort_tmp)) return rc;
        HIPCHK(c, prim::SortPairs(c->cub_tmp.p, sort_tmp, c->sort_key.as<uint32_t>(), c->sort_key2.as<uint32_t>(), c->perm.as<uint32_t>(), c->perm2.as<uint32_t>(), n, 0, 32, s));
        launch_solid_flags(s, R, c->B, c->d_rv16, c->perm2.as<uint32_t>(), c->hit_pos.as<uint32_t>(), c->events.as<uint32_t>());
        launch_qual_rewrite(s, R, c->events.as<uint32_t>(), d_quals);
    } else launch_qual_smooth(s, R, c->B, c->d_rv16, d_quals);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));
    return LEON_OK;
}

int leon_qual_smooth_batch(leon_dna_ctx* c, const uint8_t* bases, const uint64_t* off, uint64_t n, uint8_t* quals) {
    if (!c) return LEON_E_INVALID;
    if (n == 0) return LEON_OK;
    if (!bases || !off || !quals) return fail(c, LEON_E_INVALID, "null argument");
    if (off[n] < off[0]) return fail(c, LEON_E_INVALID, "offsets are not monotonic");
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t nb = off[n] - off[0];
    TmpBuf dq;
    HIPCHK(c, c->in_bases.ensure(nb + 64));
    HIPCHK(c, c->in_off.ensure((n + 1) * 8));
    HIPCHK(c, dq.ensure(nb + 64));
    HIPCHK(c, hipMemcpy(c->in_off.p, off, (n + 1) * 8, hipMemcpyHostToDevice));
    if (off[0]) {
        launch_rebase_offsets(c->stream, c->in_off.as<uint64_t>(), n + 1, off[0]);
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (nb) {
        HIPCHK(c, hipMemcpy(c->in_bases.p, bases + off[0], nb, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(dq.p, quals + off[0], nb, hipMemcpyHostToDevice));
    }
    if (int rc = leon_qual_smooth_batch_device(c, c->in_bases.as<uint8_t>(), c->in_off.as<uint64_t>(), n, dq.as<uint8_t>())) return rc;
    if (nb) HIPCHK(c, hipMemcpy(quals + off[0], dq.p, nb, hipMemcpyDeviceToHost));
    return LEON_OK;
}

int leon_dna_finish(leon_dna_ctx* c, const uint8_t** payload, uint64_t* size, uint64_t* n_anchors) {
    if (!c || !payload || !size || !n_anchors) return LEON_E_INVALID;
    if (c->poisoned) return fail(c, LEON_E_STATE, "an earlier batch failed part-way: the stream is unusable until leon_dna_reset_stream");
    auto t0 = std::chrono::steady_clock::now();
    if (c->cfg.flags & LEON_F_DICT_ON_DEVICE) {
        if (!c->finished) {
            c->dict_device_out.clear();
            if (c->shard_rank == 0) {
                HIPCHK(c, hipSetDevice(c->device));
                hipStream_t s = c->stream;
                const uint64_t nsym = c->n_anchors * c->cfg.kmer_size;
                TmpBuf dsyms, dbegin, doff, dsize, dout, dscr;
                HIPCHK(c, dsyms.ensure(nsym * 2 + 256)); HIPCHK(c, dbegin.ensure(16)); HIPCHK(c, doff.ensure(16)); HIPCHK(c, dsize.ensure(8));
                const uint64_t begin[2] = { 0, nsym }, off[2] = { 0, ((3 * nsym + 7) & ~7ull) + 64 };
                HIPCHK(c, dout.ensure(off[1] + 64)); HIPCHK(c, dscr.ensure(rc_model_scratch_bytes(1)));
                HIPCHK(c, hipMemcpy(dbegin.p, begin, 16, hipMemcpyHostToDevice));
                HIPCHK(c, hipMemcpy(doff.p, off, 16, hipMemcpyHostToDevice));
                launch_anchor_symbols(s, c->anchor_kmers.as<uint64_t>(), c->n_anchors, c->cfg.kmer_size, dsyms.as<uint8_t>());
                HIPCHK(c, hipMemsetAsync(c->errflag.p, 0, 4, s));
                launch_rc_encode(s, dsyms.as<uint8_t>(), dbegin.as<uint64_t>(), 1, dout.as<uint8_t>(), doff.as<uint64_t>(), dsize.as<uint64_t>(),
                                 dscr.as<uint32_t>(), c->errflag.as<int>(), nsym);
                HIPCHK(c, hipGetLastError());
                HIPCHK(c, hipStreamSynchronize(s));
                uint64_t sz = 0; int errflag = 0;
                HIPCHK(c, hipMemcpy(&sz, dsize.p, 8, hipMemcpyDeviceToHost));
                HIPCHK(c, hipMemcpy(&errflag, c->errflag.p, 4, hipMemcpyDeviceToHost));
                if (errflag) return fail(c, LEON_E_OVERFLOW, "dictionary stream: device range coder bound hit");
                c->dict_device_out.resize(sz);
                if (sz) HIPCHK(c, hipMemcpy(c->dict_device_out.data(), dout.p, sz, hipMemcpyDeviceToHost));
            }
            c->finished = true;
        }
        c->anchor_wait_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        *payload = c->dict_device_out.data();
        *size = c->dict_device_out.size();
        *n_anchors = c->n_anchors;
        return LEON_OK;
    }
    c->anchor_worker->drain();
    c->anchor_wait_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("LEON_TRACE_STEP")) fprintf(stderr, "[leon step] dictionary chain busy %.1f ms, of which %.1f ms waiting for its helpers' records\n",
                                           c->anchor_worker->busy_ms(), c->anchor_worker->starved_ms());
    if (!c->finished) {
        if (c->shard_rank == 0) c->anchor_worker->coder().flush();
        c->finished = true;
    }
    *payload = c->anchor_worker->coder().data();
    *size = c->shard_rank == 0 ? c->anchor_worker->coder().size() : 0;      // the dictionary stream is rank 0's to write
    *n_anchors = c->n_anchors;
    return LEON_OK;
}

int leon_dna_debug_walk_order(leon_dna_ctx* c, const uint64_t* d_keys) {
    if (!c) return LEON_E_INVALID;
    c->walk_keys = d_keys;
    return LEON_OK;
}

int leon_dna_set_shard(leon_dna_ctx* c, uint32_t rank, uint32_t world) {
    if (!c) return LEON_E_INVALID;
    if (world == 0 || rank >= world) return fail(c, LEON_E_INVALID, "set_shard: need rank < world");
    if (c->next_read) return fail(c, LEON_E_STATE, "set_shard must precede the first batch of a stream");
    c->shard_rank = rank; c->shard_world = world;
    return LEON_OK;
}

int leon_dna_set_exchange(leon_dna_ctx* c, uint32_t mode, leon_exchange_fn fn, void* user) {
    if (!c) return LEON_E_INVALID;
    if (mode > LEON_XCH_EMULATE) return fail(c, LEON_E_INVALID, "set_exchange: unknown mode");
    if (mode == LEON_XCH_BY_ANCHOR && !fn) return fail(c, LEON_E_INVALID, "set_exchange: LEON_XCH_BY_ANCHOR needs the exchange callback");
    if (c->next_read) return fail(c, LEON_E_STATE, "set_exchange must precede the first batch of a stream");
    c->xch_mode = mode; c->xch_fn = fn; c->xch_user = user;
    return LEON_OK;
}

int leon_dna_set_gather(leon_dna_ctx* c, leon_gather_fn fn, void* user) {
    if (!c) return LEON_E_INVALID;
    if (c->next_read) return fail(c, LEON_E_STATE, "set_gather must precede the first batch of a stream");
    c->gather_fn = fn; c->gather_user = user;
    return LEON_OK;
}

int leon_dna_reset_stream(leon_dna_ctx* c) {
    if (!c) return LEON_E_INVALID;
    static const bool trace_step = getenv("LEON_TRACE_STEP") != nullptr;
    const auto t_enter = std::chrono::steady_clock::now();
    struct Done { bool on; std::chrono::steady_clock::time_point t0; ~Done() { if (on) fprintf(stderr, "[leon step] %-34s %8.2f ms\n", "reset_stream", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); } } done{trace_step, t_enter};
    c->anchor_worker->reset();
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, spin_sync(c->stream));
    HIPCHK(c, hipMemsetAsync(c->fbits.p, 0, (1ull << c->fbits_log2) / 8, c->stream));
    if (c->dict_cap) {
        HIPCHK(c, hipMemsetAsync(c->d_nkeys, 0, 8, c->stream));
        launch_dict_init(c->stream, c->D, c->dict_cap, kmer_words(c->cfg.kmer_size));
        HIPCHK(c, spin_sync(c->stream));
    }
    c->n_keys = 0; c->n_anchors = 0;
    c->next_read = 0; c->next_block = 0; c->partial_seen = false; c->finished = false; c->poisoned = false;
    c->hdr_next_read = 0; c->hdr_next_block = 0; c->hdr_partial_seen = false;
    c->last_n = 0; c->last_bases = 0; c->last_d_bases = c->last_d_off = nullptr;
    return LEON_OK;
}

int leon_dna_get_stats(const leon_dna_ctx* c, leon_dna_stats* out) {
    if (!c || !out) return LEON_E_INVALID;
    *out = c->stats;
    out->n_anchors = c->n_anchors;
    out->ms_anchor_wait = (float)c->anchor_wait_ms;
    out->ms_chain_busy = c->finished ? (float)c->anchor_worker->busy_ms() : 0.f;
    return LEON_OK;
}

// ------------------------------------------------------------------------------------------------ decode
int leon_host_anchor_dict_decode(const uint8_t* payload, uint64_t size, uint64_t n_anchors, uint32_t k, uint64_t* out) {
    if ((!payload && size) || (!out && n_anchors)) return fail(nullptr, LEON_E_INVALID, "null argument");
    if (k < 3 || k > 63) return fail(nullptr, LEON_E_INVALID, "kmer_size must be in 3..63");
    if (!decode_anchor_dict(payload, size, n_anchors, k, out)) return fail(nullptr, LEON_E_INVALID, "anchor dictionary stream does not decode");
    return LEON_OK;
}

int leon_dna_decode_blocks(leon_dna_ctx* c, const uint64_t* anchors, uint64_t n_anchors, const uint8_t* payloads,
                           const uint64_t* payload_off, const uint32_t* block_n_reads, const uint64_t* block_n_bases,
                           uint64_t n_blocks, uint8_t* out_bases, uint64_t out_cap, uint32_t* out_len) {
    if (!c) return LEON_E_INVALID;
    if (n_blocks == 0) return LEON_OK;
    if (!payloads || !payload_off || !block_n_reads || !block_n_bases || !out_bases || !out_len || (!anchors && n_anchors))
        return fail(c, LEON_E_INVALID, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    static const bool trace = getenv("LEON_TRACE_DECODE") != nullptr;    // where a call's time goes, on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[leon decode] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    const uint32_t W = kmer_words(c->cfg.kmer_size);
    std::vector<uint64_t> read0(n_blocks + 1, 0), out0(n_blocks + 1, 0);
    for (uint64_t b = 0; b < n_blocks; b++) {
        if (payload_off[b + 1] < payload_off[b]) return fail(c, LEON_E_INVALID, "payload offsets are not monotonic");
        read0[b + 1] = read0[b] + block_n_reads[b];
        out0[b + 1] = out0[b] + block_n_bases[b];
        if (out0[b + 1] < out0[b] || out0[b + 1] > out_cap) return fail(c, LEON_E_INVALID, "output capacity below the sum of block_n_bases");
    }
    if (out0[n_blocks] > out_cap) return fail(c, LEON_E_INVALID, "output capacity below the sum of block_n_bases");
    const uint64_t pay_bytes = payload_off[n_blocks] - payload_off[0];
    TmpBuf d_anchors, d_off, d_nreads, d_read0, d_out0, d_err;
    DevBuf &d_pay = c->dc_pay, &d_out = c->dc_out, &d_len = c->dc_len, &d_scr = c->dc_scr, &d_pool = c->dc_pool;
    HIPCHK(c, d_anchors.ensure(std::max<uint64_t>(n_anchors * W, 1) * 8));
    HIPCHK(c, d_pay.ensure(pay_bytes + 1024));                 // the payload window reads up to 256 + 3 bytes past a block's end
    HIPCHK(c, d_off.ensure((n_blocks + 1) * 8)); HIPCHK(c, d_nreads.ensure(n_blocks * 4));
    HIPCHK(c, d_read0.ensure((n_blocks + 1) * 8)); HIPCHK(c, d_out0.ensure((n_blocks + 1) * 8));
    HIPCHK(c, d_out.ensure(out0[n_blocks] + 64)); HIPCHK(c, d_len.ensure(std::max<uint64_t>(read0[n_blocks], 1) * 4));
    HIPCHK(c, d_scr.ensure(decode_scratch_bytes(n_blocks))); HIPCHK(c, d_err.ensure(256));
    // position lists longer than a block's own scratch (8192 N or error positions in ONE read) come from this pool
    const uint64_t pool_words = std::min<uint64_t>(std::max<uint64_t>(out0[n_blocks] / 2, 1ull << 20), 1ull << 28);
    HIPCHK(c, d_pool.ensure(pool_words * 4 + 16));
    lap("buffers");
    bool cache_is_new = false;
    // The path cache: a few bytes per solid k-mer, kept from call to call for as long as the bloom's bits stay what they
    // were (their fingerprint is taken again at every call: 0.3 ms for a gigabyte).  4 slots per k-mer the bloom was sized
    // for (two orientations, half full), at most 40 % of the free memory; LEON_DC_CACHE_MB overrides (0: no cache).
    {
        uint64_t want_buckets = std::max<uint64_t>(c->cfg.bloom_tai / 12 * (W == 2 ? 2 : 1), 1024);
        size_t free_b = 0, total_b = 0;
        HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
        uint64_t budget = (uint64_t)(free_b + c->dc_cache.cap) * 2 / 5;
        if (const char* e = getenv("LEON_DC_CACHE_MB")) budget = (uint64_t)std::max<long long>(0, atoll(e)) << 20;
        uint64_t buckets = 1024;
        while (buckets < want_buckets) buckets <<= 1;
        while (buckets > 1024 && buckets * 64 > budget) buckets >>= 1;
        if (buckets * 64 > budget) buckets = 0;
        uint64_t* d_fp = d_err.as<uint64_t>() + 1;
        HIPCHK(c, hipMemsetAsync(d_fp, 0, 8, s));
        launch_bloom_fingerprint(s, c->d_bloom, c->bloom_nchar, d_fp);
        uint64_t fp = 0;
        HIPCHK(c, hipMemcpyAsync(&fp, d_fp, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (!buckets) { c->dc_cache.release(); c->dc_pc = PathCache{}; c->dc_filled = false; }
        else if (!c->dc_pc.slots || c->dc_pc.bucket_mask != buckets - 1 || !c->dc_filled || fp != c->dc_bloom_fp) {
            if (c->dc_pc.bucket_mask != buckets - 1 || !c->dc_pc.slots) {
                c->dc_cache.release();
                if (c->dc_cache.ensure(buckets * 64) != hipSuccess) { (void)hipGetLastError(); c->dc_pc = PathCache{}; buckets = 0; }   // no room: decode without it
                else { c->dc_pc.slots = c->dc_cache.as<uint64_t>(); c->dc_pc.bucket_mask = buckets - 1; }
            }
            if (buckets) { launch_path_cache_init(s, c->dc_pc, c->cfg.kmer_size); c->dc_bloom_fp = fp; c->dc_filled = true; cache_is_new = true; }
        }
    }
    lap("path cache");
    std::vector<uint64_t> rel_off(n_blocks + 1);
    for (uint64_t b = 0; b <= n_blocks; b++) rel_off[b] = payload_off[b] - payload_off[0];
    if (n_anchors) HIPCHK(c, hipMemcpyAsync(d_anchors.p, anchors, n_anchors * W * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, staged_h2d(c->device, d_pay.p, payloads + payload_off[0], pay_bytes));
    HIPCHK(c, hipMemsetAsync((uint8_t*)d_pay.p + pay_bytes, 0, 1024, s));
    HIPCHK(c, hipMemcpyAsync(d_off.p, rel_off.data(), (n_blocks + 1) * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_nreads.p, block_n_reads, n_blocks * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_read0.p, read0.data(), (n_blocks + 1) * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_out0.p, out0.data(), (n_blocks + 1) * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemsetAsync(d_err.p, 0, 256, s));
    HIPCHK(c, hipMemsetAsync((uint8_t*)d_pool.p + pool_words * 4, 0, 16, s));           // the pool's cursor lives behind it
    lap("payloads to the device");
    if (cache_is_new) {                                       // what the bloom says around every anchor, before the blocks ask (decode_kernels.hip)
        static const char* pw = getenv("LEON_DC_PREWALK");    // measurement override: steps per anchor and orientation, 0 = none
        // probes per anchor and orientation.  The walks are throughput (0.25 s for 25 M lanes x 64), the blocks' chains latency:
        // 128 pays for files of a few hundred blocks (10 M reads: 26 + 1 697 ms -> 52 + 1 593) and for two-word k-mers, 64 at
        // 2 000 blocks (249 + 2 215 ms against 534 + 2 065)
        const uint32_t steps = pw ? (uint32_t)std::max(0, atoi(pw)) : ((c->cfg.kmer_size >= 32 || n_anchors < (4u << 20)) ? 128u : 64u);
        if (steps) launch_path_cache_prewalk(s, c->B, c->dc_pc, c->d_rv16, d_anchors.as<uint64_t>(), n_anchors, steps);
        lap("path cache: walks from the anchors");
    }
    launch_decode_blocks(s, c->B, c->dc_pc, c->d_rv16, d_anchors.as<uint64_t>(), n_anchors, d_pay.as<uint8_t>(), d_off.as<uint64_t>(),
                         d_nreads.as<uint32_t>(), d_read0.as<uint64_t>(), d_out0.as<uint64_t>(), n_blocks, d_out.as<uint8_t>(),
                         d_len.as<uint32_t>(), d_scr.as<uint32_t>(), d_pool.as<uint32_t>(),
                         (unsigned long long*)((uint8_t*)d_pool.p + pool_words * 4), pool_words, d_err.as<int>(),
                         trace ? (unsigned long long*)((uint8_t*)d_err.p + 64) : nullptr);
    HIPCHK(c, hipGetLastError());
    // While the blocks decode (seconds), the caller's output buffer is touched page by page: fresh memory is mapped on first
    // write, which would otherwise happen inside the copy back -- 3.7 M page faults for a 100 M-read file, on the critical path.
    // Each page's first byte is written back as it was read, so the buffer's contents survive a call that then fails (a corrupt
    // block, a HIP error): a successful call overwrites all of it, a failed one changes nothing.
    std::vector<std::thread> toucher;
    {
        const uint64_t nb_out = out0[n_blocks];
        const uint32_t nt = nb_out >= (256ull << 20) ? 4u : 0u;
        for (uint32_t w = 0; w < nt; w++)
            toucher.emplace_back([out_bases, nb_out, w, nt] {
                volatile uint8_t* q = out_bases;
                for (uint64_t a = nb_out * w / nt; a < nb_out * (w + 1) / nt; a += 4096) q[a] = q[a];
            });
    }
    struct Join { std::vector<std::thread>& t; ~Join() { for (auto& x : t) if (x.joinable()) x.join(); } } join_touchers{toucher};
    int err[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(err, d_err.p, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    for (auto& x : toucher) x.join();
    toucher.clear();
    lap("k_decode_blocks");
    if (trace) {
        unsigned long long st[16] = {0};
        HIPCHK(c, hipMemcpy(st, (uint8_t*)d_err.p + 64, 128, hipMemcpyDeviceToHost));
        const double nr = st[5] ? (double)st[5] : 1.0;
        fprintf(stderr, "[leon decode] per read: %.2f table jumps (%.1f positions), %.2f table misses, %.2f one-k-mer probe rounds, %.2f deep probe rounds, %.2f more answered by the table\n",
                st[0] / nr, st[4] / nr, st[1] / nr, st[6] / nr, st[2] / nr, st[3] / nr);
        fprintf(stderr, "[leon decode] per read: %.2f entries taken to their end before the last base, one-k-mer rounds: %.2f on a branching k-mer, %.2f found the successor in the table\n", st[10] / nr, st[11] / nr, st[9] / nr);
        fprintf(stderr, "[leon decode] per read: %.2f us in its fields, %.2f us in its walks (100 MHz clock, the wave's own time)\n", st[7] / nr / 100.0, st[8] / nr / 100.0);
    }
    if (err[0]) {
        const char* what = err[0] == 1 ? "anchor address or position out of range" : err[0] == 2 ? "more or fewer bases than the block table says"
                         : err[0] == 4 ? "the payload ends before its reads do" : "too many N / error positions in one read";
        return fail(c, LEON_E_INVALID, std::string("block ") + std::to_string(err[1]) + " does not decode: " + what);
    }
    HIPCHK(c, staged_d2h(c->device, out_bases, d_out.p, out0[n_blocks]));     // gigabytes into the caller's pageable memory: at PCIe's rate (staging.h)
    if (read0[n_blocks]) HIPCHK(c, staged_d2h(c->device, out_len, d_len.p, read0[n_blocks] * 4));
    lap("bases to the host");
    return LEON_OK;
}

// Header blocks: the symbols on the device (one wave per block, all blocks at once), the text on host threads
extern "C++" { namespace leon {                                // (host_streams.cpp)
int header_blocks_from_symbols(const uint8_t* syms, const uint64_t* sym_begin, const uint64_t* sym_count, const uint32_t* block_n_reads, uint64_t n_blocks,
                               const uint8_t* first_header, uint64_t first_header_len, uint8_t* out, uint64_t out_cap, uint64_t* out_off,
                               uint64_t* out_size, uint32_t n_threads);
} }
struct leon_header_symbols {
    std::unique_ptr<uint8_t[]> syms;                             // every block's symbols, one byte each, block after block
    std::vector<uint64_t> begin, count;                          // block b: syms[begin[b] .. + count[b])
    bool overflowed = false;                                     // some block had more symbols than its share of the device buffer: decode the payloads on the host
};

int leon_header_decode_symbols(leon_dna_ctx* c, const uint8_t* payloads, const uint64_t* payload_off, const uint32_t* block_n_reads, uint64_t n_blocks,
                               leon_header_symbols** set) {
    if (!c) return LEON_E_INVALID;
    if (!set || (n_blocks && (!payloads || !payload_off || !block_n_reads))) return fail(c, LEON_E_INVALID, "null argument");
    *set = nullptr;
    std::unique_ptr<leon_header_symbols> H(new leon_header_symbols());
    H->begin.assign(n_blocks + 1, 0); H->count.assign(n_blocks, 0);
    if (!n_blocks) { *set = H.release(); return LEON_OK; }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // a block's share of the symbol buffer: enough for the headers sequencers write (a dozen symbols each); a block that
    // needs more (free text in every header) marks the whole set for the host decoder -- same result, its speed
    std::vector<uint64_t> rel_off(n_blocks + 1), sym_begin(n_blocks + 1, 0);
    for (uint64_t b = 0; b < n_blocks; b++) {
        if (payload_off[b + 1] < payload_off[b]) return fail(c, LEON_E_INVALID, "payload offsets are not monotonic");
        sym_begin[b + 1] = sym_begin[b] + 24ull * block_n_reads[b] + 6 * (payload_off[b + 1] - payload_off[b]) + 256;   // (SRA-style headers: 12-17 symbols, ~6 bytes each)
    }
    for (uint64_t b = 0; b <= n_blocks; b++) rel_off[b] = payload_off[b] - payload_off[0];
    const uint64_t pay_bytes = rel_off[n_blocks], sym_cap = sym_begin[n_blocks];
    TmpBuf d_pay, d_off, d_nreads, d_begin, d_count, d_syms, d_err;
    HIPCHK(c, d_pay.ensure(pay_bytes + 1024));                  // (the payload window reads up to 256 + 3 bytes past a block's end)
    HIPCHK(c, d_off.ensure((n_blocks + 1) * 8)); HIPCHK(c, d_nreads.ensure(n_blocks * 4));
    HIPCHK(c, d_begin.ensure((n_blocks + 1) * 8)); HIPCHK(c, d_count.ensure(n_blocks * 8));
    HIPCHK(c, d_syms.ensure(sym_cap + 64)); HIPCHK(c, d_err.ensure(16));
    HIPCHK(c, staged_h2d(c->device, d_pay.p, payloads + payload_off[0], pay_bytes));
    HIPCHK(c, hipMemsetAsync((uint8_t*)d_pay.p + pay_bytes, 0, 1024, s));
    HIPCHK(c, hipMemcpyAsync(d_off.p, rel_off.data(), (n_blocks + 1) * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_nreads.p, block_n_reads, n_blocks * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_begin.p, sym_begin.data(), (n_blocks + 1) * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemsetAsync(d_err.p, 0, 16, s));
    launch_hdr_decode_symbols(s, d_pay.as<uint8_t>(), d_off.as<uint64_t>(), d_nreads.as<uint32_t>(), n_blocks, d_syms.as<uint8_t>(),
                              d_begin.as<uint64_t>(), (unsigned long long*)d_count.p, d_err.as<int>());
    HIPCHK(c, hipGetLastError());
    int err[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(err, d_err.p, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (err[0] == 2) return fail(c, LEON_E_INVALID, "header block " + std::to_string(err[1]) + " does not decode");
    if (err[0] == 1) { H->overflowed = true; *set = H.release(); return LEON_OK; }
    HIPCHK(c, hipMemcpy(H->count.data(), d_count.p, n_blocks * 8, hipMemcpyDeviceToHost));
    // only the symbols that were written come back (a block's share is mostly empty)
    for (uint64_t b = 0; b < n_blocks; b++) H->begin[b + 1] = H->begin[b] + H->count[b];
    H->syms.reset(new uint8_t[H->begin[n_blocks] + 1]);
    for (uint64_t b = 0; b < n_blocks; b++)
        if (H->count[b]) HIPCHK(c, hipMemcpyAsync(H->syms.get() + H->begin[b], d_syms.as<uint8_t>() + sym_begin[b], H->count[b], hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    *set = H.release();
    return LEON_OK;
}

int leon_header_text_from_symbols(const leon_header_symbols* H, uint64_t first_block, uint64_t n_blocks, const uint32_t* block_n_reads,
                                  const uint8_t* first_header, uint64_t first_header_len, uint8_t* out, uint64_t out_cap, uint64_t* out_off,
                                  uint64_t* out_size, uint32_t n_threads) {
    if (!H || !out_size || (n_blocks && (!block_n_reads || !out_off)) || (!first_header && first_header_len)) return fail(nullptr, LEON_E_INVALID, "null argument");
    *out_size = 0;
    if (first_block > H->count.size() || n_blocks > H->count.size() - first_block) return fail(nullptr, LEON_E_INVALID, "blocks beyond the symbol set");
    if (H->overflowed) return fail(nullptr, LEON_E_STATE, "the set's symbols did not fit the device buffer: decode the payloads with leon_host_header_decode_blocks");
    if (!n_blocks) return LEON_OK;
    return leon::header_blocks_from_symbols(H->syms.get(), H->begin.data() + first_block, H->count.data() + first_block, block_n_reads, n_blocks, first_header,
                                            first_header_len, out, out_cap, out_off, out_size, n_threads);
}

void leon_header_symbols_free(leon_header_symbols* H) { delete H; }

int leon_header_decode_blocks(leon_dna_ctx* c, const uint8_t* payloads, const uint64_t* payload_off, const uint32_t* block_n_reads, uint64_t n_blocks,
                              const uint8_t* first_header, uint64_t first_header_len, uint8_t* out, uint64_t out_cap, uint64_t* out_off,
                              uint64_t* out_size, uint32_t n_threads) {
    if (!c) return LEON_E_INVALID;
    if (!out_size || (n_blocks && (!payloads || !payload_off || !block_n_reads || !out_off)) || (!first_header && first_header_len))
        return fail(c, LEON_E_INVALID, "null argument");
    *out_size = 0;
    if (!n_blocks) return LEON_OK;
    leon_header_symbols* H = nullptr;
    if (int rc = leon_header_decode_symbols(c, payloads, payload_off, block_n_reads, n_blocks, &H)) return rc;
    std::unique_ptr<leon_header_symbols> own(H);
    int rc;
    if (H->overflowed)                                           // more symbols than a block's share: the host decodes the payloads itself
        rc = leon_host_header_decode_blocks(payloads, payload_off, block_n_reads, n_blocks, first_header, first_header_len, out, out_cap, out_off, out_size, n_threads);
    else
        rc = leon_header_text_from_symbols(H, 0, n_blocks, block_n_reads, first_header, first_header_len, out, out_cap, out_off, out_size, n_threads);
    if (rc != LEON_OK) c->err = leon_last_error(nullptr);
    return rc;
}

// ------------------------------------------------------------------------------------------------ traces
int leon_dna_trace_anchors(leon_dna_ctx* c, int32_t* pos, uint32_t* addr, uint8_t* flags, uint64_t n) {
    if (!c) return LEON_E_INVALID;
    if (n != c->last_n) return fail(c, LEON_E_INVALID, "trace size differs from the last batch");
    HIPCHK(c, hipSetDevice(c->device));
    if (pos) HIPCHK(c, hipMemcpy(pos, c->anchor_pos.p, n * 4, hipMemcpyDeviceToHost));
    if (addr) HIPCHK(c, hipMemcpy(addr, c->anchor_addr.p, n * 4, hipMemcpyDeviceToHost));
    if (flags) HIPCHK(c, hipMemcpy(flags, c->flags.p, n, hipMemcpyDeviceToHost));
    return LEON_OK;
}
int leon_dna_trace_events(leon_dna_ctx* c, uint8_t* events, uint64_t n_bases) {
    if (!c || !events) return LEON_E_INVALID;
    if (n_bases != c->last_bases) return fail(c, LEON_E_INVALID, "trace size differs from the last batch");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(events, c->events.p, n_bases, hipMemcpyDeviceToHost));
    return LEON_OK;
}
int leon_dna_anchor_kmers(leon_dna_ctx* c, uint64_t* kmers, uint64_t n) {
    if (!c || (!kmers && n)) return LEON_E_INVALID;
    if (n > c->n_anchors) return fail(c, LEON_E_INVALID, "more anchors requested than exist");
    HIPCHK(c, hipSetDevice(c->device));
    if (n) HIPCHK(c, hipMemcpy(kmers, c->anchor_kmers.p, n * 8 * kmer_words(c->cfg.kmer_size), hipMemcpyDeviceToHost));
    return LEON_OK;
}

// host-only: the dictionary stream of a list of anchors (what the worker thread produces); no GPU involved
int leon_host_anchor_dict_encode(const uint64_t* kmers, uint64_t n, uint32_t k, uint8_t* out, uint64_t out_cap, uint64_t* size) {
    if ((!kmers && n) || !size || k < 1 || k > 63) return LEON_E_INVALID;
    // the same worker (chain thread + reciprocal helper thread) the contexts use, fed in a few batches
    AnchorDictWorker worker(k);
    const uint32_t W = kmer_words(k);
    const uint64_t step = std::max<uint64_t>(1, n / 3);
    for (uint64_t i = 0; i < n; i += step) {
        const uint64_t m = std::min(step, n - i);
        worker.push(std::vector<uint64_t>(kmers + i * W, kmers + (i + m) * W));
    }
    worker.drain();
    AnchorDictCoder& coder = worker.coder();
    coder.flush();
    *size = coder.size();
    if (coder.size() > out_cap) return LEON_E_OVERFLOW;
    if (out) memcpy(out, coder.data(), coder.size());
    return LEON_OK;
}

// ------------------------------------------------------------------------------------------------ raw range coder
int leon_rc_encode_streams(leon_dna_ctx* c, const uint8_t* syms, const uint64_t* begin, uint64_t n_streams,
                           uint8_t* out, uint64_t out_cap, uint64_t* sizes) {
    if (!c || !begin || !sizes || (!out && out_cap)) return LEON_E_INVALID;
    if (!n_streams) return LEON_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    uint64_t n_syms = begin[n_streams] - begin[0];
    if (begin[0] != 0) return fail(c, LEON_E_INVALID, "begin[0] must be 0");
    for (uint64_t i = 0; i < n_syms; i++) {
        uint32_t m = syms[2 * i], v = syms[2 * i + 1];
        if (m >= N_MODELS || (m < N_SMALL_MODELS && v >= small_model_size(m))) return fail(c, LEON_E_INVALID, "bad symbol");
    }
    TmpBuf dsyms, dbegin, doff, dsize, dout, dscr;
    HIPCHK(c, dsyms.ensure(n_syms * 2 + 256)); HIPCHK(c, dbegin.ensure((n_streams + 1) * 8)); HIPCHK(c, doff.ensure((n_streams + 1) * 8));
    HIPCHK(c, dsize.ensure(n_streams * 8)); HIPCHK(c, dscr.ensure(rc_model_scratch_bytes(n_streams)));
    std::vector<uint64_t> off(n_streams + 1);
    for (uint64_t b = 0; b <= n_streams; b++) off[b] = ((3 * begin[b] + 7) & ~7ull) + 64 * b;
    HIPCHK(c, dout.ensure(off[n_streams] + 64));
    if (n_syms) HIPCHK(c, hipMemcpy(dsyms.p, syms, n_syms * 2, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dbegin.p, begin, (n_streams + 1) * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(doff.p, off.data(), (n_streams + 1) * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemsetAsync(c->errflag.p, 0, 4, s));
    uint64_t longest = 0;
    for (uint64_t b = 0; b < n_streams; b++) longest = std::max(longest, begin[b + 1] - begin[b]);
    // (test hook, LEON_RC_STREAMS_ON_HOST=1: the same streams through the host chains fed by the device's modelers -- host_blocks.h --
    // so that the tests hold the two coders against each other and against the oracle on the same symbols)
    if (const char* e = getenv("LEON_RC_STREAMS_ON_HOST")) if (e[0] == '1' && longest + 1024 < (1ull << HB_COUNT_BITS)) {
        if (int rc = rc_blocks_on_host(c, dsyms.as<uint8_t>(), dbegin.as<uint64_t>(), n_streams, n_syms, SMALL_SIZES_DNA, N_SMALL_MODELS))
            return rc == 1 ? fail(c, LEON_E_HIP, "no memory for the host chains' records") : rc;
        uint64_t w = 0;
        for (uint64_t b = 0; b < n_streams; b++) {
            sizes[b] = c->hb_coders[b].size();
            if (w + sizes[b] > out_cap) return fail(c, LEON_E_OVERFLOW, "out_cap too small");
            memcpy(out + w, c->hb_coders[b].data(), sizes[b]);
            w += sizes[b];
        }
        return LEON_OK;
    }
    launch_rc_encode(s, dsyms.as<uint8_t>(), dbegin.as<uint64_t>(), n_streams, dout.as<uint8_t>(), doff.as<uint64_t>(),
                     dsize.as<uint64_t>(), dscr.as<uint32_t>(), c->errflag.as<int>(), longest);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));
    int errflag = 0;
    HIPCHK(c, hipMemcpy(&errflag, c->errflag.p, 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(sizes, dsize.p, n_streams * 8, hipMemcpyDeviceToHost));
    int rc = LEON_OK;
    if (errflag) rc = fail(c, LEON_E_OVERFLOW, "range coder output exceeded its bound");
    uint64_t w = 0;
    for (uint64_t b = 0; b < n_streams && rc == LEON_OK; b++) {
        if (w + sizes[b] > out_cap) { rc = fail(c, LEON_E_OVERFLOW, "out_cap too small"); break; }
        hipError_t e = hipMemcpy(out + w, dout.as<uint8_t>() + off[b], sizes[b], hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rc = fail(c, LEON_E_HIP, hipGetErrorString(e)); break; }
        w += sizes[b];
    }
    return rc;
}

}  // extern "C"

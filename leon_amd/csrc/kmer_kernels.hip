// kmer_kernels.hip -- solid k-mer counting on the device: the step right before the DNA encode path (upstream: DSK,
// kmer/impl/SortingCountAlgorithm [RECALLED], whose solid k-mers Leon::createBloom inserts).  Sort-based: the canonical
// k-mers of the reads are emitted in hash partitions that fit the device, each partition is radix-sorted (rocPRIM, prim.h), runs
// of at least min_abundance equal k-mers are kept.  k-mers containing an N are skipped, as DSK does.
#include "../../include/leon_dna.h"
#include "kernels.h"

#include "prim.h"
#include "staging.h"

#include <algorithm>
#include <mutex>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace leon {
void set_create_error(const std::string& msg);   // capi.hip: message behind leon_last_error(NULL)

namespace {

// k-mers of partition `part` (of n_parts, by hash), in ONE pass over the reads.  The order of a partition's k-mers does not
// matter -- the sort follows -- so a wave takes the buffer chunk by chunk (PART_CHUNK slots, one atomic add on the cursor per
// chunk) and writes what it selects behind one another inside its chunk; what a chunk has left when the next 64 positions might not
// fit, and at the wave's end, is filled with a key no k-mer has (all ones: a canonical k-mer is never all G, its reverse
// complement all C is smaller), which the sort moves to the end of the partition; *npad counts them.  Round 2 counted per
// read, scanned, then emitted at exact offsets: two passes of ~36 ms per partition at 100 M reads, half of the counter's time.
// (Measured on the way: staging the k-mers in LDS and flushing them in full lines, 64 ms per pass with 256 k-mers per flush
// -- 5.6 M atomic adds on one address go through one L2 channel one after the other -- and 52 ms with 1 024.)
// *cursor may run past cap (nothing is written there): the caller checks.
constexpr uint32_t PART_CHUNK = 1024;
// the partition of a canonical k-mer: every pass asks it of every k-mer of the input, so it is a few 32-bit operations (a
// multiply per word, one mixing round, a multiply-high for the range) rather than the 64-bit mixer and a modulo by a run-time
// divisor, which were most of the pass's instructions (40 -> ... ms per pass at 100 M reads).  Any function of the canonical
// k-mer does; this one only has to spread the k-mers evenly (the buffers have a quarter of slack).
__device__ inline uint32_t part_mix(uint32_t x) { x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 13; return x; }
__device__ inline uint32_t part_of(uint64_t c, uint32_t n_parts) {
    return __umulhi(part_mix((uint32_t)c * 0x9E3779B1u + (uint32_t)(c >> 32) * 0x85EBCA77u), n_parts);
}
__device__ inline uint32_t part_of(u128 c, uint32_t n_parts) {
    const uint64_t lo = (uint64_t)c, hi = (uint64_t)(c >> 64);
    return __umulhi(part_mix((uint32_t)lo * 0x9E3779B1u + (uint32_t)(lo >> 32) * 0x85EBCA77u + (uint32_t)hi * 0xC2B2AE3Du + (uint32_t)(hi >> 32) * 0x27D4EB2Fu), n_parts);
}
template <typename K>
__global__ void __launch_bounds__(256) k_part_kmers(ReadsDev R, uint32_t n_parts, uint32_t part, uint64_t* keys, uint64_t cap, unsigned long long* cursor,
                                                    unsigned long long* npad) {
    const uint32_t lane = lane_id(), k = R.k;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint64_t at = 0, end = 0, pads = 0;                         // wave-uniform: the chunk's next free slot and its end
    auto pad = [&]() {
        for (uint64_t i = at + lane; i < end; i += 64)
            if (i < cap) store_kmer(keys + i * KT<K>::W, ~(K)0);
        pads += end - at;
    };
    // a wave's reads come one after the other, and each needs its length, its slot and then its first dwords: the next read's
    // are asked for while the current one is worked on (two dependent round trips per read otherwise, with nothing beside them)
    uint64_t i = wave;
    uint32_t len = 0, ncnt = 0, words0 = 0;
    uint64_t so = 0;
    if (i < R.n) { len = R.len[i]; so = R.slot_off[i]; ncnt = R.n_count[i]; words0 = pass_words(R.packed + 2 * so, 0, lane); }
    while (i < R.n) {
        const uint64_t inext = i + nwaves;
        uint32_t len_n = 0, ncnt_n = 0;
        uint64_t so_n = 0;
        if (inext < R.n) { len_n = R.len[inext]; so_n = R.slot_off[inext]; ncnt_n = R.n_count[inext]; }
        uint32_t words_n = 0;
        if (len >= k) {
            const uint32_t* pk = R.packed + 2 * so;
            const uint32_t* nm = R.nmask + so;
            const bool hasN = ncnt != 0;
            const uint32_t nk = len - k + 1;
            for (uint32_t base = 0; base < nk; base += 64) {
                const uint32_t words = base == 0 ? words0 : pass_words(pk, base, lane);
                if (base + 64 >= nk && inext < R.n) words_n = pass_words(R.packed + 2 * so_n, 0, lane);   // (the next read's slot is here by now)
                const uint32_t p = base + lane;
                bool valid = p < nk;
                const K cn = canon_from_words<K>(words, base, valid ? p : nk - 1, k);
                if (valid && hasN) {                          // any N in [p, p + k) ?
                    for (uint32_t d = p >> 5; d <= (p + k - 1) >> 5 && valid; d++) {
                        const uint32_t lo = d == (p >> 5) ? (p & 31) : 0, hi = d == ((p + k - 1) >> 5) ? ((p + k - 1) & 31) : 31;
                        const uint32_t m = (hi == 31 ? 0xFFFFFFFFu : ((1u << (hi + 1)) - 1)) & ~((1u << lo) - 1);
                        if (nm[d] & m) valid = false;
                    }
                }
                if (valid && n_parts > 1) valid = part_of(cn, n_parts) == part;
                const unsigned long long b = __ballot(valid);
                const uint32_t c = (uint32_t)__popcll(b);
                if (!c) continue;
                if (at + c > end) {                           // the next chunk
                    pad();
                    unsigned long long nb = 0;
                    if (lane == 0) nb = atomicAdd(cursor, (unsigned long long)PART_CHUNK);
                    nb = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(nb >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nb);
                    at = nb; end = nb + PART_CHUNK;
                }
                const uint64_t slot = at + (uint32_t)__popcll(b & ((1ull << lane) - 1));
                if (valid && slot < cap) store_kmer(keys + slot * KT<K>::W, cn);
                at += c;
            }
        } else if (inext < R.n) words_n = pass_words(R.packed + 2 * so_n, 0, lane);
        i = inext; len = len_n; so = so_n; ncnt = ncnt_n; words0 = words_n;
    }
    pad();
    if (lane == 0 && pads) atomicAdd(npad, (unsigned long long)pads);
}

__global__ void k_positions(const uint64_t* off, uint64_t n, uint32_t k, uint64_t* out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t len = off[i + 1] - off[i];
        out[i] = len >= k ? len - k + 1 : 0;
    }
}
// sorted keys -> flag the first element of every run of at least T equal keys; optional histogram of run lengths
template <uint32_t W>
__global__ void __launch_bounds__(256) k_flag_runs(const uint64_t* keys, uint64_t n, uint32_t T, uint8_t* flags, unsigned long long* hist, uint8_t* runlen) {
    // the spectrum is counted per workgroup in LDS and flushed once: nearly every distinct k-mer of a read set has abundance
    // 1 or 2, and hundreds of millions of atomics on the same two global words took longer than the sort (1.9 s at 5 M reads)
    __shared__ unsigned int sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        auto eq = [&](uint64_t a, uint64_t b) { return keys[a * W] == keys[b * W] && (W == 1 || keys[a * W + 1] == keys[b * W + 1]); };
        const bool head = i == 0 || !eq(i, i - 1);
        uint8_t f = 0;
        if (head) {
            f = (i + T - 1 < n && eq(i, i + T - 1)) ? 1 : 0;
            if (hist) {                                        // run length by doubling + binary search (runs are short)
                uint64_t lo = i, step = 1;
                while (lo + step < n && eq(i, lo + step)) { lo += step; step <<= 1; }
                while (step > 1) { step >>= 1; if (lo + step < n && eq(i, lo + step)) lo += step; }
                const uint64_t run = lo - i + 1;
                atomicAdd(&sh[run > 255 ? 255 : run], 1u);
                if (runlen) runlen[i] = (uint8_t)(run > 255 ? 255 : run);
            }
        }
        flags[i] = f;
    }
    __syncthreads();
    if (hist && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}
// ---- compaction of the flagged heads, in order: per-tile counts, a scan of the (few) tile counts, then every tile writes its own ----
// (rocPRIM's select took 22-27 ms per partition of 2^30 keys, ten times the bandwidth's time: the flags are sparse -- one key in
// twenty-five is the head of a solid run -- and byte flags of 8 items are one 64-bit load here)
constexpr uint32_t CT_TILE = 256 * 8;
__device__ inline uint64_t ct_flags8(const uint8_t* flags, uint64_t i0, uint64_t n) {
    if (i0 + 8 <= n) return *(const uint64_t*)(flags + i0) & 0x0101010101010101ull;
    uint64_t x = 0;
    for (uint32_t j = 0; j < 8 && i0 + j < n; j++) x |= (uint64_t)(flags[i0 + j] & 1u) << (8 * j);
    return x;
}
__device__ inline uint32_t ct_wave_incl(uint32_t c, uint32_t lane) {                // inclusive scan over the wave's lanes
    for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)c, d); if (lane >= d) c += o; }
    return c;
}
__global__ void __launch_bounds__(256) k_tile_counts(const uint8_t* flags, uint64_t n, uint64_t* counts) {
    __shared__ uint32_t part[4];
    const uint64_t i0 = blockIdx.x * (uint64_t)CT_TILE + threadIdx.x * 8;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t c = i0 < n ? (uint32_t)__popcll(ct_flags8(flags, i0, n)) : 0u;
    const uint32_t inc = ct_wave_incl(c, lane);
    if (lane == 63) part[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = (uint64_t)part[0] + part[1] + part[2] + part[3];
}
template <uint32_t W>
__global__ void __launch_bounds__(256) k_tile_compact(const uint8_t* flags, uint64_t n, const uint64_t* tile_off, const uint64_t* keys, uint64_t* dst,
                                                       const uint8_t* runlen, uint8_t* runsel) {
    __shared__ uint32_t part[4];
    const uint64_t i0 = blockIdx.x * (uint64_t)CT_TILE + threadIdx.x * 8;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint64_t x = i0 < n ? ct_flags8(flags, i0, n) : 0ull;
    const uint32_t c = (uint32_t)__popcll(x);
    const uint32_t inc = ct_wave_incl(c, lane);
    if (lane == 63) part[w] = inc;
    __syncthreads();
    uint64_t at = tile_off[blockIdx.x] + (inc - c);
    for (uint32_t v = 0; v < w; v++) at += part[v];
    for (uint64_t m = x; m; m &= m - 1) {
        const uint64_t i = i0 + ((uint32_t)__builtin_ctzll(m) >> 3);
        dst[at * W] = keys[i * W];
        if (W == 2) dst[at * W + 1] = keys[i * W + 1];
        if (runlen) runsel[at] = runlen[i];
        at++;
    }
}
__global__ void k_flag_at_least(const uint8_t* counts, uint64_t n, uint32_t T, uint8_t* flags) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) flags[i] = counts[i] >= T ? 1 : 0;
}
__global__ void k_split_words(const uint64_t* keys, uint64_t n, uint64_t* lo, uint64_t* hi) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { lo[i] = keys[2 * i]; hi[i] = keys[2 * i + 1]; }
}
__global__ void k_join_words(const uint64_t* lo, const uint64_t* hi, uint64_t n, uint64_t* keys) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { keys[2 * i] = lo[i]; keys[2 * i + 1] = hi[i]; }
}
struct Pair128 { uint64_t lo, hi; };

// The counter's large buffers are PARKED when it is done, not freed: memory this process has freed comes back from hipMalloc at
// the driver's wiping rate (10-43 GB/s), and the first thing a caller does after counting is create its context and size its
// per-batch buffers -- at 100 M reads one of those (16.9 GB) landed on the counter's 22 GB and took 1.6 s.  Parked buffers
// are taken again by the next counter call (same device, large enough) and returned by leon_device_trim(), which
// leon_dna_reserve calls once its own buffers exist and any allocation of the library calls before giving up.
struct Parked { void* p; size_t bytes; int device; };
std::mutex g_park_mu;
std::vector<Parked> g_parked;

struct Buf {
    void* p = nullptr;
    size_t bytes_ = 0;
    int park_device = -1;                                       // >= 0: park instead of free
    hipError_t alloc(size_t bytes) {
        static const bool trace = getenv("LEON_TRACE_ALLOC") != nullptr;       // (as in capi.hip: allocations of 1 ms or more on stderr)
        bytes_ = bytes ? bytes : 16;
        if (park_device >= 0) {                                 // a parked buffer of this device that is large enough (and not absurdly larger)?
            std::lock_guard<std::mutex> g(g_park_mu);
            for (size_t i = 0; i < g_parked.size(); i++)
                if (g_parked[i].device == park_device && g_parked[i].bytes >= bytes_ && g_parked[i].bytes <= 2 * bytes_ + (64u << 20)) {
                    p = g_parked[i].p; bytes_ = g_parked[i].bytes;
                    g_parked.erase(g_parked.begin() + (long)i);
                    return hipSuccess;
                }
        }
        if (park_device >= 0) leon_device_trim();               // nothing parked fits: what is parked goes back first (the list never grows past one call's buffers)
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&p, bytes_);
        if (e != hipSuccess) { (void)hipGetLastError(); leon_device_trim(); e = hipMalloc(&p, bytes_); }
        if (trace) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms >= 1.0) fprintf(stderr, "[leon alloc] kmer: %.1f MB in %.1f ms\n", bytes_ / 1e6, ms);
        }
        return e;
    }
    ~Buf() {
        if (!p) return;
        if (park_device >= 0) { std::lock_guard<std::mutex> g(g_park_mu); g_parked.push_back({p, bytes_, park_device}); return; }
        static const bool trace = getenv("LEON_TRACE_ALLOC") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        (void)hipFree(p);
        if (trace) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms >= 1.0) fprintf(stderr, "[leon alloc] kmer: free in %.1f ms\n", ms);
        }
    }
    template <typename T> T* as() { return (T*)p; }
};
uint32_t grid(uint64_t n, uint32_t per = 256, uint32_t cap = 8192) { return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n + per - 1) / per, cap)); }

#define KCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_create_error(std::string(#call) + ": " + hipGetErrorString(e_)); return LEON_E_HIP; } } while (0)

}  // namespace
}  // namespace leon

using namespace leon;

extern "C" {

void leon_device_free(void* p) { if (p) (void)hipFree(p); }

void leon_device_trim(void) {
    std::vector<Parked> take;
    { std::lock_guard<std::mutex> g(g_park_mu); take.swap(g_parked); }
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (const Parked& b : take) { if (hipSetDevice(b.device) == hipSuccess) (void)hipFree(b.p); }
    if (cur >= 0) (void)hipSetDevice(cur);
}

int leon_device_count(int* n_devices) {
    if (!n_devices) return LEON_E_INVALID;
    *n_devices = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { set_create_error("no usable HIP device"); return LEON_E_NO_DEVICE; }
    *n_devices = n;
    return LEON_OK;
}
int leon_device_alloc(int device_id, uint64_t bytes, void** d_ptr) {
    if (!d_ptr) return LEON_E_INVALID;
    *d_ptr = nullptr;
    KCHK(hipSetDevice(device_id));
    KCHK(hipMalloc(d_ptr, bytes ? bytes : 16));
    return LEON_OK;
}
int leon_device_upload(int device_id, void* d_dst, const void* src, uint64_t bytes) {
    if (bytes && (!d_dst || !src)) return LEON_E_INVALID;
    KCHK(hipSetDevice(device_id));
    if (bytes) KCHK(staged_h2d(device_id, d_dst, src, bytes));
    return LEON_OK;
}
int leon_device_copy(int device_id, void* d_dst, const void* d_src, uint64_t bytes) {
    if (bytes && (!d_dst || !d_src)) return LEON_E_INVALID;
    KCHK(hipSetDevice(device_id));
    if (bytes) {
        KCHK(hipMemcpy(d_dst, d_src, bytes, hipMemcpyDeviceToDevice));
        // (a device-to-device hipMemcpy may return before the copy has run: a caller that hands d_dst to work on ANOTHER stream next --
        // the exchange callbacks of leon_dna_set_exchange / leon_dna_set_gather do -- must find it complete)
        KCHK(hipStreamSynchronize(nullptr));
    }
    return LEON_OK;
}

int leon_device_download(int device_id, void* dst, const void* d_src, uint64_t bytes) {
    if (bytes && (!dst || !d_src)) return LEON_E_INVALID;
    KCHK(hipSetDevice(device_id));
    if (bytes) KCHK(staged_d2h(device_id, dst, d_src, bytes));          // (at PCIe's rate for large copies: staging.h)
    return LEON_OK;
}

// The abundance threshold Leon uses when `-abundance` is absent ("default: automatic", /root/reference/README.md:54; upstream
// asks DSK for `-abundance-min auto` [RECALLED]): the first local minimum of the abundance spectrum -- the valley between the
// sequencing-error k-mers (abundance 1, 2, ...) and the genomic ones around the coverage -- never below 2, and 2 when the
// spectrum has no valley (coverage too low to separate them).  histogram[a] = distinct k-mers of abundance a, a = 255 and more
// in the last entry.  Host-only.
int leon_kmer_auto_cutoff(const uint64_t* histogram, uint32_t* cutoff) {
    if (!histogram || !cutoff) return LEON_E_INVALID;
    *cutoff = 2;
    uint32_t last = 0;
    for (uint32_t a = 1; a < 256; a++) if (histogram[a]) last = a;
    for (uint32_t a = 1; a < last; a++)                          // the spectrum stops falling at a: k-mers seen a times or more are kept
        if (histogram[a + 1] >= histogram[a]) { *cutoff = a < 2 ? 2 : a; break; }
    return LEON_OK;
}

int leon_kmer_solid_device(int device_id, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint32_t k,
                           uint32_t min_abundance, uint64_t max_keys_per_pass, uint64_t** d_solid, uint64_t* n_solid,
                           uint64_t* histogram) {
    if (!d_solid || !n_solid || (n_reads && (!d_bases || !d_offsets))) return LEON_E_INVALID;
    if (k < 3 || k > 63) { set_create_error("kmer_solid: need 3 <= k <= 63"); return LEON_E_INVALID; }
    // min_abundance 0 = automatic (Leon's default, /root/reference/README.md:54): k-mers seen twice or more are kept with their
    // abundance while the partitions go by, the threshold comes from the whole spectrum (leon_kmer_auto_cutoff), then they are filtered
    const bool automatic = min_abundance == 0;
    if (automatic) min_abundance = 2;
    *d_solid = nullptr; *n_solid = 0;
    if (histogram) memset(histogram, 0, 256 * sizeof(uint64_t));
    if (!n_reads) return LEON_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) { set_create_error("kmer_solid: no such HIP device"); return LEON_E_NO_DEVICE; }
    KCHK(hipSetDevice(device_id));
    hipStream_t s = nullptr;
    const uint32_t W = kmer_words(k);
    // ---- pack the reads once ----
    Buf slot_off, packed, nmask, rlen, ncount, tmp, pos;
    Buf bad;
    KCHK(slot_off.alloc((n_reads + 1) * 8));
    KCHK(bad.alloc(4));
    KCHK(hipMemsetAsync(bad.p, 0, 4, s));
    launch_read_slots(s, d_offsets, n_reads, slot_off.as<uint64_t>(), bad.as<uint32_t>());
    size_t tb = 0;
    KCHK(prim::ExclusiveSum(nullptr, tb, slot_off.as<uint64_t>(), slot_off.as<uint64_t>(), n_reads + 1, s));
    KCHK(tmp.alloc(tb));
    KCHK(prim::ExclusiveSum(tmp.p, tb, slot_off.as<uint64_t>(), slot_off.as<uint64_t>(), n_reads + 1, s));
    uint64_t n_slots = 0;
    KCHK(hipMemcpy(&n_slots, slot_off.as<uint64_t>() + n_reads, 8, hipMemcpyDeviceToHost));
    uint32_t bad_offsets = 0;
    KCHK(hipMemcpy(&bad_offsets, bad.p, 4, hipMemcpyDeviceToHost));
    if (bad_offsets) { set_create_error("kmer_solid: offsets are not monotonic (or a read is longer than 2^31 bases)"); return LEON_E_INVALID; }
    KCHK(packed.alloc((n_slots * 2 + 16) * 4)); KCHK(nmask.alloc((n_slots + 4) * 4)); KCHK(rlen.alloc(n_reads * 4)); KCHK(ncount.alloc(n_reads * 4));
    KCHK(hipMemsetAsync(packed.as<uint32_t>() + n_slots * 2, 0, 64, s));
    launch_pack(s, d_bases, d_offsets, slot_off.as<uint64_t>(), n_reads, packed.as<uint32_t>(), nmask.as<uint32_t>(), rlen.as<uint32_t>(), ncount.as<uint32_t>());
    ReadsDev R{};
    R.packed = packed.as<uint32_t>(); R.nmask = nmask.as<uint32_t>(); R.slot_off = slot_off.as<uint64_t>(); R.base_off = d_offsets;
    R.len = rlen.as<uint32_t>(); R.n_count = ncount.as<uint32_t>(); R.n = n_reads; R.k = k;
    // ---- how many k-mer positions, hence how many partitions ----
    KCHK(pos.alloc((n_reads + 1) * 8));
    hipLaunchKernelGGL(k_positions, dim3(grid(n_reads)), dim3(256), 0, s, d_offsets, n_reads, k, pos.as<uint64_t>());
    Buf tmp2; size_t tb2 = 0;
    KCHK(prim::Sum(nullptr, tb2, pos.as<uint64_t>(), pos.as<uint64_t>() + n_reads, n_reads, s));
    KCHK(tmp2.alloc(tb2));
    KCHK(prim::Sum(tmp2.p, tb2, pos.as<uint64_t>(), pos.as<uint64_t>() + n_reads, n_reads, s));
    uint64_t total = 0;
    KCHK(hipMemcpy(&total, pos.as<uint64_t>() + n_reads, 8, hipMemcpyDeviceToHost));
    if (!total) return LEON_OK;
    if (!max_keys_per_pass) {
        // 2^30 k-mers per pass (17-34 GB of sort buffers), less when a third of the free HBM does not hold them.  NOT "as
        // many as fit": the driver wipes freed VRAM at ~55 GB/s before handing it out again, so a counter that took and
        // returned 200 GB made the first allocation of the encode path wait 3.8 s (profiles/README.md, round 2), and
        // passes of 2^30 keys sort no slower than one of 2^32.
        size_t free_b = 0, total_b = 0;
        KCHK(hipMemGetInfo(&free_b, &total_b));
        max_keys_per_pass = std::max<uint64_t>(1ull << 24, std::min<uint64_t>(1ull << 30, free_b / 3 / (8 * W * 2 + 1)));
    }
    uint32_t n_parts = (uint32_t)((total + max_keys_per_pass - 1) / max_keys_per_pass);
    if (n_parts < 1) n_parts = 1;
    const uint32_t part_grid = grid(n_reads, 4, 256 * 16);
    // (every wave of the partition pass leaves at most one chunk partly used, and up to 63 slots at the end of each chunk)
    const uint64_t cap = (n_parts == 1 ? total : (uint64_t)(total / n_parts * 1.25) + (1u << 20)) + (uint64_t)part_grid * 4 * PART_CHUNK + total / n_parts / 8;
    // ---- per-partition buffers ----
    Buf keys, alt, alt2, alt3, flags, nsel, hist, sort_tmp, sel_tmp, runlen, runsel;
    keys.park_device = alt.park_device = alt2.park_device = alt3.park_device = device_id;       // (the large ones: see Parked)
    KCHK(keys.alloc(cap * 8 * W)); KCHK(alt.alloc(cap * 8 * W)); KCHK(flags.alloc(cap));
    if (automatic) { KCHK(runlen.alloc(cap)); KCHK(runsel.alloc(cap)); }
    if (W == 2) { KCHK(alt2.alloc(cap * 8)); KCHK(alt3.alloc(cap * 8)); }
    KCHK(nsel.alloc(8)); KCHK(hist.alloc(256 * 8));
    KCHK(hipMemset(hist.p, 0, 256 * 8));
    size_t st = 0, st2 = 0, sl = 0;
    if (W == 1) { KCHK(prim::SortKeys(nullptr, st, keys.as<uint64_t>(), alt.as<uint64_t>(), cap, 0, 2 * k, s)); }
    else {
        KCHK(prim::SortPairs(nullptr, st, keys.as<uint64_t>(), alt.as<uint64_t>(), alt2.as<uint64_t>(), alt3.as<uint64_t>(), cap, 0, 64, s));
        st2 = st;
    }
    KCHK(sort_tmp.alloc(std::max(st, st2)));
    // the solid k-mers accumulate here (grown geometrically)
    uint64_t out_cap = std::max<uint64_t>(total / 16, 1u << 20), out_n = 0;
    uint64_t* out = nullptr;
    uint8_t* out_cnt = nullptr;                               // automatic: abundance (clipped at 255) of every kept k-mer
    KCHK(hipMalloc((void**)&out, out_cap * 8 * W));
    auto fail_free = [&](int code) { if (out) (void)hipFree(out); if (out_cnt) (void)hipFree(out_cnt); return code; };
    const bool want_hist = histogram != nullptr || automatic;
#define KCHK2(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_create_error(std::string(#call) + ": " + hipGetErrorString(e_)); return fail_free(LEON_E_HIP); } } while (0)
    if (automatic) KCHK2(hipMalloc((void**)&out_cnt, out_cap));
    Buf cursor, tile_cnt, tile_tmp;
    KCHK2(cursor.alloc(16));
    const size_t max_tiles = (size_t)(cap / CT_TILE) + 2;
    KCHK2(tile_cnt.alloc(max_tiles * 8));
    size_t tile_tb = 0;
    KCHK2(prim::ExclusiveSum(nullptr, tile_tb, tile_cnt.as<uint64_t>(), tile_cnt.as<uint64_t>(), max_tiles, s));
    KCHK2(tile_tmp.alloc(tile_tb));
    const bool trace = getenv("LEON_TRACE_KMER") != nullptr;
    for (uint32_t part = 0; part < n_parts; part++) {
        // one pass: the partition's k-mers in whatever order the waves' chunks make it (the sort follows), padded with all-ones keys
        KCHK2(hipMemsetAsync(cursor.p, 0, 16, s));
        if (W == 1) hipLaunchKernelGGL((k_part_kmers<uint64_t>), dim3(part_grid), dim3(256), 0, s, R, n_parts, part, keys.as<uint64_t>(), cap, cursor.as<unsigned long long>(), cursor.as<unsigned long long>() + 1);
        else hipLaunchKernelGGL((k_part_kmers<u128>), dim3(part_grid), dim3(256), 0, s, R, n_parts, part, keys.as<uint64_t>(), cap, cursor.as<unsigned long long>(), cursor.as<unsigned long long>() + 1);
        uint64_t cur[2] = {0, 0};
        KCHK2(hipMemcpy(cur, cursor.p, 16, hipMemcpyDeviceToHost));
        if (cur[0] > cap) { set_create_error("kmer_solid: a hash partition exceeds its buffer (very skewed k-mer spectrum); lower max_keys_per_pass"); return fail_free(LEON_E_OVERFLOW); }
        const uint64_t n_all = cur[0], n = cur[0] - cur[1];      // with and without the padding, which sorts to the end
        if (trace) fprintf(stderr, "[leon kmer] partition %u of %u: %llu k-mers (+ %llu of padding), buffer %llu\n", part, n_parts, (unsigned long long)n, (unsigned long long)cur[1], (unsigned long long)cap);
        if (!n) continue;
        uint64_t* sorted = nullptr;
        if (W == 1) {
            KCHK2(prim::SortKeys(sort_tmp.p, st, keys.as<uint64_t>(), alt.as<uint64_t>(), n_all, 0, 2 * k, s));
            sorted = alt.as<uint64_t>();
        } else {                                              // 128-bit keys: LSD in two stable passes (low word, then high word)
            uint64_t* lo = alt.as<uint64_t>(); uint64_t* hi = lo + cap;
            hipLaunchKernelGGL(k_split_words, dim3(grid(n_all)), dim3(256), 0, s, keys.as<uint64_t>(), n_all, lo, hi);
            KCHK2(prim::SortPairs(sort_tmp.p, st, lo, alt2.as<uint64_t>(), hi, alt3.as<uint64_t>(), n_all, 0, 64, s));     // by low
            KCHK2(prim::SortPairs(sort_tmp.p, st, alt3.as<uint64_t>(), hi, alt2.as<uint64_t>(), lo, n_all, 0, 2 * k - 64 > 0 ? 2 * k - 64 : 1, s));   // by high (stable)
            hipLaunchKernelGGL(k_join_words, dim3(grid(n_all)), dim3(256), 0, s, lo, hi, n_all, keys.as<uint64_t>());
            sorted = keys.as<uint64_t>();
        }
        if (W == 1) hipLaunchKernelGGL(k_flag_runs<1>, dim3(grid(n)), dim3(256), 0, s, sorted, n, min_abundance, flags.as<uint8_t>(), want_hist ? hist.as<unsigned long long>() : nullptr, runlen.as<uint8_t>());
        else hipLaunchKernelGGL(k_flag_runs<2>, dim3(grid(n)), dim3(256), 0, s, sorted, n, min_abundance, flags.as<uint8_t>(), want_hist ? hist.as<unsigned long long>() : nullptr, runlen.as<uint8_t>());
        // compact the heads of solid runs behind what earlier partitions produced
        uint64_t* dst = (sorted == keys.as<uint64_t>()) ? alt.as<uint64_t>() : keys.as<uint64_t>();
        const uint32_t n_tiles = (uint32_t)((n + CT_TILE - 1) / CT_TILE);
        KCHK2(hipMemsetAsync(tile_cnt.as<uint64_t>() + n_tiles, 0, 8, s));
        hipLaunchKernelGGL(k_tile_counts, dim3(n_tiles), dim3(256), 0, s, flags.as<uint8_t>(), n, tile_cnt.as<uint64_t>());
        KCHK2(prim::ExclusiveSum(tile_tmp.p, tile_tb, tile_cnt.as<uint64_t>(), tile_cnt.as<uint64_t>(), (size_t)n_tiles + 1, s));
        if (W == 1) hipLaunchKernelGGL(k_tile_compact<1>, dim3(n_tiles), dim3(256), 0, s, flags.as<uint8_t>(), n, tile_cnt.as<uint64_t>(), sorted, dst,
                                       automatic ? runlen.as<uint8_t>() : (const uint8_t*)nullptr, runsel.as<uint8_t>());
        else hipLaunchKernelGGL(k_tile_compact<2>, dim3(n_tiles), dim3(256), 0, s, flags.as<uint8_t>(), n, tile_cnt.as<uint64_t>(), sorted, dst,
                                automatic ? runlen.as<uint8_t>() : (const uint8_t*)nullptr, runsel.as<uint8_t>());
        uint64_t ns = 0;
        KCHK2(hipMemcpy(&ns, tile_cnt.as<uint64_t>() + n_tiles, 8, hipMemcpyDeviceToHost));
        if (out_n + ns > out_cap) {
            uint64_t nc = std::max(out_cap * 2, out_n + ns);
            uint64_t* bigger = nullptr;
            KCHK2(hipMalloc((void**)&bigger, nc * 8 * W));
            if (out_n) KCHK2(hipMemcpy(bigger, out, out_n * 8 * W, hipMemcpyDeviceToDevice));
            (void)hipFree(out); out = bigger;
            if (automatic) {
                uint8_t* bc = nullptr;
                KCHK2(hipMalloc((void**)&bc, nc));
                if (out_n) KCHK2(hipMemcpy(bc, out_cnt, out_n, hipMemcpyDeviceToDevice));
                (void)hipFree(out_cnt); out_cnt = bc;
            }
            out_cap = nc;
        }
        if (ns) KCHK2(hipMemcpy(out + out_n * W, dst, ns * 8 * W, hipMemcpyDeviceToDevice));
        if (ns && automatic) KCHK2(hipMemcpy(out_cnt + out_n, runsel.p, ns, hipMemcpyDeviceToDevice));
        out_n += ns;
    }
    uint64_t h_hist[256] = {0};
    if (want_hist) KCHK2(hipMemcpy(h_hist, hist.p, 256 * 8, hipMemcpyDeviceToHost));
    if (histogram) memcpy(histogram, h_hist, sizeof h_hist);
    if (automatic && out_n) {
        uint32_t cutoff = 2;
        (void)leon_kmer_auto_cutoff(h_hist, &cutoff);
        if (cutoff > 2) {                                       // drop what is below the spectrum's own threshold
            Buf f2, kept;
            KCHK2(f2.alloc(out_n)); KCHK2(kept.alloc(out_n * 8 * W));
            hipLaunchKernelGGL(k_flag_at_least, dim3(grid(out_n)), dim3(256), 0, s, out_cnt, out_n, cutoff, f2.as<uint8_t>());
            size_t need3 = 0;
            if (W == 1) {
                KCHK2(prim::Flagged(nullptr, need3, out, f2.as<uint8_t>(), kept.as<uint64_t>(), nsel.as<uint64_t>(), out_n, s));
                if (need3 > sl) { if (sel_tmp.p) { (void)hipFree(sel_tmp.p); sel_tmp.p = nullptr; } KCHK2(sel_tmp.alloc(need3)); sl = need3; }
                KCHK2(prim::Flagged(sel_tmp.p, need3, out, f2.as<uint8_t>(), kept.as<uint64_t>(), nsel.as<uint64_t>(), out_n, s));
            } else {
                KCHK2(prim::Flagged(nullptr, need3, (Pair128*)out, f2.as<uint8_t>(), (Pair128*)kept.p, nsel.as<uint64_t>(), out_n, s));
                if (need3 > sl) { if (sel_tmp.p) { (void)hipFree(sel_tmp.p); sel_tmp.p = nullptr; } KCHK2(sel_tmp.alloc(need3)); sl = need3; }
                KCHK2(prim::Flagged(sel_tmp.p, need3, (Pair128*)out, f2.as<uint8_t>(), (Pair128*)kept.p, nsel.as<uint64_t>(), out_n, s));
            }
            uint64_t nk = 0;
            KCHK2(hipMemcpy(&nk, nsel.p, 8, hipMemcpyDeviceToHost));
            if (nk) KCHK2(hipMemcpy(out, kept.p, nk * 8 * W, hipMemcpyDeviceToDevice));
            out_n = nk;
        }
    }
    KCHK2(hipDeviceSynchronize());
    if (out_cnt) (void)hipFree(out_cnt);
    *d_solid = out; *n_solid = out_n;
    return LEON_OK;
}

int leon_kmer_solid(int device_id, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t k, uint32_t min_abundance,
                    uint64_t max_keys_per_pass, uint64_t* out, uint64_t out_cap, uint64_t* n_solid, uint64_t* histogram) {
    if (!n_solid || (n_reads && (!bases || !offsets))) return LEON_E_INVALID;
    *n_solid = 0;
    if (!n_reads) return LEON_OK;
    if (hipSetDevice(device_id) != hipSuccess) { set_create_error("kmer_solid: no such HIP device"); return LEON_E_NO_DEVICE; }
    const uint64_t nb = offsets[n_reads] - offsets[0];
    std::vector<uint64_t> rel(n_reads + 1);
    for (uint64_t i = 0; i <= n_reads; i++) rel[i] = offsets[i] - offsets[0];
    Buf db, doff;
    KCHK(db.alloc(nb + 64)); KCHK(doff.alloc((n_reads + 1) * 8));
    KCHK(hipMemcpy(db.p, bases + offsets[0], nb, hipMemcpyHostToDevice));
    KCHK(hipMemcpy(doff.p, rel.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice));
    uint64_t* d = nullptr;
    int rc = leon_kmer_solid_device(device_id, db.as<uint8_t>(), doff.as<uint64_t>(), n_reads, k, min_abundance, max_keys_per_pass, &d, n_solid, histogram);
    if (rc) return rc;
    const uint32_t W = kmer_words(k);
    if (out && *n_solid <= out_cap && *n_solid) KCHK(hipMemcpy(out, d, *n_solid * 8 * W, hipMemcpyDeviceToHost));
    leon_device_free(d);
    return (out && *n_solid > out_cap) ? LEON_E_OVERFLOW : LEON_OK;
}

}  // extern "C"

// deflate_kernels.hip -- the quality stream's zlib blocks on the device (upstream deflates the block's
// buffered quality lines, Leon::writeBlockLena [RECALLED]).  One zlib stream per read block, as leon_host_qual_encode_blocks
// writes them, any inflate reads them; what differs is the encoder.  zlib's default strategy spends its time in hash chains
// looking for matches that quality strings rarely pay for (a three-byte match costs more bits than three literals of a 3-bit
// alphabet); what does pay is the run -- a quality value repeated -- and the entropy code.  So this is deflate with the RLE
// strategy (zlib's own Z_RLE: matches at distance 1 only) and dynamic Huffman codes per 32 KB of text, which needs no search
// and no serial parse: where a run starts and how long it is follows from comparing neighbours, every position knows from
// (run start, run length) alone which token starts at it, and the bit offsets are a prefix sum.
//   text -> chunks of DF_CHUNK bytes, one workgroup each, one deflate block each (dynamic codes, or stored when that is smaller),
//   closed on a byte boundary by an empty stored block (Z_SYNC_FLUSH's marker), so the chunks of a read block are concatenated
//   bytes; the read block's stream = 78 01, its chunks, a final empty stored block, Adler-32 (combined on the host from the
//   chunks' own).
#include "../../include/leon_dna.h"
#include "kernels.h"
#include "prim.h"
#include "staging.h"

#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

namespace leon {
void set_create_error(const std::string& msg);

namespace {

constexpr uint32_t DF_CHUNK = 32768;                          // text bytes per deflate block
constexpr uint32_t DF_T = 256;
constexpr uint32_t DF_PER = DF_CHUNK / DF_T;                  // consecutive positions per thread
constexpr uint32_t DF_OUT_STRIDE = DF_CHUNK + 64;             // bytes a chunk may write (stored form: text + 5)
constexpr uint32_t DF_NLIT = 286, DF_NCL = 19;
constexpr uint32_t DF_MAXSYM = 288;

// the quality lines of reads [r0, r1) each followed by '\n', as one text
__global__ void k_qual_text(const uint8_t* quals, const uint64_t* off, uint64_t n_reads, uint64_t q0, uint8_t* text) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n_reads; r += nwaves) {
        const uint64_t a = off[r] - q0, b = off[r + 1] - q0;
        uint8_t* d = text + a + r;
        for (uint64_t i = lane; i < b - a; i += 64) d[i] = quals[a + i];
        if (lane == 0) d[b - a] = '\n';
    }
}

__device__ const uint16_t df_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t df_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint8_t df_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ inline uint32_t df_len_code(uint32_t len) {        // 3..258 -> 0..28
    if (len == 258) return 28;
    const uint32_t l = len - 3;
    if (l < 8) return l;
    const uint32_t e = 29 - (uint32_t)__builtin_clz(l);       // extra bits: l in [2^(e+2), 2^(e+3))
    return 4 * e + 4 + ((l >> e) & 3u);
}
__device__ inline uint32_t df_rev(uint32_t code, uint32_t len) { return __builtin_bitreverse32(code) >> (32 - len); }

// Huffman code lengths of `n` symbols from their frequencies, none above `limit`, by the whole workgroup: the used symbols are
// ranked by (frequency, symbol) -- every thread counts the symbols before its own, broadcast reads, no sort -- then one lane runs
// Moffat & Katajainen's in-place algorithm over the ranked frequencies; when the depth exceeds the limit the frequencies are
// halved (rounded up) and it runs again (zlib moves the overflowing leaves instead; either gives a complete code).
// Scratch in LDS: A[n], order[n], fr2[n], ctl[2].
__device__ void df_code_lengths(const uint32_t* freq, uint32_t n, uint32_t limit, uint8_t* len, uint32_t* A, uint16_t* order, uint32_t* fr2, uint32_t* ctl) {
    const uint32_t t = threadIdx.x;
    for (uint32_t i = t; i < n; i += DF_T) { fr2[i] = freq[i]; len[i] = 0; }
    __syncthreads();
    for (;;) {
        if (t == 0) ctl[0] = 0;
        __syncthreads();
        for (uint32_t i = t; i < n; i += DF_T) {
            const uint32_t f = fr2[i];
            if (!f) continue;
            uint32_t r = 0;
            for (uint32_t j = 0; j < n; j++) { const uint32_t g = fr2[j]; r += (g && (g < f || (g == f && j < i))) ? 1u : 0u; }
            order[r] = (uint16_t)i; A[r] = f;
            atomicAdd(&ctl[0], 1u);
        }
        __syncthreads();
        if (t == 0) {
            const uint32_t m = ctl[0];
            uint32_t done = 1;
            if (m == 1) { const uint32_t i = order[0]; len[i] = 1; len[i ? i - 1 : 1] = 1; }   // one symbol: it and a neighbour get one bit each (inflate wants a complete code)
            else if (m >= 2) {
                A[0] += A[1];
                uint32_t root = 0, leaf = 2;
                for (uint32_t next = 1; next + 1 < m; next++) {
                    if (leaf >= m || A[root] < A[leaf]) { A[next] = A[root]; A[root++] = next; } else A[next] = A[leaf++];
                    if (leaf >= m || (root < next && A[root] < A[leaf])) { A[next] += A[root]; A[root++] = next; } else A[next] += A[leaf++];
                }
                A[m - 2] = 0;
                for (int next = (int)m - 3; next >= 0; next--) A[next] = A[A[next]] + 1;
                int avbl = 1, usedn = 0, dpth = 0, rt = (int)m - 2, nx = (int)m - 1;
                while (avbl > 0) {
                    while (rt >= 0 && (int)A[rt] == dpth) { usedn++; rt--; }
                    while (avbl > usedn) { A[nx--] = (uint32_t)dpth; avbl--; }
                    avbl = 2 * usedn; dpth++; usedn = 0;
                }
                done = A[0] <= limit ? 1u : 0u;               // A[0]: the rarest symbol's length, the longest
            }
            ctl[1] = done | (m << 1);
        }
        __syncthreads();
        const uint32_t m = ctl[1] >> 1;
        if (ctl[1] & 1u) {
            if (m >= 2) for (uint32_t i = t; i < m; i += DF_T) len[order[i]] = (uint8_t)A[i];
            __syncthreads();
            return;
        }
        for (uint32_t i = t; i < n; i += DF_T) if (fr2[i]) fr2[i] = (fr2[i] + 1) >> 1;
        __syncthreads();
    }
}
// canonical codes (RFC 1951 3.2.2), bit-reversed for the LSB-first stream; cnt[16], nxt[16] in LDS
__device__ void df_codes(const uint8_t* len, uint32_t n, uint16_t* code, uint32_t* cnt, uint32_t* nxt) {
    const uint32_t t = threadIdx.x;
    if (t < 16) cnt[t] = 0;
    __syncthreads();
    for (uint32_t i = t; i < n; i += DF_T) if (len[i]) atomicAdd(&cnt[len[i]], 1u);
    __syncthreads();
    if (t == 0) { uint32_t c = 0; nxt[0] = 0; for (uint32_t b = 1; b < 16; b++) { c = (c + cnt[b - 1]) << 1; nxt[b] = c; } }
    __syncthreads();
    for (uint32_t i = t; i < n; i += DF_T) {
        const uint32_t l = len[i];
        uint32_t r = 0;
        if (l) for (uint32_t j = 0; j < i; j++) r += len[j] == l ? 1u : 0u;
        code[i] = l ? (uint16_t)df_rev(nxt[l] + r, l) : (uint16_t)0;
    }
    __syncthreads();
}

struct BitW {                                                  // one lane's writer into a zeroed word buffer
    uint32_t* w; uint64_t pos;
    __device__ void put(uint32_t v, uint32_t nbits) {
        if (!nbits) return;
        const uint32_t i = (uint32_t)(pos >> 5), s = (uint32_t)(pos & 31);
        w[i] |= v << s;
        if (s + nbits > 32) w[i + 1] |= v >> (32 - s);
        pos += nbits;
    }
};

// The tokens of the positions [a, b] of one maximal run of equal bytes clipped to a thread's range: the run is [S, S + L) in the
// chunk.  RLE parse: the run's first byte is a literal; the R = L - 1 bytes after it are matches of 258 at distance 1, then one
// match of the remainder when it is 3 or more, else literals.  f(kind, length): kind 0 literal (length = count), 1 match.
template <typename F> __device__ inline void df_run_tokens(uint32_t S, uint32_t L, uint32_t a, uint32_t b, F f) {
    if (a == S) { f(0u, 1u); if (a == b) return; }
    const uint32_t R = L - 1, full = R / 258, rem = R - full * 258;
    const uint32_t ja = (a == S ? a + 1 : a) - (S + 1), jb = b - (S + 1);     // offsets into the R region, inclusive
    if (full) {                                                              // full matches start at j = 0, 258, ...: those with ja <= j <= jb
        const uint32_t first = (ja + 257) / 258, last = jb / 258;
        const uint32_t lastc = last < full - 1 ? last : full - 1;
        for (uint32_t i = first; i <= lastc && first <= lastc; i++) f(1u, 258u);
    }
    const uint32_t j0 = full * 258;                                          // the remainder region [j0, R)
    if (rem && jb >= j0) {
        if (rem >= 3) { if (ja <= j0) f(1u, rem); }
        else f(0u, (jb - (ja > j0 ? ja : j0)) + 1);
    }
}

// One chunk = one deflate block.  sizes[c] = bytes written at out + c * DF_OUT_STRIDE; adler[c] = Adler-32 of the chunk's text.
__global__ void __launch_bounds__(DF_T) k_deflate_chunks(const uint8_t* text, const uint64_t* chunk_begin, const uint32_t* chunk_len, uint64_t n_chunks,
                                                          uint8_t* out, uint32_t* sizes, uint32_t* adler) {
    __shared__ uint32_t textw[DF_CHUNK / 4 + 2];
    __shared__ uint32_t outw[DF_OUT_STRIDE / 4 + 4];
    __shared__ uint32_t freq[DF_MAXSYM], A[DF_MAXSYM], fr2[DF_MAXSYM], scan_sh[DF_T + 1], clfreq[DF_NCL];
    __shared__ uint16_t code[DF_MAXSYM], order[DF_MAXSYM], clcode[DF_NCL], hdr_sym[DF_MAXSYM + 8];
    __shared__ uint8_t len[DF_MAXSYM], cllen[DF_NCL], hdr_extra[DF_MAXSYM + 8];
    __shared__ uint8_t tfirst[DF_T], tlast[DF_T];
    __shared__ uint16_t tlead[DF_T], ttrail[DF_T], tn[DF_T];
    __shared__ uint32_t sh_misc[8], ctl[2], cnt16[16], nxt16[16];                               // 0 header bits, 1 total bits, 2 n header symbols, 3 hlit, 4 hclen
    uint8_t* const tb = (uint8_t*)textw;
    const uint32_t t = threadIdx.x;
    for (uint64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const uint32_t n = chunk_len[c];
        const uint8_t* src = text + chunk_begin[c];
        __syncthreads();
        for (uint32_t i = t; i < n; i += DF_T) tb[i] = src[i];
        for (uint32_t i = t; i < DF_OUT_STRIDE / 4 + 4; i += DF_T) outw[i] = 0;
        for (uint32_t i = t; i < DF_MAXSYM; i += DF_T) freq[i] = 0;
        if (t < DF_NCL) clfreq[t] = 0;
        __syncthreads();
        // ---- this thread's positions [p0, p1), their first / last byte, leading and trailing runs; Adler-32 partials ----
        const uint32_t p0 = t * DF_PER < n ? t * DF_PER : n, p1 = (t + 1) * DF_PER < n ? (t + 1) * DF_PER : n, cnt = p1 - p0;
        uint32_t lead = 0, trail = 0, sa = 0, sb = 0;
        if (cnt) {
            const uint8_t f0 = tb[p0], l0 = tb[p1 - 1];
            while (lead < cnt && tb[p0 + lead] == f0) lead++;
            while (trail < cnt && tb[p1 - 1 - trail] == l0) trail++;
            for (uint32_t i = 0; i < cnt; i++) { sa += tb[p0 + i]; sb += sa; }       // (128 * 255 and 128 * 129 / 2 * 255: no overflow)
            tfirst[t] = f0; tlast[t] = l0;
        }
        tlead[t] = (uint16_t)lead; ttrail[t] = (uint16_t)trail; tn[t] = (uint16_t)cnt;
        // Adler-32 over the chunk: A = 1 + sum of bytes, B = n + sum over bytes of (n - i) * byte = n + sum_t (sb_t + (n - p1_t) * sa_t)
        {
            uint64_t va = sa, vb = (uint64_t)sb + (uint64_t)(n - p1) * sa;
            for (uint32_t d = 32; d; d >>= 1) { va += __shfl_down((unsigned long long)va, d); vb += __shfl_down((unsigned long long)vb, d); }
            __syncthreads();
            if ((t & 63) == 0) { scan_sh[2 * (t >> 6)] = (uint32_t)(va % 65521u); scan_sh[2 * (t >> 6) + 1] = (uint32_t)(vb % 65521u); }
            __syncthreads();
            if (t == 0) {
                const uint32_t a32 = (1u + scan_sh[0] + scan_sh[2] + scan_sh[4] + scan_sh[6]) % 65521u;
                const uint32_t b32 = (uint32_t)(((uint64_t)n + scan_sh[1] + scan_sh[3] + scan_sh[5] + scan_sh[7]) % 65521u);
                adler[c] = (b32 << 16) | a32;
            }
        }
        __syncthreads();
        // how far the run at this thread's first position reaches back, how far the run at its last position reaches forward
        uint32_t back = 0, fwd = 0;
        if (cnt) {
            for (int q = (int)t - 1; q >= 0 && tn[q] && tlast[q] == tfirst[t]; q--) { back += ttrail[q]; if (ttrail[q] < tn[q]) break; }
            for (uint32_t q = t + 1; q < DF_T && tn[q] && tfirst[q] == tlast[t]; q++) { fwd += tlead[q]; if (tlead[q] < tn[q]) break; }
        }
        // every maximal local run [a, b] of this thread with its global start S and length L
        auto each_run = [&](auto g) {
            uint32_t a = p0;
            while (a < p1) {
                const uint8_t v = tb[a];
                uint32_t b = a;
                while (b + 1 < p1 && tb[b + 1] == v) b++;
                const uint32_t S = a == p0 ? a - back : a;
                const uint32_t E = b == p1 - 1 ? b + fwd : b;
                g(v, S, E - S + 1, a, b);
                a = b + 1;
            }
        };
        // ---- pass 1: symbol frequencies ----
        each_run([&](uint8_t v, uint32_t S, uint32_t L, uint32_t a, uint32_t b) {
            df_run_tokens(S, L, a, b, [&](uint32_t kind, uint32_t x) {
                if (kind == 0) atomicAdd(&freq[v], x);
                else atomicAdd(&freq[257 + df_len_code(x)], 1u);
            });
        });
        if (t == 0) atomicAdd(&freq[256], 1u);
        __syncthreads();
        // ---- the codes (the whole workgroup) and the block header (one lane) ----
        df_code_lengths(freq, DF_NLIT, 15, len, A, order, fr2, ctl);
        df_codes(len, DF_NLIT, code, cnt16, nxt16);
        if (t == 0) {
            uint32_t hlit = DF_NLIT;
            while (hlit > 257 && len[hlit - 1] == 0) hlit--;
            // lengths of the hlit literal/length codes, then two distance codes of one bit each (only code 0, distance 1, is ever used;
            // two make the distance code complete), run-length coded with the code-length alphabet (16: repeat, 17 / 18: zeros)
            uint32_t nh = 0;
            auto lens_at = [&](uint32_t i) -> uint32_t { return i < hlit ? len[i] : 1u; };
            const uint32_t total = hlit + 2;
            for (uint32_t i = 0; i < total;) {
                const uint32_t v = lens_at(i);
                uint32_t r = 1;
                while (i + r < total && lens_at(i + r) == v) r++;
                if (v == 0 && r >= 3) {
                    uint32_t left = r;
                    while (left >= 3) {
                        const uint32_t k = left > 138 ? 138 : left;
                        if (k >= 11) { hdr_sym[nh] = 18; hdr_extra[nh++] = (uint8_t)(k - 11); } else { hdr_sym[nh] = 17; hdr_extra[nh++] = (uint8_t)(k - 3); }
                        left -= k;
                    }
                    for (; left; left--) { hdr_sym[nh] = 0; hdr_extra[nh++] = 0; }
                } else if (v != 0 && r >= 4) {
                    hdr_sym[nh] = (uint16_t)v; hdr_extra[nh++] = 0;
                    uint32_t left = r - 1;
                    while (left >= 3) { const uint32_t k = left > 6 ? 6 : left; hdr_sym[nh] = 16; hdr_extra[nh++] = (uint8_t)(k - 3); left -= k; }
                    for (; left; left--) { hdr_sym[nh] = (uint16_t)v; hdr_extra[nh++] = 0; }
                } else {
                    for (uint32_t k = 0; k < r; k++) { hdr_sym[nh] = (uint16_t)v; hdr_extra[nh++] = 0; }
                }
                i += r;
            }
            for (uint32_t i = 0; i < nh; i++) clfreq[hdr_sym[i]]++;
            sh_misc[2] = nh; sh_misc[3] = hlit;
        }
        __syncthreads();
        df_code_lengths(clfreq, DF_NCL, 7, cllen, A, order, fr2, ctl);
        df_codes(cllen, DF_NCL, clcode, cnt16, nxt16);
        if (t == 0) {
            const uint32_t nh = sh_misc[2], hlit = sh_misc[3];
            uint32_t hclen = DF_NCL;
            while (hclen > 4 && cllen[df_cl_order[hclen - 1]] == 0) hclen--;
            BitW bw{outw, 0};
            bw.put(0u, 1); bw.put(2u, 2);                         // BFINAL 0, BTYPE 10
            bw.put(hlit - 257, 5); bw.put(2 - 1, 5); bw.put(hclen - 4, 4);
            for (uint32_t i = 0; i < hclen; i++) bw.put(cllen[df_cl_order[i]], 3);
            for (uint32_t i = 0; i < nh; i++) {
                const uint32_t s = hdr_sym[i];
                bw.put(clcode[s], cllen[s]);
                if (s == 16) bw.put(hdr_extra[i], 2); else if (s == 17) bw.put(hdr_extra[i], 3); else if (s == 18) bw.put(hdr_extra[i], 7);
            }
            sh_misc[0] = (uint32_t)bw.pos;
        }
        __syncthreads();
        // ---- pass 2: bits per thread, prefix sum ----
        uint32_t mybits = 0;
        each_run([&](uint8_t v, uint32_t S, uint32_t L, uint32_t a, uint32_t b) {
            df_run_tokens(S, L, a, b, [&](uint32_t kind, uint32_t x) {
                if (kind == 0) mybits += x * len[v];
                else { const uint32_t lc = df_len_code(x); mybits += len[257 + lc] + df_len_extra[lc] + 1u; }
            });
        });
        {
            uint32_t inc = mybits;
            for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d); if ((t & 63) >= d) inc += o; }
            if ((t & 63) == 63) scan_sh[t >> 6] = inc;
            __syncthreads();
            uint32_t base = sh_misc[0];
            for (uint32_t w = 0; w < (t >> 6); w++) base += scan_sh[w];
            if (t == DF_T - 1) sh_misc[1] = base + inc + len[256];              // body and end-of-block
            mybits = base + inc - mybits;                                      // this thread's first bit
        }
        __syncthreads();
        const uint32_t total_bits = sh_misc[1];
        // the block in its dynamic form + the empty stored block that closes it on a byte boundary: 3 bits, padding, 00 00 FF FF
        const uint32_t dyn_bytes = (total_bits + 3 + 7) / 8 + 4;
        const bool stored = dyn_bytes >= n + 5;
        uint8_t* dst = out + c * (uint64_t)DF_OUT_STRIDE;
        if (!stored) {
            // ---- pass 3: the tokens' bits ----
            uint64_t pos = mybits;
            auto put = [&](uint32_t v, uint32_t nbits) {                         // other lanes write the same words: atomic OR
                const uint32_t i = (uint32_t)(pos >> 5), s = (uint32_t)(pos & 31);
                atomicOr(&outw[i], v << s);
                if (s + nbits > 32) atomicOr(&outw[i + 1], v >> (32 - s));
                pos += nbits;
            };
            each_run([&](uint8_t v, uint32_t S, uint32_t L, uint32_t a, uint32_t b) {
                df_run_tokens(S, L, a, b, [&](uint32_t kind, uint32_t x) {
                    if (kind == 0) { for (uint32_t i = 0; i < x; i++) put(code[v], len[v]); }
                    else {
                        const uint32_t lc = df_len_code(x);
                        put(code[257 + lc], len[257 + lc]);
                        if (df_len_extra[lc]) put(x - df_len_base[lc], df_len_extra[lc]);
                        put(0u, 1);                                              // distance code 0 (distance 1): the one-bit code 0
                    }
                });
            });
            __syncthreads();
            if (t == 0) {
                BitW bw{outw, (uint64_t)total_bits - len[256]};
                bw.put(code[256], len[256]);
                bw.put(0u, 3);                                                   // BFINAL 0, BTYPE 00: the empty stored block
                bw.pos = (bw.pos + 7) & ~7ull;
                bw.put(0xFFFF0000u, 32);
                sizes[c] = (uint32_t)(bw.pos >> 3);
            }
            __syncthreads();
            const uint32_t nbytes = dyn_bytes;
            for (uint32_t i = t; i < (nbytes + 3) / 4; i += DF_T) ((uint32_t*)dst)[i] = outw[i];
        } else {
            if (t == 0) {
                dst[0] = 0; dst[1] = (uint8_t)n; dst[2] = (uint8_t)(n >> 8); dst[3] = (uint8_t)~n; dst[4] = (uint8_t)(~n >> 8);
                sizes[c] = n + 5;
            }
            for (uint32_t i = t; i < n; i += DF_T) dst[5 + i] = tb[i];
        }
    }
}

// the read blocks' streams, contiguous: 78 01, the chunks, the final empty stored block; the Adler-32 is added on the host
__global__ void k_deflate_gather(const uint8_t* chunks, const uint32_t* sizes, const uint64_t* chunk_dst, uint64_t n_chunks, uint8_t* dst) {
    for (uint64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const uint8_t* s = chunks + c * (uint64_t)DF_OUT_STRIDE;
        uint8_t* d = dst + chunk_dst[c];
        const uint32_t n = sizes[c];
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
    }
}

// Device buffers kept from call to call (one set per process, taken under a lock): a call per batch of a file would otherwise
// allocate and free ~1.5 GB each time, and memory this process has freed comes back from hipMalloc at the driver's wiping
// rate (~43 GB/s), for this call and for whoever allocates next.  Large calls go through in slices, so the set stays small.
struct Grow {
    void* p = nullptr; size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 4 + 256;
        const hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    template <typename T> T* as() { return (T*)p; }
};
// One set PER DEVICE (it was one for the process, dropped and rebuilt whenever the device changed: callers on several GPUs serialised
// behind one mutex and thrashed 1.5 GB each way -- ADVICE r3).  A device's set is used by one call at a time.
struct DeflateScratch {
    std::mutex mu;
    int device = -1;
    Grow off, text, cb, cl, out, sizes, adler, cdst, fin;
    std::vector<uint8_t> h;
    void drop() { for (Grow* g : {&off, &text, &cb, &cl, &out, &sizes, &adler, &cdst, &fin}) { if (g->p) (void)hipFree(g->p); g->p = nullptr; g->cap = 0; } }
};
constexpr int DF_MAX_DEVICES = 64;
DeflateScratch* scratch_table() { static DeflateScratch* t = new DeflateScratch[DF_MAX_DEVICES]; return t; }     // (never destroyed: the HIP runtime may be gone by then)
DeflateScratch& scratch(int device_id) { return scratch_table()[device_id]; }

constexpr uint64_t DF_SLICE_TEXT = 512ull << 20;                // text bytes per slice of a large call

#define DCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_create_error(std::string(#call) + ": " + hipGetErrorString(e_)); return LEON_E_HIP; } } while (0)

// blocks [b0, b1) of the call: reads [r0, r1), whose qualities start at d_quals + (offsets[r0] - offsets[0])
int deflate_slice(DeflateScratch& S, int device_id, const uint8_t* d_quals, const uint64_t* offsets, uint64_t r0, uint64_t r1, uint32_t reads_per_block,
                  leon_block_sink sink, void* user, uint64_t first_block_id) {
    hipStream_t s = nullptr;
    static const bool trace = getenv("LEON_TRACE_DEFLATE") != nullptr;           // measurement aid: a slice's stages on stderr
    const auto t_start = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    double ms_tables = 0, ms_kernels = 0, ms_d2h = 0;
    const uint64_t n_reads = r1 - r0, n_blocks = (n_reads + reads_per_block - 1) / reads_per_block;
    const uint64_t* off = offsets + r0;
    const uint64_t q0 = off[0], n_q = off[n_reads] - q0, n_text = n_q + n_reads;
    const uint8_t* dq = d_quals + (q0 - offsets[0]);
    // chunk table: block b's text is [blk_text[b], blk_text[b + 1]) of the slice's text, cut in DF_CHUNK pieces
    std::vector<uint64_t> blk_text(n_blocks + 1), chunk_begin, blk_chunk0(n_blocks + 1);
    std::vector<uint32_t> chunk_len;
    for (uint64_t b = 0; b <= n_blocks; b++) {
        const uint64_t r = std::min<uint64_t>(n_reads, b * reads_per_block);
        blk_text[b] = off[r] - q0 + r;
    }
    for (uint64_t b = 0; b < n_blocks; b++) {
        blk_chunk0[b] = chunk_begin.size();
        for (uint64_t p = blk_text[b]; p < blk_text[b + 1]; p += DF_CHUNK) {
            chunk_begin.push_back(p);
            chunk_len.push_back((uint32_t)std::min<uint64_t>(DF_CHUNK, blk_text[b + 1] - p));
        }
    }
    blk_chunk0[n_blocks] = chunk_begin.size();
    const uint64_t n_chunks = chunk_begin.size();
    ms_tables = ms_since(t_start);
    const auto t_k = std::chrono::steady_clock::now();
    DCHK(S.off.ensure((n_reads + 1) * 8)); DCHK(S.text.ensure(n_text + 64));
    DCHK(staged_h2d(device_id, S.off.p, off, (n_reads + 1) * 8));
    hipLaunchKernelGGL(k_qual_text, dim3((uint32_t)std::min<uint64_t>((n_reads + 3) / 4, 1u << 16)), dim3(256), 0, s, dq, S.off.as<uint64_t>(), n_reads, q0,
                       S.text.as<uint8_t>());
    std::vector<uint32_t> sizes(n_chunks), adl(n_chunks);
    if (n_chunks) {
        DCHK(S.cb.ensure(n_chunks * 8)); DCHK(S.cl.ensure(n_chunks * 4)); DCHK(S.out.ensure(n_chunks * (uint64_t)DF_OUT_STRIDE));
        DCHK(S.sizes.ensure(n_chunks * 4)); DCHK(S.adler.ensure(n_chunks * 4)); DCHK(S.cdst.ensure(n_chunks * 8));
        DCHK(hipMemcpyAsync(S.cb.p, chunk_begin.data(), n_chunks * 8, hipMemcpyHostToDevice, s));
        DCHK(hipMemcpyAsync(S.cl.p, chunk_len.data(), n_chunks * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_deflate_chunks, dim3((uint32_t)std::min<uint64_t>(n_chunks, 1u << 20)), dim3(DF_T), 0, s, S.text.as<uint8_t>(), S.cb.as<uint64_t>(),
                           S.cl.as<uint32_t>(), n_chunks, S.out.as<uint8_t>(), S.sizes.as<uint32_t>(), S.adler.as<uint32_t>());
        DCHK(hipGetLastError());
        DCHK(hipMemcpy(sizes.data(), S.sizes.p, n_chunks * 4, hipMemcpyDeviceToHost));
        DCHK(hipMemcpy(adl.data(), S.adler.p, n_chunks * 4, hipMemcpyDeviceToHost));
    }
    // layout of the final streams: per block 2 bytes of zlib header, its chunks, 5 bytes of final stored block, 4 of Adler-32
    std::vector<uint64_t> blk_dst(n_blocks + 1, 0), chunk_dst(n_chunks);
    for (uint64_t b = 0; b < n_blocks; b++) {
        uint64_t at = blk_dst[b] + 2;
        for (uint64_t c = blk_chunk0[b]; c < blk_chunk0[b + 1]; c++) { chunk_dst[c] = at; at += sizes[c]; }
        blk_dst[b + 1] = at + 5 + 4;
    }
    const uint64_t total = blk_dst[n_blocks];
    if (S.h.size() < total) S.h.resize(total + total / 4);
    uint8_t* h = S.h.data();
    if (n_chunks) {
        DCHK(S.fin.ensure(total + 16));
        DCHK(hipMemcpyAsync(S.cdst.p, chunk_dst.data(), n_chunks * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_deflate_gather, dim3((uint32_t)std::min<uint64_t>(n_chunks, 1u << 20)), dim3(256), 0, s, S.out.as<uint8_t>(), S.sizes.as<uint32_t>(),
                           S.cdst.as<uint64_t>(), n_chunks, S.fin.as<uint8_t>());
        DCHK(hipGetLastError());
        DCHK(hipStreamSynchronize(s));
        ms_kernels = ms_since(t_k);
        const auto t_d = std::chrono::steady_clock::now();
        DCHK(staged_d2h(device_id, h, S.fin.p, total));
        ms_d2h = ms_since(t_d);
    }
    const auto t_sink = std::chrono::steady_clock::now();
    for (uint64_t b = 0; b < n_blocks; b++) {
        uint8_t* p = h + blk_dst[b];
        p[0] = 0x78; p[1] = 0x01;
        uLong ad = adler32(0L, Z_NULL, 0);
        for (uint64_t c = blk_chunk0[b]; c < blk_chunk0[b + 1]; c++) ad = adler32_combine(ad, adl[c], (z_off_t)chunk_len[c]);
        uint8_t* e = h + blk_dst[b + 1] - 9;
        e[0] = 1; e[1] = 0; e[2] = 0; e[3] = 0xFF; e[4] = 0xFF;                  // BFINAL 1, stored, empty
        e[5] = (uint8_t)(ad >> 24); e[6] = (uint8_t)(ad >> 16); e[7] = (uint8_t)(ad >> 8); e[8] = (uint8_t)ad;
        const uint32_t nr = (uint32_t)std::min<uint64_t>(reads_per_block, n_reads - b * reads_per_block);
        if (sink(user, first_block_id + b, p, blk_dst[b + 1] - blk_dst[b], nr)) { set_create_error("qual_deflate: block sink returned non-zero"); return LEON_E_SINK; }
    }
    if (trace) fprintf(stderr, "[leon deflate] %llu blocks, %.1f MB of text -> %.1f MB: tables %.1f ms, offsets + kernels %.1f, D2H %.1f, sink %.1f\n",
                       (unsigned long long)n_blocks, n_text / 1e6, total / 1e6, ms_tables, ms_kernels, ms_d2h, ms_since(t_sink));
    return LEON_OK;
}

}  // namespace
}  // namespace leon

using namespace leon;

extern "C" int leon_qual_deflate_blocks_device(int device_id, const uint8_t* d_quals, const uint64_t* offsets, uint64_t n_reads,
                                               uint32_t reads_per_block, leon_block_sink sink, void* user, uint64_t first_block_id) {
    if ((n_reads && (!d_quals || !offsets)) || !sink || !reads_per_block) { set_create_error("qual_deflate: null argument"); return LEON_E_INVALID; }
    if (!n_reads) return LEON_OK;
    for (uint64_t i = 0; i < n_reads; i++)
        if (offsets[i + 1] < offsets[i]) { set_create_error("qual_deflate: offsets are not monotonic"); return LEON_E_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) { set_create_error("qual_deflate: no such HIP device"); return LEON_E_NO_DEVICE; }
    if (device_id >= DF_MAX_DEVICES) { set_create_error("qual_deflate: device ordinal beyond the scratch table"); return LEON_E_INVALID; }
    DCHK(hipSetDevice(device_id));
    DeflateScratch& S = scratch(device_id);
    std::lock_guard<std::mutex> lock(S.mu);
    S.device = device_id;
    const uint64_t n_blocks = (n_reads + reads_per_block - 1) / reads_per_block;
    for (uint64_t b0 = 0; b0 < n_blocks;) {                      // slices of whole blocks, ~DF_SLICE_TEXT of text each
        uint64_t b1 = b0 + 1;
        const uint64_t r0 = b0 * reads_per_block;
        while (b1 < n_blocks && offsets[std::min<uint64_t>(n_reads, (b1 + 1) * reads_per_block)] - offsets[r0] <= DF_SLICE_TEXT) b1++;
        const uint64_t r1 = std::min<uint64_t>(n_reads, b1 * reads_per_block);
        const int rc = deflate_slice(S, device_id, d_quals, offsets, r0, r1, reads_per_block, sink, user, first_block_id + b0);
        if (rc) return rc;
        b0 = b1;
    }
    return LEON_OK;
}

/* the device buffers leon_qual_deflate_blocks_device keeps from call to call (about 1.5 GB after a large call) */
extern "C" void leon_qual_deflate_release(void) {
    int before = -1;
    const bool had_device = hipGetDevice(&before) == hipSuccess;
    for (int d = 0; d < DF_MAX_DEVICES; d++) {
        DeflateScratch& S = scratch(d);
        std::lock_guard<std::mutex> lock(S.mu);
        if (S.device >= 0 && hipSetDevice(S.device) == hipSuccess) S.drop();
        S.device = -1;
        std::vector<uint8_t>().swap(S.h);
    }
    if (had_device) (void)hipSetDevice(before);
}

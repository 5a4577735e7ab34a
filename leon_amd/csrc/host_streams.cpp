// host_streams.cpp -- the host-side entry points of the C-ABI that need no GPU:
//   * HeaderDecoder over read blocks (gatb HeaderCoder.cpp [RECALLED lo]; record layout: DESIGN.md section 1.3), blocks in
//     parallel on host threads: inside a block every header is rebuilt from the one before it, so a block is one serial
//     chain, and `-d` has as many chains as the file has blocks;
//   * the quality stream in its lossless form (`-lossless`): per read block, the qualities joined by '\n' through zlib
//     (upstream: DnaEncoder buffers the block's quality lines and deflates them, Leon::writeBlockLena [RECALLED med]).
// Nothing here is on the DNA encode path; these are the streams either side of it.
#include "../../include/leon_dna.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace leon { void set_create_error(const std::string& msg); }

namespace {

int fail(int code, const std::string& msg) { leon::set_create_error(msg); return code; }

// "all cores" = what this process may actually use: the container's CPU quota (cgroup v2 cpu.max, v1 cfs_quota) when it is
// below the number of logical CPUs -- 256 threads on a 16-CPU quota only get each other throttled
uint32_t usable_cpus() {
    uint32_t n = std::max(1u, std::thread::hardware_concurrency());
    long long quota = -1, period = -1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else {
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = -1; fclose(g); }
    }
    if (quota > 0 && period > 0) n = std::min<uint32_t>(n, (uint32_t)std::max<long long>(1, (quota + period - 1) / period));
    return n;
}
}  // namespace
namespace leon { uint32_t usable_cpus() { return ::usable_cpus(); } }       // (capi.hip sizes its host-chain pool by it)
namespace {

template <typename F> void parallel_blocks(uint64_t n_blocks, uint32_t n_threads, F&& f) {
    if (n_threads == 0) n_threads = usable_cpus();
    n_threads = (uint32_t)std::min<uint64_t>(n_threads, std::max<uint64_t>(n_blocks, 1));
    std::atomic<uint64_t> next{0};
    auto work = [&] { for (uint64_t b; (b = next.fetch_add(1)) < n_blocks;) f(b); };
    if (n_threads <= 1) { work(); return; }
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < n_threads; t++) th.emplace_back(work);
    for (auto& t : th) t.join();
}

// ---- Order0Model with the two-level layout of the device coder (F(x) = H[x >> 4] + Lw[x]): update touches <= 31 words ----
struct Model256 {
    uint32_t H[17], Lw[257];
    uint32_t size;
    void init(uint32_t n) { size = n; for (uint32_t x = 0; x <= 256; x++) Lw[x] = x < 256 ? (x & 15u) : 0u; for (uint32_t j = 0; j <= 16; j++) H[j] = 16 * j; }
    uint32_t F(uint32_t x) const { return H[x >> 4] + Lw[x]; }
    uint32_t total() const { return F(size); }
    uint32_t find(uint64_t value) const {                    // last c < size with F(c) <= value
        uint32_t j = 0;
        const uint32_t jmax = (size - 1) >> 4;
        while (j < jmax && H[j + 1] <= value) j++;
        uint32_t c = 16 * j;
        const uint32_t cmax = std::min(size - 1, 16 * j + 15);
        while (c < cmax && F(c + 1) <= value) c++;
        return c;
    }
    void update(uint32_t c) {                                // Order0Model::update: F(x) += 1 for x > c
        for (uint32_t x = c + 1; x <= (c | 15u); x++) Lw[x]++;
        for (uint32_t j = (c >> 4) + 1; j <= 16; j++) H[j]++;
    }
};
struct RangeDecoder {
    uint64_t low = 0, range = ~0ull, code = 0;
    const uint8_t* p; uint64_t n, i = 0;
    uint32_t past = 0;
    bool bad = false;                                         // the payload is not a stream this coder wrote (checked by the caller per header)
    RangeDecoder(const uint8_t* p_, uint64_t n_) : p(p_), n(n_) { for (int k = 0; k < 8; k++) code = (code << 8) | byte(); }
    uint8_t byte() { return i < n ? p[i++] : 0; }
    uint32_t next(Model256& m) {
        range /= m.total();
        if (range == 0 || bad) { bad = true; return 0; }
        const uint32_t c = m.find((code - low) / range);
        low += (uint64_t)m.F(c) * range;
        range *= (uint64_t)(m.F(c + 1) - m.F(c));
        while ((low ^ (low + range)) < (1ull << 56) || (range < (1ull << 48) && ((range = (0 - low) & ((1ull << 48) - 1)), true))) {
            code = (code << 8) | byte();
            range <<= 8;
            low <<= 8;
            if (i >= n && ++past > 16) { bad = true; return 0; }   // far past the end (a range of 0 would spin here)
        }
        m.update(c);
        return c;
    }
};

enum { H_END = 1, H_END_MATCH, H_FIELD_ASCII, H_FIELD_NUMERIC, H_FIELD_DELTA, H_FIELD_DELTA_2, H_FIELD_ZERO_ONLY, H_FIELD_ZERO_AND_NUMERIC, H_TYPE_COUNT };

struct HeaderModels {
    Model256 type, field_index, field_column, mis_size, ascii, zero, numeric[9];
    HeaderModels() {
        type.init(H_TYPE_COUNT); field_index.init(256); field_column.init(256); mis_size.init(256); ascii.init(256); zero.init(256);
        for (auto& m : numeric) m.init(256);
    }
};
// Where the header decoder's symbols come from: the block's range-coded payload (decoded here, on a host thread), or the
// block's symbols already decoded on the device (leon_header_decode_blocks: the arithmetic decoding is the expensive part
// of a header block and a serial chain per block -- the device runs all the chains at once --, the text is not).
enum ModelId { MI_TYPE, MI_FIELD_INDEX, MI_FIELD_COLUMN, MI_MIS_SIZE, MI_ASCII, MI_ZERO, MI_NUMERIC0 };   // MI_NUMERIC0 + i: byte-count model, then one per byte
struct RangeSource {
    RangeDecoder d;
    HeaderModels* M;
    RangeSource(const uint8_t* p, uint64_t n) : d(p, n), M(new HeaderModels()) {}
    ~RangeSource() { delete M; }
    RangeSource(const RangeSource&) = delete;
    bool bad() const { return d.bad; }
    uint32_t next(uint32_t id) {
        Model256& m = id == MI_TYPE ? M->type : id == MI_FIELD_INDEX ? M->field_index : id == MI_FIELD_COLUMN ? M->field_column
                    : id == MI_MIS_SIZE ? M->mis_size : id == MI_ASCII ? M->ascii : id == MI_ZERO ? M->zero : M->numeric[id - MI_NUMERIC0];
        return d.next(m);
    }
};
struct SymbolSource {
    const uint8_t* p; uint64_t n, i = 0;
    bool over = false;
    SymbolSource(const uint8_t* p_, uint64_t n_) : p(p_), n(n_) {}
    bool bad() const { return over; }
    uint32_t next(uint32_t) { if (i < n) return p[i++]; over = true; return 0; }
};
template <typename S> uint64_t decode_numeric(S& d) {
    uint32_t bc = d.next(MI_NUMERIC0);
    if (bc > 8) bc = 8;
    uint64_t v = 0;
    for (uint32_t i = 0; i < bc; i++) v |= (uint64_t)d.next(MI_NUMERIC0 + 1 + i) << (8 * i);
    return v;
}
template <typename S> uint64_t decode_count(S& d, uint32_t id) {
    const uint64_t x = d.next(id);
    return x < 255 ? x : 255 + decode_numeric(d);
}
inline bool is_alnum(uint8_t c) { return (uint8_t)(c - '0') < 10 || (uint8_t)((c | 32) - 'a') < 26; }
struct Field { uint64_t len = 0; bool numeric = false, has_sep = false; uint8_t sep = 0; uint64_t value = 0; };
// the field of h starting at pos: token + one separator; numeric = digits without a leading zero, 1..18 of them
Field field_at(const uint8_t* h, uint64_t len, uint64_t pos) {
    Field f;
    uint64_t e = pos;
    bool digits = true;
    while (e < len && is_alnum(h[e])) { digits = digits && (uint8_t)(h[e] - '0') < 10; e++; }
    const uint64_t tok = e - pos;
    if (e < len) { f.has_sep = true; f.sep = h[e]; e++; }
    f.len = e - pos;
    if (digits && tok >= 1 && tok <= 18 && (tok == 1 || h[pos] != '0') && !(f.has_sep && f.sep == 0)) {
        f.numeric = true;
        for (uint64_t i = pos; i < pos + tok; i++) f.value = f.value * 10 + (uint64_t)(h[i] - '0');
    }
    return f;
}

// one block; out receives the headers back to back, off[n + 1] (relative to out); false when the payload does not decode
template <typename S>
bool decode_header_block(S& d, uint32_t n, const uint8_t* first, uint64_t first_len, std::string& out, std::vector<uint64_t>& off) {
    out.clear(); off.assign(1, 0);
    std::string prev(reinterpret_cast<const char*>(first), first_len), cur;
    const uint64_t max_header = 1ull << 31;
    for (uint32_t r = 0; r < n; r++) {
        cur.clear();
        const uint8_t* ph = reinterpret_cast<const uint8_t*>(prev.data());
        const uint64_t lp = prev.size();
        uint64_t pp = 0, nf = 0;
        auto copy_prev_until = [&](uint64_t limit) {             // fields nf .. limit-1 are the previous header's
            while (nf < limit && pp < lp) { const Field p = field_at(ph, lp, pp); cur.append(prev, pp, p.len); pp += p.len; nf++; }
        };
        for (;;) {
            const uint32_t t = d.next(MI_TYPE);
            if (d.bad()) return false;
            if (t == H_END_MATCH) { copy_prev_until(~0ull); break; }
            if (t == H_END) {
                const uint64_t f = decode_count(d, MI_FIELD_INDEX);
                if (f < nf) return false;
                copy_prev_until(f);
                if (nf != f) return false;
                break;
            }
            if (t < H_FIELD_ASCII || t >= H_TYPE_COUNT) return false;
            const uint64_t idx = decode_count(d, MI_FIELD_INDEX);
            if (idx < nf) return false;
            copy_prev_until(idx);
            if (nf != idx) return false;
            Field p;
            const bool have_p = pp < lp;
            const uint64_t p0 = pp;
            if (have_p) { p = field_at(ph, lp, pp); pp += p.len; }
            if (t == H_FIELD_ASCII) {
                const uint64_t col = decode_count(d, MI_FIELD_COLUMN), sz = decode_count(d, MI_MIS_SIZE);
                if (col > p.len || sz > max_header || cur.size() + col + sz > max_header) return false;
                cur.append(prev, p0, col);
                for (uint64_t j = 0; j < sz && !d.bad(); j++) cur.push_back((char)d.next(MI_ASCII));
            } else {
                uint64_t v = 0, z = 0; uint8_t sep = 0; bool has_sep = false;
                if (t == H_FIELD_DELTA || t == H_FIELD_DELTA_2) {
                    const uint64_t dv = decode_numeric(d);
                    if (!have_p || !p.numeric) return false;
                    v = t == H_FIELD_DELTA ? p.value + dv : p.value - dv;
                    sep = p.sep; has_sep = p.has_sep;
                } else {
                    if (t != H_FIELD_NUMERIC) z = decode_count(d, MI_ZERO);
                    if (t != H_FIELD_ZERO_ONLY) v = decode_numeric(d);
                    sep = (uint8_t)d.next(MI_ASCII); has_sep = sep != 0;
                }
                if (z > max_header || cur.size() + z > max_header) return false;
                cur.append(z, '0');
                if (t != H_FIELD_ZERO_ONLY) cur += std::to_string(v);
                if (has_sep) cur.push_back((char)sep);
            }
            nf++;
        }
        out += cur;
        off.push_back(out.size());
        prev.swap(cur);
    }
    return true;
}

// blocks in parallel on host threads, then the texts back to back with the offsets of every header
template <typename F>
int header_blocks_common(uint64_t n_blocks, uint8_t* out, uint64_t out_cap, uint64_t* out_off, uint64_t* out_size, uint32_t n_threads, F&& decode_one) {
    std::vector<std::string> texts(n_blocks);
    std::vector<std::vector<uint64_t>> offs(n_blocks);
    std::atomic<int64_t> bad{-1};
    parallel_blocks(n_blocks, n_threads, [&](uint64_t b) {
        if (!decode_one(b, texts[b], offs[b])) {
            int64_t expect = -1;
            bad.compare_exchange_strong(expect, (int64_t)b);
        }
    });
    if (bad.load() >= 0) return fail(LEON_E_INVALID, "header block " + std::to_string(bad.load()) + " does not decode");
    uint64_t w = 0;
    for (uint64_t b = 0; b < n_blocks; b++) w += texts[b].size();
    *out_size = w;
    if (w > out_cap || !out) return fail(LEON_E_OVERFLOW, "header output needs " + std::to_string(w) + " bytes");
    // every block's text and offsets to their final places, again by all threads (gigabytes: not a job for one core)
    std::vector<uint64_t> w0(n_blocks + 1, 0), r0(n_blocks + 1, 0);
    for (uint64_t b = 0; b < n_blocks; b++) { w0[b + 1] = w0[b] + texts[b].size(); r0[b + 1] = r0[b] + (offs[b].empty() ? 0 : offs[b].size() - 1); }
    out_off[0] = 0;
    parallel_blocks(n_blocks, n_threads, [&](uint64_t b) {
        if (!texts[b].empty()) memcpy(out + w0[b], texts[b].data(), texts[b].size());
        for (size_t i = 1; i < offs[b].size(); i++) out_off[r0[b] + i] = w0[b] + offs[b][i];
        std::string().swap(texts[b]);
    });
    return LEON_OK;
}

}  // namespace

namespace leon {
// the text of header blocks whose symbols the device has decoded (capi.hip, leon_header_decode_blocks): block b's symbols are
// syms[sym_begin[b] .. + sym_count[b]), one byte each, in the order the HeaderDecoder asks for them
int header_blocks_from_symbols(const uint8_t* syms, const uint64_t* sym_begin, const uint64_t* sym_count, const uint32_t* block_n_reads, uint64_t n_blocks,
                               const uint8_t* first_header, uint64_t first_header_len, uint8_t* out, uint64_t out_cap, uint64_t* out_off,
                               uint64_t* out_size, uint32_t n_threads) {
    return header_blocks_common(n_blocks, out, out_cap, out_off, out_size, n_threads, [&](uint64_t b, std::string& text, std::vector<uint64_t>& off) {
        SymbolSource src(syms + sym_begin[b], sym_count[b]);
        return decode_header_block(src, block_n_reads[b], first_header, first_header_len, text, off) && !src.over && src.i == src.n;
    });
}
}  // namespace leon

extern "C" {

int leon_host_header_decode_blocks(const uint8_t* payloads, const uint64_t* payload_off, const uint32_t* block_n_reads, uint64_t n_blocks,
                                   const uint8_t* first_header, uint64_t first_header_len, uint8_t* out, uint64_t out_cap,
                                   uint64_t* out_off, uint64_t* out_size, uint32_t n_threads) {
    if (!out_size || (n_blocks && (!payloads || !payload_off || !block_n_reads || !out_off)) || (!first_header && first_header_len))
        return fail(LEON_E_INVALID, "null argument");
    *out_size = 0;
    if (!n_blocks) return LEON_OK;
    for (uint64_t b = 0; b < n_blocks; b++)
        if (payload_off[b + 1] < payload_off[b]) return fail(LEON_E_INVALID, "payload offsets are not monotonic");
    return header_blocks_common(n_blocks, out, out_cap, out_off, out_size, n_threads, [&](uint64_t b, std::string& text, std::vector<uint64_t>& off) {
        RangeSource src(payloads + payload_off[b], payload_off[b + 1] - payload_off[b]);
        return decode_header_block(src, block_n_reads[b], first_header, first_header_len, text, off);
    });
}

// ---- quality stream, lossless: one zlib stream per read block over the block's quality lines, each followed by '\n' ----
int leon_host_qual_encode_blocks(const uint8_t* quals, const uint64_t* offsets, uint64_t n_reads, uint32_t reads_per_block,
                                 int zlib_level, uint32_t n_threads, leon_block_sink sink, void* user, uint64_t first_block_id) {
    if ((n_reads && (!quals || !offsets)) || !sink || !reads_per_block) return fail(LEON_E_INVALID, "null argument");
    if (zlib_level < -1 || zlib_level > 9) return fail(LEON_E_INVALID, "zlib level must be in -1..9");
    const uint64_t n_blocks = (n_reads + reads_per_block - 1) / reads_per_block;
    for (uint64_t i = 0; i < n_reads; i++)
        if (offsets[i + 1] < offsets[i]) return fail(LEON_E_INVALID, "offsets are not monotonic");
    std::vector<std::vector<uint8_t>> outs(n_blocks);
    std::atomic<int> zerr{0};
    parallel_blocks(n_blocks, n_threads, [&](uint64_t b) {
        const uint64_t r0 = b * reads_per_block, r1 = std::min<uint64_t>(n_reads, r0 + reads_per_block);
        std::vector<uint8_t> text;
        text.reserve((size_t)(offsets[r1] - offsets[r0] + (r1 - r0)));
        for (uint64_t r = r0; r < r1; r++) {
            text.insert(text.end(), quals + offsets[r], quals + offsets[r + 1]);
            text.push_back('\n');
        }
        uLongf cap = compressBound((uLong)text.size());
        outs[b].resize(cap);
        if (compress2(outs[b].data(), &cap, text.data(), (uLong)text.size(), zlib_level) != Z_OK) zerr.store(1);
        outs[b].resize(cap);
    });
    if (zerr.load()) return fail(LEON_E_INVALID, "zlib compress2 failed");
    for (uint64_t b = 0; b < n_blocks; b++) {
        const uint32_t nr = (uint32_t)std::min<uint64_t>(reads_per_block, n_reads - b * reads_per_block);
        if (sink(user, first_block_id + b, outs[b].data(), outs[b].size(), nr)) return fail(LEON_E_SINK, "block sink returned non-zero");
    }
    return LEON_OK;
}

int leon_host_qual_decode_blocks(const uint8_t* payloads, const uint64_t* payload_off, const uint32_t* block_n_reads,
                                 const uint64_t* block_n_bytes, uint64_t n_blocks, uint8_t* out, uint64_t out_cap, uint64_t* out_off,
                                 uint32_t n_threads) {
    if (n_blocks && (!payloads || !payload_off || !block_n_reads || !block_n_bytes || !out || !out_off)) return fail(LEON_E_INVALID, "null argument");
    if (!n_blocks) return LEON_OK;
    // block_n_bytes: quality bytes per block WITHOUT the newlines (what the container's block table records)
    std::vector<uint64_t> o0(n_blocks + 1, 0), r0(n_blocks + 1, 0);
    for (uint64_t b = 0; b < n_blocks; b++) {
        if (payload_off[b + 1] < payload_off[b]) return fail(LEON_E_INVALID, "payload offsets are not monotonic");
        o0[b + 1] = o0[b] + block_n_bytes[b];
        r0[b + 1] = r0[b] + block_n_reads[b];
        if (o0[b + 1] < o0[b] || o0[b + 1] > out_cap) return fail(LEON_E_INVALID, "output capacity below the sum of block_n_bytes");
    }
    std::atomic<int64_t> bad{-1};
    parallel_blocks(n_blocks, n_threads, [&](uint64_t b) {
        std::vector<uint8_t> text((size_t)(block_n_bytes[b] + block_n_reads[b]));
        uLongf got = (uLongf)text.size();
        bool ok = uncompress(text.data(), &got, payloads + payload_off[b], (uLong)(payload_off[b + 1] - payload_off[b])) == Z_OK && got == text.size();
        uint64_t w = o0[b], r = r0[b], at = 0;
        if (ok && b == 0) out_off[0] = 0;
        for (uint32_t i = 0; ok && i < block_n_reads[b]; i++) {
            const uint8_t* nl = (const uint8_t*)memchr(text.data() + at, '\n', text.size() - at);
            if (!nl) { ok = false; break; }
            const uint64_t len = (uint64_t)(nl - (text.data() + at));
            if (w + len > o0[b + 1]) { ok = false; break; }
            memcpy(out + w, text.data() + at, len);
            w += len; at += len + 1;
            out_off[++r] = w;
        }
        if (ok && (w != o0[b + 1] || at != text.size())) ok = false;
        if (!ok) { int64_t expect = -1; bad.compare_exchange_strong(expect, (int64_t)b); }
    });
    if (bad.load() >= 0) return fail(LEON_E_INVALID, "quality block " + std::to_string(bad.load()) + " does not decode");
    return LEON_OK;
}

}  // extern "C"

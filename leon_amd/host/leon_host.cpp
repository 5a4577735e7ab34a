// leon_host.cpp -- see leon_host.hpp.  Host glue only: the DNA and header streams come from libleon_dna.so's HIP path.
//
// -c, one pass over the input (bank.hpp), batch by batch:
//      headers  -> leon_header_encode_batch            -> leon/header/block_<i>
//      bases    -> appended to a device-resident copy of the reads (never all of it on the host)
//      qualities (-lossless) -> leon_host_qual_encode_blocks (zlib on host threads) -> leon/qual/block_<i>
//    then: solid k-mers of the resident reads (leon_kmer_solid_device, automatic threshold unless -abundance) -> bloom ->
//      leon_dna_encode_batch_device over the resident reads -> leon/dna/block_<i>, anchor dictionary, bloom, tables;
//      lossy qualities (the default) need the bloom: a second pass over the file smooths them on the device
//      (leon_qual_smooth_batch_device) before the same zlib blocks.
// -d, block group by block group: leon_dna_decode_blocks (device), leon_host_header_decode_blocks,
//      leon_host_qual_decode_blocks, records written as FASTA / FASTQ text.
#include "leon_host.hpp"

#include "bank.hpp"
#include "leon_container.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <deque>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <future>
#include <iostream>
#include <memory>
#include <mutex>
#include <thread>

// threads this process may keep busy: the logical CPUs, capped by the container's CPU quota (cgroup v2 cpu.max)
static uint64_t usableCpus() {
    uint64_t n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        long long period = 0;
        if (fscanf(f, "%63s %lld", q, &period) == 2 && std::string(q) != "max" && period > 0)
            n = std::min<uint64_t>(n, (uint64_t)std::max<long long>(1, (atoll(q) + period - 1) / period));
        fclose(f);
    }
    return n;
}

namespace leon_host {

const char* Leon::STR_COMPRESS = "-c";
const char* Leon::STR_DECOMPRESS = "-d";

namespace {

using namespace layout;

void check(leon_dna_ctx* ctx, int rc, const char* what) {
    if (rc != LEON_OK) throw Exception(std::string(what) + ": " + leon_last_error(ctx));
}
bool ends_with(const std::string& s, const std::string& suffix) {
    return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}
double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

struct CtxDeleter { void operator()(leon_dna_ctx* c) const { if (c) leon_dna_ctx_destroy(c); } };
typedef std::unique_ptr<leon_dna_ctx, CtxDeleter> CtxPtr;

CtxPtr make_ctx(uint32_t k, uint64_t tai, int device, uint32_t n_hash = 7, uint32_t block_nbits = 12, uint32_t rpb = Leon::READ_PER_BLOCK) {
    leon_dna_cfg cfg = {};
    cfg.struct_size = sizeof(cfg);
    cfg.kmer_size = k; cfg.reads_per_block = rpb;
    cfg.bloom_n_hash = n_hash; cfg.bloom_block_nbits = block_nbits; cfg.bloom_tai = tai; cfg.device_id = device;
    leon_dna_ctx* c = nullptr;
    check(nullptr, leon_dna_ctx_create(&cfg, &c), "leon_dna_ctx_create");
    return CtxPtr(c);
}

// One stream's blocks on their way into the container (Leon::writeBlock / writeBlockLena).  Blocks of one stream come from
// one thread in increasing id, or -- DNA blocks of a multi-GPU run -- from one thread per GPU with disjoint ids.
struct StreamWriter {
    Container* out = nullptr;
    std::mutex* mu = nullptr;
    const char* group = nullptr;
    std::vector<uint64_t> sizes, reads;          // per block id
    uint64_t bytes = 0;
    std::string error;
    static int sink(void* user, uint64_t block_id, const uint8_t* payload, uint64_t size, uint32_t n_reads) {
        StreamWriter* w = static_cast<StreamWriter*>(user);
        try {
            std::lock_guard<std::mutex> g(*w->mu);
            if (block_id >= w->sizes.size()) { w->sizes.resize(block_id + 1, ~0ull); w->reads.resize(block_id + 1, 0); }
            if (w->sizes[block_id] != ~0ull) throw Exception("block " + std::to_string(block_id) + " arrived twice");
            w->out->putBytes(Container::blockPath(w->group, block_id), payload, size);
            w->sizes[block_id] = size; w->reads[block_id] = n_reads; w->bytes += size;
            return 0;
        } catch (const std::exception& e) {        // no exception may cross the C boundary
            w->error = e.what();
            return 1;
        }
    }
    void complete(uint64_t n_blocks, const char* what) const {
        if (sizes.size() != n_blocks) throw Exception(std::string(what) + ": " + std::to_string(sizes.size()) + " blocks written, " + std::to_string(n_blocks) + " expected");
        for (uint64_t s : sizes) if (s == ~0ull) throw Exception(std::string(what) + ": a block is missing");
    }
};
void check_sink(leon_dna_ctx* ctx, int rc, const StreamWriter& w, const char* what) {
    if (rc == LEON_E_SINK && !w.error.empty()) throw Exception(std::string(what) + ": " + w.error);
    check(ctx, rc, what);
}

// the reads' bases, resident in HBM on one device: appended batch by batch, offsets kept on the host until the pass ends
struct DeviceReads {
    int device = 0;
    uint8_t* d_bases = nullptr;
    uint64_t* d_off = nullptr;
    uint64_t cap = 0, n_bases = 0;
    ~DeviceReads() { leon_device_free(d_bases); leon_device_free(d_off); }
    void reserve(uint64_t need) {
        if (need <= cap) return;
        const uint64_t nc = std::max<uint64_t>(need + need / 4, 1ull << 26);
        void* p = nullptr;
        check(nullptr, leon_device_alloc(device, nc + 64, &p), "leon_device_alloc");
        if (n_bases) check(nullptr, leon_device_copy(device, p, d_bases, n_bases), "leon_device_copy");
        leon_device_free(d_bases);
        d_bases = static_cast<uint8_t*>(p); cap = nc;
    }
    void append(const std::string& bases) {
        reserve(n_bases + bases.size());
        check(nullptr, leon_device_upload(device, d_bases + n_bases, bases.data(), bases.size()), "leon_device_upload");
        n_bases += bases.size();
    }
    void set_offsets(const std::vector<uint64_t>& off) {
        void* p = nullptr;
        check(nullptr, leon_device_alloc(device, off.size() * 8, &p), "leon_device_alloc");
        d_off = static_cast<uint64_t*>(p);
        check(nullptr, leon_device_upload(device, d_off, off.data(), off.size() * 8), "leon_device_upload");
    }
};

int device_for(int gpu_index) {
    // LEON_SHARE_GPU=1 lets a multi-GPU run rehearse on fewer devices (ranks wrap around); otherwise -gpus must fit
    static const char* share = getenv("LEON_SHARE_GPU");
    static const int n_dev = [] { int n = 0; return leon_device_count(&n) == LEON_OK ? n : 0; }();
    if (n_dev <= 0) throw Exception("no HIP device: the DNA encode path has no CPU fallback");
    if (gpu_index >= n_dev && !(share && share[0] == '1')) throw Exception("-gpus " + std::to_string(gpu_index + 1) + " asked, " + std::to_string(n_dev) + " device(s) present");
    return gpu_index % n_dev;
}


// Which encoder writes the quality blocks.  The default is zlib itself on the host threads (leon_host_qual_encode_blocks: compress2 at
// its default level, the bytes upstream writes [RECALLED] and the one stream of the file a third-party library can pin byte for byte).
// `-qual-deflate device` hands them to the device's deflate (leon_qual_deflate_blocks_device), which looks for runs only -- zlib's
// Z_RLE strategy -- and is tens of times faster than zlib's default strategy on all the host's cores; its blocks inflate to the same
// text but are not zlib's bytes.  `-qual-deflate auto` lets a sample decide (the first reads' lines, up to 256 KB, through zlib both
// ways: the device takes the stream unless that would cost more than 2 % -- where the lines resemble one another the default
// strategy's matches against earlier lines win).  Without the flag, LEON_QUAL_DEFLATE=auto|device|host in the environment is the
// tests' hook for the same choice.  The container records who wrote the blocks (layout::P_QUAL_ENCODER).
enum class QualEncoder { Auto, Device, Host };
QualEncoder qual_encoder_choice(const std::string& flag) {
    const char* e = flag.empty() ? getenv("LEON_QUAL_DEFLATE") : flag.c_str();
    const char* what = flag.empty() ? "LEON_QUAL_DEFLATE=" : "option -qual-deflate: ";
    if (!e || !*e || !strcmp(e, "host")) return QualEncoder::Host;
    if (!strcmp(e, "auto")) return QualEncoder::Auto;
    if (!strcmp(e, "device")) return QualEncoder::Device;
    throw Exception(std::string(what) + "'" + e + "': expected host, device or auto");
}
size_t deflated_size(const std::string& text, int strategy) {
    z_stream z{};
    if (deflateInit2(&z, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15, 8, strategy) != Z_OK) return 0;
    std::vector<uint8_t> out(deflateBound(&z, (uLong)text.size()));
    z.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(text.data())); z.avail_in = (uInt)text.size();
    z.next_out = out.data(); z.avail_out = (uInt)out.size();
    const int rc = deflate(&z, Z_FINISH);
    const size_t n = rc == Z_STREAM_END ? (size_t)z.total_out : 0;
    deflateEnd(&z);
    return n;
}
bool runs_are_enough(const char* quals, const uint64_t* off, uint64_t n_reads) {
    std::string text;
    for (uint64_t r = 0; r < n_reads && text.size() < (256u << 10); r++) { text.append(quals + (off[r] - off[0]), off[r + 1] - off[r]); text.push_back('\n'); }
    if (text.size() < 1024) return true;
    const size_t d = deflated_size(text, Z_DEFAULT_STRATEGY), r = deflated_size(text, Z_RLE);
    return d && r && (double)r <= 1.02 * (double)d;
}
}  // namespace

// ------------------------------------------------------------------------------------------------ Leon
Leon::Leon() {}
Leon::~Leon() {}

void Leon::run(int argc, char* argv[]) {
    try {
        for (int i = 1; i < argc; i++) {
            std::string a = argv[i];
            auto need = [&](const char* flag) -> std::string {
                if (i + 1 >= argc) throw Exception(std::string("option ") + flag + " needs a value");
                return argv[++i];
            };
            auto number = [&](const char* flag) -> long {
                const std::string v = need(flag);
                size_t used = 0;
                long x = 0;
                try { x = std::stol(v, &used); } catch (const std::exception&) { used = 0; }
                if (used != v.size() || v.empty()) throw Exception(std::string("option ") + flag + ": '" + v + "' is not a number");
                return x;
            };
            if (a == "-file") _inputFilename = need("-file");
            else if (a == STR_COMPRESS) _compress = true;
            else if (a == STR_DECOMPRESS) _decompress = true;
            else if (a == "-kmer-size") _kmerSize = (size_t)number("-kmer-size");
            else if (a == "-abundance") { _abundance = (int)number("-abundance"); if (_abundance < 1) throw Exception("-abundance must be at least 1"); }
            else if (a == "-nb-cores") _nbCores = (int)std::max<long>(0, number("-nb-cores"));
            else if (a == "-gpus") { _gpus = (int)number("-gpus"); if (_gpus < 1 || _gpus > 64) throw Exception("-gpus must be in 1..64"); }
            else if (a == "-verbose") _verbose = number("-verbose") != 0;
            else if (a == "-lossless") _lossless = true;
            else if (a == "-seq-only") _seqOnly = true;
            else if (a == "-noheader") _noHeader = true;
            else if (a == "-noqual") _noQual = true;
            else if (a == "-test-file") _testFile = true;
            else if (a == "-qual-deflate") { _qualDeflate = need("-qual-deflate"); (void)qual_encoder_choice(_qualDeflate); }
            else throw Exception("unknown option " + a);
        }
        if (_inputFilename.empty()) throw Exception("option -file is mandatory");
        if (_compress == _decompress) throw Exception("choose one of -c (compress) or -d (decompress)");
        if (_seqOnly) _noHeader = _noQual = true;             // "same as -noheader -noqual", /root/reference/README.md:56
        execute();
    } catch (const Exception&) {
        throw;
    } catch (const std::exception& e) {                        // bad_alloc and friends follow the EXCEPTION: contract too
        throw Exception(e.what());
    }
}

void Leon::execute() {
    if (_compress) executeCompression(); else executeDecompression();
}

// ------------------------------------------------------------------------------------------------ -c
void Leon::executeCompression() {
    if (_kmerSize < 3 || _kmerSize > 63) throw Exception("-kmer-size must be in 3..63");
    const auto t_start = std::chrono::steady_clock::now();
    const uint32_t k = (uint32_t)_kmerSize, rpb = READ_PER_BLOCK;
    Bank bank(_inputFilename);
    const bool fastq = bank.isFastq();
    const bool keep_header = !_noHeader, keep_qual = fastq && !_noQual;
    // X.fastq.gz -> X.fastq.leon, data/toy.fasta -> data/toy.fasta.leon (/root/reference/scripts/simple_test.sh:51,54, INSTALL:21-23)
    std::string stem = _inputFilename;
    if (ends_with(stem, ".gz")) stem.resize(stem.size() - 3);
    _outputFilename = stem + ".leon";
    const std::string tmp_name = _outputFilename + ".tmp";     // renamed at the very end: a failed run leaves no .leon behind
    struct TmpGuard { std::string p; bool keep = false; ~TmpGuard() { if (!keep) std::remove(p.c_str()); } } guard{tmp_name};
    Container out(tmp_name, Container::CREATE);
    std::mutex out_mu;
    StreamWriter wh, wd, wq;
    wh.out = wd.out = wq.out = &out; wh.mu = wd.mu = wq.mu = &out_mu;
    wh.group = GROUP_HEADER; wd.group = GROUP_DNA; wq.group = GROUP_QUAL;

    // ---- the pass over the file ----
    const int n_gpus = _gpus;
    std::vector<std::unique_ptr<DeviceReads>> store;
    for (int g = 0; g < n_gpus; g++) { store.emplace_back(new DeviceReads()); store.back()->device = device_for(g); }
    CtxPtr hdr_ctx;
    if (keep_header) hdr_ctx = make_ctx(k, 1000, store[0]->device);
    std::vector<uint64_t> offsets(1, 0);                         // base offsets of every read, absolute in the resident copy
    std::string first_header;
    uint64_t n_reads = 0, header_bytes = 0, qual_bytes = 0;
    std::vector<uint64_t> hdr_text;                              // bytes of header text per read block: the decoder sizes its buffers from it
    const uint64_t batch_reads = 64ull * rpb;
    ReadBatch batch;
    const QualEncoder qual_enc = qual_encoder_choice(_qualDeflate);
    bool qual_on_device = false;
    void* d_qbuf = nullptr; uint64_t d_qbuf_cap = 0;             // the batch's qualities on the device, for its deflate (one job at a time uses it)
    struct QBufGuard { void** p; ~QBufGuard() { leon_device_free(*p); } } qbuf_guard{&d_qbuf};
    // Lossy qualities need the bloom, which needs the whole file: they stay resident on device 0 (one byte per base, indexed
    // like the bases) until then, unless the file is too large for that (LEON_QUAL_RESIDENT_MB, default 64 GB): then a
    // second pass over the file feeds them through.
    std::unique_ptr<DeviceReads> qstore;
    uint64_t qual_resident_max = 64ull << 30;
    if (const char* e = getenv("LEON_QUAL_RESIDENT_MB")) qual_resident_max = (uint64_t)std::max<long long>(0, atoll(e)) << 20;
    if (fastq && !_noQual && !_lossless && qual_resident_max) { qstore.reset(new DeviceReads()); qstore->device = store[0]->device; }
    const bool qstore_used = qstore != nullptr;
    // the reader runs one batch ahead on its own thread: batch i + 1 is parsed while batch i's headers are coded on the device
    // and its bases (and qualities) cross to it
    ReadBatch buffers[2];
    std::future<void> qual_job;                                  // (declared after what it reads -- these buffers, d_qbuf -- so that it is joined before they go)
    auto parse_next = [&bank, batch_reads](ReadBatch* b) -> uint64_t { b->clear(); return bank.next(*b, batch_reads); };
    std::future<uint64_t> parsing = std::async(std::launch::async, parse_next, &buffers[0]);
    struct ParseJoin { std::future<uint64_t>& f; ~ParseJoin() { if (f.valid()) { try { f.get(); } catch (...) {} } } } parse_join{parsing};   // (never left running over dead buffers)
    double w_reader = 0, w_headers = 0, w_quals = 0, w_uploads = 0;   // where the pass's own thread spent its time (-verbose)
    for (uint32_t cur = 0;; cur ^= 1) {
        auto t_w = std::chrono::steady_clock::now();
        const uint64_t got = parsing.get();                      // (the reader's exceptions surface here)
        w_reader += seconds_since(t_w);
        if (!got) break;
        ReadBatch& batch = buffers[cur];
        // the quality blocks of the batch before read this buffer's twin in place: they are done (they started a whole batch ago)
        // before the reader is allowed to fill it again
        t_w = std::chrono::steady_clock::now();
        if (qual_job.valid()) qual_job.get();
        w_quals += seconds_since(t_w);
        if (got == batch_reads) parsing = std::async(std::launch::async, parse_next, &buffers[cur ^ 1]);
        if (n_reads == 0) first_header.assign(batch.headers, 0, batch.header_off[1]);
        t_w = std::chrono::steady_clock::now();
        if (keep_header) {
            check_sink(hdr_ctx.get(), leon_header_encode_batch(hdr_ctx.get(), reinterpret_cast<const uint8_t*>(batch.headers.data()), batch.header_off.data(), got, n_reads,
                                                               reinterpret_cast<const uint8_t*>(first_header.data()), first_header.size(), StreamWriter::sink, &wh),
                       wh, "leon_header_encode_batch");
            header_bytes += batch.headers.size();
            for (uint64_t r = 0; r < got; r += rpb) hdr_text.push_back(batch.header_off[std::min<uint64_t>(got, r + rpb)] - batch.header_off[r]);   // (batches are whole blocks but the last)
        }
        w_headers += seconds_since(t_w);
        qual_bytes += batch.quals.size();
        t_w = std::chrono::steady_clock::now();
        if (keep_qual && _lossless) {                            // deflated while the next batch is being parsed: on the device, or on the host threads
            if (n_reads == 0) qual_on_device = qual_enc == QualEncoder::Device || (qual_enc == QualEncoder::Auto && runs_are_enough(batch.quals.data(), batch.qual_off.data(), got));
            // (read in place: moving the strings out would make the reader allocate and fault in half a gigabyte per batch -- 0.23 s
            // of every batch at 100 M reads when this path did)
            const std::string* quals = &batch.quals;
            const std::vector<uint64_t>* qoff = &batch.qual_off;
            const uint64_t first_block = n_reads / rpb;
            const uint32_t cores = (uint32_t)_nbCores;
            const int qdev = store[0]->device;
            const bool on_device = qual_on_device;
            qual_job = std::async(std::launch::async, [quals, qoff, got, first_block, cores, qdev, on_device, &wq, &d_qbuf, &d_qbuf_cap] {
                if (on_device) {
                    if (quals->size() > d_qbuf_cap) {
                        leon_device_free(d_qbuf); d_qbuf = nullptr; d_qbuf_cap = 0;
                        const uint64_t want = quals->size() + quals->size() / 8 + 64;
                        check(nullptr, leon_device_alloc(qdev, want, &d_qbuf), "leon_device_alloc");
                        d_qbuf_cap = want;
                    }
                    if (!quals->empty()) check(nullptr, leon_device_upload(qdev, d_qbuf, quals->data(), quals->size()), "leon_device_upload");
                    int rc = leon_qual_deflate_blocks_device(qdev, static_cast<const uint8_t*>(d_qbuf), qoff->data(), got, READ_PER_BLOCK, StreamWriter::sink, &wq, first_block);
                    check_sink(nullptr, rc, wq, "leon_qual_deflate_blocks_device");
                    return;
                }
                int rc = leon_host_qual_encode_blocks(reinterpret_cast<const uint8_t*>(quals->data()), qoff->data(), got, READ_PER_BLOCK, -1, cores, StreamWriter::sink, &wq,
                                                      first_block);
                check_sink(nullptr, rc, wq, "leon_host_qual_encode_blocks");
            });
        }
        w_quals += seconds_since(t_w);
        t_w = std::chrono::steady_clock::now();
        for (auto& st : store) st->append(batch.bases);
        if (qstore) {                                            // lossy mode: the qualities wait in HBM, beside the bases, for the bloom
            try { qstore->append(batch.quals); }
            catch (const Exception&) { qstore.reset(); }         // no room: they are read again from the file afterwards
            if (qstore && qstore->n_bases > qual_resident_max) qstore.reset();
        }
        for (uint64_t i = 1; i <= got; i++) offsets.push_back(offsets[n_reads] + batch.base_off[i]);
        w_uploads += seconds_since(t_w);
        n_reads += got;
        if (got < batch_reads) break;                            // the partial batch is the last one
    }
    if (qual_job.valid()) qual_job.get();
    const uint64_t n_blocks = (n_reads + rpb - 1) / rpb, n_bases = offsets.back();
    for (auto& st : store) st->set_offsets(offsets);
    hdr_ctx.reset();
    const double t_parse = seconds_since(t_start);

    // ---- solid k-mers -> bloom (Leon::executeCompression: DSK, then createBloom) ----
    const auto t_count = std::chrono::steady_clock::now();
    uint64_t hist[256] = {0};
    uint64_t* d_solid = nullptr; uint64_t n_solid = 0;
    if (n_reads) {
        int rc = leon_kmer_solid_device(store[0]->device, store[0]->d_bases, store[0]->d_off, n_reads, k, (uint32_t)std::max(_abundance, 0), 0, &d_solid, &n_solid, hist);
        if (rc != LEON_OK) throw Exception(std::string("leon_kmer_solid_device: ") + leon_last_error(nullptr));
    }
    struct SolidGuard { uint64_t* p; ~SolidGuard() { leon_device_free(p); } } solid_guard{d_solid};
    const double t_kmers = seconds_since(t_count);
    uint32_t abundance = (uint32_t)_abundance;
    if (_abundance == 0) (void)leon_kmer_auto_cutoff(hist, &abundance);
    const uint64_t tai = std::max<uint64_t>(n_solid * 12, 1000);         // NBITS_PER_KMER = 12 [RECALLED]
    // the DNA stream goes to the device in batches of whole read blocks: the file at once up to 100 M reads, else 2 000 blocks
    // (100 M reads) per leon_dna_encode_batch_device call -- the reads stay resident in HBM, the per-batch buffers are bounded
    uint64_t batch_blocks = std::max<uint64_t>(std::min<uint64_t>(n_blocks, 2000), 1);
    if (const char* e = getenv("LEON_BATCH_BLOCKS")) { const long v = atol(e); if (v > 0) batch_blocks = (uint64_t)v; }   // (tests: several batches on a small file)
    uint64_t batch_max_reads = 0, batch_max_bases = 0;
    for (uint64_t b0 = 0; b0 < n_blocks; b0 += batch_blocks) {
        const uint64_t r0 = b0 * rpb, r1 = std::min<uint64_t>(n_reads, (b0 + batch_blocks) * rpb);
        batch_max_reads = std::max(batch_max_reads, r1 - r0); batch_max_bases = std::max(batch_max_bases, offsets[r1] - offsets[r0]);
    }
    std::vector<CtxPtr> ctx;
    for (int g = 0; g < n_gpus; g++) {
        ctx.push_back(make_ctx(k, tai, store[g]->device));
        check(ctx[g].get(), leon_dna_set_shard(ctx[g].get(), (uint32_t)g, (uint32_t)n_gpus), "leon_dna_set_shard");
        // every per-batch buffer sized now, in one go, before the bloom is built and copied: the first (usually only) encode
        // call then allocates nothing large
        check(ctx[g].get(), leon_dna_reserve(ctx[g].get(), batch_max_reads, batch_max_bases), "leon_dna_reserve");
    }
    check(ctx[0].get(), leon_dna_bloom_insert_device(ctx[0].get(), d_solid, n_solid), "leon_dna_bloom_insert_device");
    uint64_t bloom_bytes = 0;
    check(ctx[0].get(), leon_dna_bloom_nbytes(ctx[0].get(), &bloom_bytes), "leon_dna_bloom_nbytes");
    std::vector<uint8_t> bloom(bloom_bytes);
    check(ctx[0].get(), leon_dna_bloom_download(ctx[0].get(), bloom.data(), bloom_bytes), "leon_dna_bloom_download");
    for (int g = 1; g < n_gpus; g++) check(ctx[g].get(), leon_dna_bloom_upload(ctx[g].get(), bloom.data(), bloom_bytes), "leon_dna_bloom_upload");
    const double t_bloom = seconds_since(t_count);

    // ---- the DNA stream: Dispatcher::iterate(bank, DnaEncoder(this)) upstream, the device path here ----
    const auto t_dna = std::chrono::steady_clock::now();
    std::vector<std::string> gpu_error(n_gpus);
    auto encode_on = [&](int g) {
        try {
            for (uint64_t b0 = 0; b0 < n_blocks; b0 += batch_blocks) {
                const uint64_t r0 = b0 * rpb, r1 = std::min<uint64_t>(n_reads, (b0 + batch_blocks) * rpb);
                int rc = leon_dna_encode_batch_device(ctx[g].get(), store[g]->d_bases, store[g]->d_off + r0, r1 - r0, r0, StreamWriter::sink, &wd);
                check_sink(ctx[g].get(), rc, wd, "leon_dna_encode_batch_device");
            }
        } catch (const std::exception& e) { gpu_error[g] = e.what(); }
    };
    if (n_gpus == 1) encode_on(0);
    else {
        std::vector<std::thread> th;
        for (int g = 0; g < n_gpus; g++) th.emplace_back(encode_on, g);
        for (auto& t : th) t.join();
    }
    for (const std::string& e : gpu_error) if (!e.empty()) throw Exception(e);
    const uint8_t* dict = nullptr; uint64_t dict_size = 0, n_anchors = 0;
    check(ctx[0].get(), leon_dna_finish(ctx[0].get(), &dict, &dict_size, &n_anchors), "leon_dna_finish");
    for (int g = 1; g < n_gpus; g++) { const uint8_t* d; uint64_t ds, na; check(ctx[g].get(), leon_dna_finish(ctx[g].get(), &d, &ds, &na), "leon_dna_finish"); }
    const double t_encode = seconds_since(t_dna);

    // ---- lossy qualities (the default): smoothed against the bloom on the device, then the same zlib blocks ----
    if (keep_qual && !_lossless && qstore && qstore->n_bases == n_bases) {
        // smoothed where they lie, then back to the host chunk by chunk: chunk i is deflated by the host threads while
        // chunk i + 1 is copied
        // ONE call over the whole file: the library smooths the reads in the order of their minimizers, so that reads of the same
        // locus follow one another and share their bloom probes in cache -- which needs them all in one call
        // ... up to the DNA stream's batch size (2 000 blocks = 100 M reads: a call's work buffers grow with its reads, on top of the
        // resident bases and qualities).  A call the device has no room for is retried in halves: smoothing is idempotent (it
        // depends on the bases and on whether a quality is above '@'), and the reads' order inside a call only decides who
        // shares probes with whom, never the bytes.
        for (uint64_t r = 0, step = batch_blocks * rpb; r < n_reads;) {
            const uint64_t got = std::min<uint64_t>(step, n_reads - r);
            const int rc = leon_qual_smooth_batch_device(ctx[0].get(), store[0]->d_bases, store[0]->d_off + r, got, qstore->d_bases + offsets[r]);
            if (rc == LEON_E_HIP && step > 64ull * rpb) { step = std::max<uint64_t>(64ull * rpb, (step / 2 + rpb - 1) / rpb * rpb); leon_device_trim(); continue; }
            check(ctx[0].get(), rc, "leon_qual_smooth_batch_device");
            r += got;
        }
        {   // the sample that picks the encoder: the first reads' smoothed lines
            const uint64_t ns = std::min<uint64_t>(n_reads, 4000);
            std::string sample(offsets[ns] - offsets[0], '\0');
            if (!sample.empty()) check(nullptr, leon_device_download(store[0]->device, &sample[0], qstore->d_bases, sample.size()), "leon_device_download");
            qual_on_device = qual_enc == QualEncoder::Device || (qual_enc == QualEncoder::Auto && runs_are_enough(sample.data(), offsets.data(), ns));
        }
        for (uint64_t r = 0; r < n_reads;) {
            const uint64_t got = std::min<uint64_t>(batch_reads, n_reads - r);
            const uint64_t nb = offsets[r + got] - offsets[r];
            const uint64_t first_block = r / rpb;
            if (qual_on_device) {                                // deflated where they lie
                int rc = leon_qual_deflate_blocks_device(store[0]->device, qstore->d_bases + offsets[r], offsets.data() + r, got, READ_PER_BLOCK, StreamWriter::sink, &wq, first_block);
                check_sink(nullptr, rc, wq, "leon_qual_deflate_blocks_device");
                r += got;
                continue;
            }
            auto quals = std::make_shared<std::string>();
            quals->resize(nb);
            if (nb) check(nullptr, leon_device_download(store[0]->device, &(*quals)[0], qstore->d_bases + offsets[r], nb), "leon_device_download");
            auto qoff = std::make_shared<std::vector<uint64_t>>(got + 1);
            for (uint64_t i = 0; i <= got; i++) (*qoff)[i] = offsets[r + i] - offsets[r];
            if (qual_job.valid()) qual_job.get();
            const uint32_t cores = (uint32_t)_nbCores;
            qual_job = std::async(std::launch::async, [quals, qoff, got, first_block, cores, &wq] {
                int rc = leon_host_qual_encode_blocks(reinterpret_cast<const uint8_t*>(quals->data()), qoff->data(), got, READ_PER_BLOCK, -1, cores, StreamWriter::sink, &wq,
                                                      first_block);
                check_sink(nullptr, rc, wq, "leon_host_qual_encode_blocks");
            });
            r += got;
        }
        if (qual_job.valid()) qual_job.get();
        qstore.reset();
    } else if (keep_qual && !_lossless) {
        Bank again(_inputFilename);
        uint64_t r = 0;
        void* d_q = nullptr; uint64_t d_q_cap = 0;
        struct QGuard { void** p; ~QGuard() { leon_device_free(*p); } } qg{&d_q};
        for (;;) {
            batch.clear();
            const uint64_t got = again.next(batch, batch_reads);
            if (!got) break;
            if (r + got > n_reads || batch.bases.size() != offsets[r + got] - offsets[r]) throw Exception(_inputFilename + " changed while it was being compressed");
            if (batch.quals.size() > d_q_cap) {
                leon_device_free(d_q); d_q = nullptr;
                d_q_cap = batch.quals.size() + batch.quals.size() / 8 + 64;
                check(nullptr, leon_device_alloc(store[0]->device, d_q_cap, &d_q), "leon_device_alloc");
            }
            check(nullptr, leon_device_upload(store[0]->device, d_q, batch.quals.data(), batch.quals.size()), "leon_device_upload");
            check(ctx[0].get(), leon_qual_smooth_batch_device(ctx[0].get(), store[0]->d_bases, store[0]->d_off + r, got, static_cast<uint8_t*>(d_q)),
                  "leon_qual_smooth_batch_device");
            if (r == 0) {                                        // the sample that picks the encoder: the first reads' smoothed lines
                const uint64_t ns = std::min<uint64_t>(got, 4000);
                std::string sample(batch.qual_off[ns] - batch.qual_off[0], '\0');
                if (!sample.empty()) check(nullptr, leon_device_download(store[0]->device, &sample[0], d_q, sample.size()), "leon_device_download");
                qual_on_device = qual_enc == QualEncoder::Device || (qual_enc == QualEncoder::Auto && runs_are_enough(sample.data(), batch.qual_off.data(), ns));
            }
            if (qual_on_device) {
                int rc = leon_qual_deflate_blocks_device(store[0]->device, static_cast<const uint8_t*>(d_q), batch.qual_off.data(), got, rpb, StreamWriter::sink, &wq, r / rpb);
                check_sink(nullptr, rc, wq, "leon_qual_deflate_blocks_device");
            } else {
                check(nullptr, leon_device_download(store[0]->device, &batch.quals[0], d_q, batch.quals.size()), "leon_device_download");
                int rc = leon_host_qual_encode_blocks(reinterpret_cast<const uint8_t*>(batch.quals.data()), batch.qual_off.data(), got, rpb, -1, (uint32_t)_nbCores,
                                                      StreamWriter::sink, &wq, r / rpb);
                check_sink(nullptr, rc, wq, "leon_host_qual_encode_blocks");
            }
            r += got;
            if (got < batch_reads) break;
        }
        if (r != n_reads) throw Exception(_inputFilename + " changed while it was being compressed");
    }

    // ---- tables and metadata ----
    wd.complete(n_blocks, "DNA stream");
    if (keep_header) wh.complete(n_blocks, "header stream");
    if (keep_qual) wq.complete(n_blocks, "quality stream");
    std::vector<uint64_t> table;
    for (uint64_t b = 0; b < n_blocks; b++) {
        const uint64_t r0 = b * rpb, r1 = std::min<uint64_t>(n_reads, r0 + rpb);
        table.push_back(wd.sizes[b]); table.push_back(wd.reads[b]); table.push_back(offsets[r1] - offsets[r0]);
    }
    out.putU64(DS_DNA_TABLE, table.data(), table.size());
    if (keep_header) {
        table.clear();
        for (uint64_t b = 0; b < n_blocks; b++) { table.push_back(wh.sizes[b]); table.push_back(wh.reads[b]); table.push_back(hdr_text[b]); }
        out.putU64(DS_HEADER_TABLE, table.data(), table.size());
        out.putBytes(DS_FIRST_HEADER, first_header.data(), first_header.size());
    }
    if (keep_qual) {
        table.clear();
        for (uint64_t b = 0; b < n_blocks; b++) {
            const uint64_t r0 = b * rpb, r1 = std::min<uint64_t>(n_reads, r0 + rpb);
            table.push_back(wq.sizes[b]); table.push_back(wq.reads[b]); table.push_back(offsets[r1] - offsets[r0]);
        }
        out.putU64(DS_QUAL_TABLE, table.data(), table.size());
    }
    if (fastq && keep_header && keep_qual) {                     // '+' lines that repeat the header (or carry text): `diff` sees them (simple_test.sh:62)
        const std::string plus = bank.plusLines();
        if (!plus.empty()) out.putBytes(DS_PLUS_LINES, plus.data(), plus.size());
    }
    out.putBytes(DS_ANCHOR_DICT, dict, dict_size);
    out.putBytes(DS_BLOOM_BITS, bloom.data(), bloom.size());
    const uint8_t info = (uint8_t)((fastq ? 0 : INFO_FASTA) | (keep_header ? 0 : INFO_NO_HEADER) | (keep_qual ? 0 : INFO_NO_QUAL) | (_lossless ? INFO_LOSSLESS : 0));
    out.putBytes(DS_INFOBYTE, &info, 1);
    uint64_t params[PARAM_COUNT] = {0};
    params[P_VERSION_MAJOR] = 1; params[P_VERSION_MINOR] = 1; params[P_VERSION_PATCH] = 0;       // Leon 1.1.0, /root/reference/CMakeLists.txt:9-11
    params[P_KMER_SIZE] = k; params[P_READS_PER_BLOCK] = rpb; params[P_N_READS] = n_reads; params[P_N_ANCHORS] = n_anchors;
    params[P_ABUNDANCE] = abundance; params[P_BLOOM_TAI] = tai; params[P_BLOOM_N_HASH] = 7; params[P_BLOOM_BLOCK_NBITS] = 12;
    params[P_TOTAL_BASES] = n_bases; params[P_FASTA_LINE_WIDTH] = fastq ? 0 : bank.fastaLineWidth();
    params[P_CONTAINER_REV] = CONTAINER_REV;
    params[P_QUAL_ENCODER] = !keep_qual ? QUAL_ENC_NONE : qual_on_device ? QUAL_ENC_DEVICE_RLE : QUAL_ENC_ZLIB;
    out.putU64(DS_PARAMS, params, PARAM_COUNT);
    out.close();
    if (std::rename(tmp_name.c_str(), _outputFilename.c_str()) != 0) throw Exception("cannot write " + _outputFilename);
    guard.keep = true;

    const uint64_t dna_bytes = wd.bytes + dict_size;
    std::cout << "DNA stream: " << n_reads << " reads, " << n_bases << " bases -> " << dna_bytes << " bytes (" << n_anchors << " anchors, "
              << n_blocks << " blocks, abundance threshold " << abundance << (_abundance ? "" : " (automatic)") << ", " << n_solid << " solid k-mers)\n";
    if (keep_header) std::cout << "header stream: " << header_bytes << " bytes -> " << wh.bytes << " bytes\n";
    if (keep_qual) std::cout << "quality stream (" << (_lossless ? "lossless" : "lossy") << "): " << qual_bytes << " bytes -> " << wq.bytes << " bytes"
                             << (qual_on_device ? " (deflated on the device: runs + dynamic Huffman codes)" : " (zlib on the host threads)") << "\n";
    std::cout << "written to " << _outputFilename << std::endl;
    if (_verbose)
        std::cout << "time: parse + headers" << (keep_qual && _lossless ? " + qualities " : " ") << t_parse << " s, k-mer counting " << t_kmers << " s, contexts + bloom "
                  << t_bloom - t_kmers << " s, DNA encode " << t_encode << " s, total " << seconds_since(t_start) << " s\n"
                  << "the pass over the file: waited for the reader " << w_reader << " s, header blocks " << w_headers << " s, waited for the previous batch's quality blocks "
                  << w_quals << " s, bases" << (qstore_used ? " + qualities" : "") << " to the device " << w_uploads << " s" << std::endl;
}

// ------------------------------------------------------------------------------------------------ -d
void Leon::executeDecompression() {
    const auto t_start = std::chrono::steady_clock::now();
    Container in(_inputFilename, Container::READ);
    const std::vector<uint8_t> infov = in.getBytes(DS_INFOBYTE);
    const std::vector<uint64_t> params = in.getU64(DS_PARAMS);
    if (infov.size() != 1 || params.size() < PARAM_COUNT_REV1) throw Exception(_inputFilename + ": metadata is not what this build writes");
    const uint8_t info = infov[0];
    const bool fasta_in = info & INFO_FASTA, has_header = !(info & INFO_NO_HEADER), has_qual = !(info & INFO_NO_QUAL);
    const uint64_t k = params[P_KMER_SIZE], rpb = params[P_READS_PER_BLOCK], n_reads = params[P_N_READS], n_anchors = params[P_N_ANCHORS];
    const uint64_t tai = params[P_BLOOM_TAI], n_hash = params[P_BLOOM_N_HASH], nbits = params[P_BLOOM_BLOCK_NBITS], total_bases = params[P_TOTAL_BASES];
    if (params[P_VERSION_MAJOR] != 1) throw Exception(_inputFilename + " was written by an incompatible version");
    // the container's own revision (layout::CONTAINER_REV): files of revision 1 have no such word
    const uint64_t rev = params.size() > P_CONTAINER_REV ? params[P_CONTAINER_REV] : 1;
    if (rev < 1 || rev > CONTAINER_REV)
        throw Exception(_inputFilename + " was written by a later build (container revision " + std::to_string(rev) + ", this build reads up to " + std::to_string(CONTAINER_REV) + ")");
    if (k < 3 || k > 63 || rpb == 0 || rpb > (1u << 30) || n_hash < 1 || n_hash > 10 || nbits < 4 || nbits > 16 || n_anchors > (1ull << 32) ||
        n_reads > (1ull << 40) || total_bases > (1ull << 46))
        throw Exception(_inputFilename + ": implausible parameters in the metadata");
    _kmerSize = (size_t)k;
    const uint64_t n_blocks = (n_reads + rpb - 1) / rpb;
    const uint32_t W = k >= 32 ? 2 : 1;
    // block tables, checked against the header's totals before anything is sized from them
    const std::vector<uint64_t> tdna = in.getU64(DS_DNA_TABLE);
    if (tdna.size() != 3 * n_blocks) throw Exception(_inputFilename + ": the DNA block table does not match the read count");
    uint64_t sum_reads = 0, sum_bases = 0;
    for (uint64_t b = 0; b < n_blocks; b++) {
        if (tdna[3 * b + 1] > rpb || tdna[3 * b + 2] > total_bases - sum_bases) throw Exception(_inputFilename + ": the DNA block table does not add up");
        sum_reads += tdna[3 * b + 1]; sum_bases += tdna[3 * b + 2];
    }
    if (sum_reads != n_reads || sum_bases != total_bases) throw Exception(_inputFilename + ": the DNA block table does not add up");
    std::vector<uint64_t> thdr, tqual;
    std::vector<uint8_t> first_header;
    if (has_header) {
        thdr = in.getU64(DS_HEADER_TABLE);
        first_header = in.getBytes(DS_FIRST_HEADER);
        // revision 1 before the text-bytes column existed: 2 words per block (payload bytes, reads).  Such a file decodes like any
        // other: its text buffers start from a guess (64 bytes per header) and the decoder asks for more where that falls short.
        if (rev == 1 && thdr.size() == 2 * n_blocks && n_blocks) {
            std::vector<uint64_t> t3(3 * n_blocks);
            for (uint64_t b = 0; b < n_blocks; b++) { t3[3 * b] = thdr[2 * b]; t3[3 * b + 1] = thdr[2 * b + 1]; t3[3 * b + 2] = 64 * std::min<uint64_t>(thdr[2 * b + 1], rpb); }
            thdr.swap(t3);
        }
        if (thdr.size() != 3 * n_blocks) throw Exception(_inputFilename + ": the header block table does not match the read count");
        for (uint64_t b = 0; b < n_blocks; b++) {
            if (thdr[3 * b + 1] != tdna[3 * b + 1]) throw Exception(_inputFilename + ": header and DNA blocks disagree");
            if (thdr[3 * b + 2] > (1ull << 40)) throw Exception(_inputFilename + ": the header block table does not add up");      // (buffers are sized from it)
        }
    }
    if (has_qual) {
        tqual = in.getU64(DS_QUAL_TABLE);
        if (tqual.size() != 3 * n_blocks) throw Exception(_inputFilename + ": the quality block table does not match the read count");
        for (uint64_t b = 0; b < n_blocks; b++)
            if (tqual[3 * b + 1] != tdna[3 * b + 1] || tqual[3 * b + 2] != tdna[3 * b + 2]) throw Exception(_inputFilename + ": quality and DNA blocks disagree");
    }

    CtxPtr ctx = make_ctx((uint32_t)k, tai, device_for(0), (uint32_t)n_hash, (uint32_t)nbits, (uint32_t)rpb);
    CtxPtr hdr_ctx;                                              // header blocks decode on the device too, on a stream of their own
    if (has_header) hdr_ctx = make_ctx((uint32_t)k, 1000, device_for(0));
    uint64_t header_blocks_on_device = 384;                      // rounds of at least this many blocks (LEON_HEADER_DEVICE_BLOCKS; tests: 0 = always, a huge number = never)
    if (const char* e = getenv("LEON_HEADER_DEVICE_BLOCKS")) header_blocks_on_device = (uint64_t)std::max<long long>(0, atoll(e));
    {
        const std::vector<uint8_t> bloom = in.getBytes(DS_BLOOM_BITS);
        check(ctx.get(), leon_dna_bloom_upload(ctx.get(), bloom.data(), bloom.size()), "leon_dna_bloom_upload");
    }
    // the dictionary stream is one serial chain on a host core (3.7 s at 100 M reads): it runs beside the container reads and
    // the first round's header / quality blocks, and is waited for right before the first DNA blocks go to the device
    std::vector<uint64_t> anchors(std::max<uint64_t>(n_anchors * W, 1));
    std::future<void> dict_job;
    {
        auto dict = std::make_shared<std::vector<uint8_t>>(in.getBytes(DS_ANCHOR_DICT));
        const uint64_t dsz = dict->size();
        dict->push_back(0);
        uint64_t* out_kmers = anchors.data();
        const uint32_t kk = (uint32_t)k;
        dict_job = std::async(std::launch::async, [dict, dsz, n_anchors, kk, out_kmers] {
            if (leon_host_anchor_dict_decode(dict->data(), dsz, n_anchors, kk, out_kmers) != LEON_OK)
                throw Exception(std::string("leon_host_anchor_dict_decode: ") + leon_last_error(nullptr));
        });
    }

    // X.fastq.leon -> X.fastq.d (/root/reference/scripts/simple_test.sh:54,62)
    std::string stem = _inputFilename;
    if (ends_with(stem, ".leon")) stem.resize(stem.size() - 5);
    _outputFilename = stem + ".d";
    struct Fd {
        int fd = -1; std::string path; bool keep = false;
        ~Fd() { if (fd >= 0) ::close(fd); if (!keep && !path.empty()) std::remove(path.c_str()); }     // a failed run leaves no partial output behind
    } ofd;
    ofd.fd = ::open(_outputFilename.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (ofd.fd < 0) throw Exception("cannot write " + _outputFilename);
    ofd.path = _outputFilename;
    const bool fastq_out = !fasta_in && has_qual;               // "-noqual ... will decompress to fasta"
    PlusLines plus;                                             // FASTQ '+' lines that are not bare (absent: all of them are)
    if (fastq_out && has_header && in.exists(DS_PLUS_LINES)) {
        const std::vector<uint8_t> blob = in.getBytes(DS_PLUS_LINES);
        plus = PlusLines::decode(blob.data(), blob.size());
        if (!plus.exc.empty() && plus.exc.back().read >= n_reads) throw Exception(_inputFilename + ": malformed '+'-line table");
    }
    // bytes a read's '+' line has after the '+': nothing, its header again, or its own text
    auto plus_extra = [&plus](uint64_t read, uint64_t hl, size_t* hint) -> uint64_t {
        const PlusLines::Exc* e = plus.exc.empty() ? nullptr : plus.find(read, hint);
        const uint8_t kind = e ? e->kind : plus.def;
        return kind == 1 ? hl : kind == 2 ? e->text.size() : 0;
    };
    const uint64_t wrap = fasta_in ? params[P_FASTA_LINE_WIDTH] : 0;      // sequences wrapped at this width in the original (0: one line)
    const char lead = fastq_out ? '@' : '>';

    // Blocks decoded per round.  A round costs the device ONE block's serial chain whatever the number of blocks in it (up to
    // a few thousand: one wave per block), so rounds should be large; up to five are alive at a time (see the stages below), each
    // ~5 bytes per base in bases, qualities and text: a twentieth of the available RAM each.  Files of more than a few hundred blocks are cut into at least four rounds, so that the host's share of the
    // work (header and quality blocks, formatting, writing) of one round runs beside the decoding of the next.
    uint64_t group = n_blocks ? n_blocks : 1;
    {
        const long pages = sysconf(_SC_AVPHYS_PAGES), page = sysconf(_SC_PAGESIZE);
        const uint64_t avail = pages > 0 && page > 0 ? (uint64_t)pages * (uint64_t)page : (8ull << 30);
        const uint64_t bases_per_block = n_blocks ? std::max<uint64_t>(total_bases / n_blocks, 1) : 1;
        const uint64_t fit = avail / 20 / 5 / bases_per_block;
        group = std::min<uint64_t>(group, std::max<uint64_t>(fit, 64));
        if (n_blocks >= 800) group = std::min<uint64_t>(group, (n_blocks + 3) / 4);
        if (const char* e = getenv("LEON_DECODE_BLOCKS")) { const long v = atol(e); if (v > 0) group = (uint64_t)v; }   // (tests: several rounds on a small file)
    }
    const uint32_t cores = (uint32_t)_nbCores;
    const uint32_t n_cpu = (uint32_t)(_nbCores > 0 ? (uint64_t)_nbCores : usableCpus());
    // bytes that are about to be overwritten anyway: std::vector would clear all of them first, on the critical path
    struct RawBytes {
        std::unique_ptr<uint8_t[]> p; uint64_t n = 0;
        void resize(uint64_t want) { if (want > n) { p.reset(); p.reset(new uint8_t[want]); n = want; } }
        uint8_t* data() { return p.get(); }
        uint64_t size() const { return n; }
    };
    // what one round hands from the decoding stage to the writing stage
    struct DnaGroup { RawBytes bases; std::unique_ptr<uint32_t[]> lens; };   // the DNA blocks of one device call: the bases and lengths of its rounds
    struct Round {
        uint64_t read_index = 0, file_off = 0, g_reads = 0, g_bases = 0, n_text = 0, nb = 0, hdr_text_bytes = 0, block0 = 0;
        std::shared_ptr<DnaGroup> dna; uint64_t base0 = 0, read0 = 0;   // this round's share of them
        const uint8_t* bases() const { return dna->bases.p.get() + base0; }
        const uint32_t* lens() const { return dna->lens.get() + read0; }
        RawBytes hdr, qual, pay_h, pay_q;                        // (gigabytes each: never zero-filled)
        std::vector<uint64_t> off_h, off_q, blk_bases;
        std::vector<uint32_t> blk_reads;
        std::vector<uint64_t> hdr_off, qual_off;
    };
    std::unique_ptr<char[]> text;                                // the writing stage's records (never zero-filled)
    uint64_t text_cap = 0;
    double t_read = 0, t_dna = 0, t_hdr = 0, t_qual = 0, t_text = 0, t_write = 0, t_writer_wait = 0;
    auto lap = [](std::chrono::steady_clock::time_point& t, double& acc) { const auto n = std::chrono::steady_clock::now(); acc += std::chrono::duration<double>(n - t).count(); t = n; };
    // the writing stage: every read's place in the text is known from the lengths, so a round is formatted by all cores at
    // once and written by several (pwrite at disjoint offsets)
    auto write_round = [&](std::shared_ptr<Round> R) {
        auto tl = std::chrono::steady_clock::now();
        const uint64_t g_reads = R->g_reads;
        std::vector<uint64_t> rec_off(g_reads + 1, 0), base_at(g_reads + 1, 0);
        auto seq_text_len = [&](uint64_t len) -> uint64_t { return wrap && len > wrap ? len + (len + wrap - 1) / wrap : len + 1; };
        size_t hint0 = plus.lower(R->read_index);
        for (uint64_t r = 0; r < g_reads; r++) {
            const uint64_t hl = has_header ? R->hdr_off[r + 1] - R->hdr_off[r] : std::to_string(R->read_index + r).size();
            if (fastq_out && R->qual_off[r + 1] - R->qual_off[r] != R->lens()[r]) throw Exception(_inputFilename + ": a read's quality and sequence lengths differ");
            rec_off[r + 1] = rec_off[r] + 1 + hl + 1 + seq_text_len(R->lens()[r]) + (fastq_out ? 2 + (uint64_t)R->lens()[r] + 1 : 0)
                             + (plus.trivial() ? 0 : plus_extra(R->read_index + r, hl, &hint0));
            base_at[r + 1] = base_at[r] + R->lens()[r];
        }
        const uint64_t n_text = rec_off[g_reads];
        if (n_text != R->n_text) throw Exception(_inputFilename + ": the decoded reads do not add up to their blocks' sizes");
        if (n_text > text_cap) { text.reset(); text_cap = n_text + n_text / 16; text.reset(new char[text_cap]); }
        const uint32_t n_fmt = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_cpu, g_reads / 4096 + 1));
        std::mutex err_mu;
        std::string werr;
        auto format_range = [&](uint64_t ra, uint64_t rb) {
            size_t hint = plus.lower(R->read_index + ra);
            for (uint64_t r = ra; r < rb; r++) {
                char* w = text.get() + rec_off[r];
                *w++ = lead;
                if (has_header) { const uint64_t hl = R->hdr_off[r + 1] - R->hdr_off[r]; memcpy(w, R->hdr.data() + R->hdr_off[r], hl); w += hl; }
                else { const std::string idx = std::to_string(R->read_index + r); memcpy(w, idx.data(), idx.size()); w += idx.size(); }
                *w++ = '\n';
                const char* seq = reinterpret_cast<const char*>(R->bases()) + base_at[r];
                const uint64_t len = R->lens()[r];
                if (wrap && len > wrap) {
                    for (uint64_t o2 = 0; o2 < len; o2 += wrap) { const uint64_t m = std::min<uint64_t>(wrap, len - o2); memcpy(w, seq + o2, m); w += m; *w++ = '\n'; }
                } else { memcpy(w, seq, len); w += len; *w++ = '\n'; }
                if (fastq_out) {
                    *w++ = '+';
                    if (!plus.trivial()) {
                        const PlusLines::Exc* e = plus.exc.empty() ? nullptr : plus.find(R->read_index + r, &hint);
                        const uint8_t kind = e ? e->kind : plus.def;
                        if (kind == 1) { const uint64_t hl = R->hdr_off[r + 1] - R->hdr_off[r]; memcpy(w, R->hdr.data() + R->hdr_off[r], hl); w += hl; }
                        else if (kind == 2) { memcpy(w, e->text.data(), e->text.size()); w += e->text.size(); }
                    }
                    *w++ = '\n'; memcpy(w, R->qual.data() + R->qual_off[r], len); w += len; *w++ = '\n';
                }
            }
            // this thread's share of the text goes out as soon as it is formatted
            uint64_t at = rec_off[ra];
            const uint64_t end = rec_off[rb];
            while (at < end) {
                const ssize_t got = ::pwrite(ofd.fd, text.get() + at, (size_t)std::min<uint64_t>(end - at, 1ull << 30), (off_t)(R->file_off + at));
                if (got <= 0) { std::lock_guard<std::mutex> g(err_mu); werr = "cannot write " + _outputFilename; return; }
                at += (uint64_t)got;
            }
        };
        if (n_fmt <= 1) format_range(0, g_reads);
        else {
            std::vector<std::thread> th;
            for (uint32_t t = 0; t < n_fmt; t++) th.emplace_back(format_range, g_reads * t / n_fmt, g_reads * (t + 1) / n_fmt);
            for (auto& t : th) t.join();
        }
        if (!werr.empty()) throw Exception(werr);
        lap(tl, t_text);
    };
    // Three stages, each round through them in turn, the stages of different rounds side by side:
    //   A (this thread)   the payloads out of the container, the DNA blocks on the device -- of `dna_rounds` rounds per device call:
    //                     a call costs one block's serial chain whatever the number of blocks, so a large file makes two calls;
    //   B (a task)        a round's header blocks (their symbols on the device for rounds of many blocks: the host threads are idle
    //                     meanwhile) and, beside them, its quality blocks; one round at a time;
    //   C (a task)        formatting and writing, in file order, once A and B of the round are done.
    // A never waits for B or C of its own rounds -- only for C of rounds further back (memory).
    // (Round 5, measured at configuration #3 and not kept -- profiles/r5_cli_decode_trials.txt: all four rounds in ONE call, one chain instead
    // of two: 9.96 s for this stage against 7.15, `-d -test-file` 23.9 s against 20.2 -- the header and quality blocks of ALL rounds then run
    // beside the call on the same 16 CPUs that stage its 15 GB of output back; and -test-file's comparison moved into the formatting threads
    // (each compares its share with the original as it writes it) instead of both files read back at the end: 22.2 s.  What bounds -d on a
    // box that grants 16 CPUs is the host's share -- header text 6.2 s, quality blocks 3.2-3.5 s, formatting + writing 6.8-7.8 s of all
    // cores each -- not the number of device calls or where the comparison runs.)
    uint64_t dna_rounds = n_blocks >= 800 ? 2 : 1;
    if (const char* e = getenv("LEON_DECODE_DNA_ROUNDS")) { const long v = atol(e); if (v > 0) dna_rounds = (uint64_t)v; }   // (tests)
    // (what the header-symbol task and the rounds' tasks read: declared before `drain`, so destroyed after it has waited for them)
    RawBytes all_pay_h; std::vector<uint64_t> all_off_h; std::vector<uint32_t> all_reads_h;
    struct SymSet { leon_header_symbols* h = nullptr; ~SymSet() { leon_header_symbols_free(h); } };
    auto hdr_set = std::make_shared<SymSet>();
    std::shared_future<void> hdr_symbols;
    struct Drain {                                               // (no task outlives what it refers to, whatever way this function is left)
        std::vector<std::shared_future<void>> all;
        ~Drain() { for (auto& f : all) if (f.valid()) f.wait(); }
    } drain;
    std::shared_future<void> host_before, writer_before;
    std::deque<std::shared_future<void>> pending;               // stage C of the rounds in flight, oldest first
    uint64_t read_index = 0, bases_out = 0, next_file_off = 0;  // next_file_off: owned by stage C (one round at a time)
    // Header blocks whose symbols the device decodes: ALL of the file's blocks in ONE device call, started here, beside everything else.
    // A header block is one serial chain on one wave (~1.5 s for 50 000 headers) whether the call holds 200 blocks or 2 000, so a call
    // per round paid that chain every round (4 x 2.7 s of configuration #3's 20 s); the rounds now only rebuild their blocks' text
    // from the symbols, on the host threads.
    const bool hdr_on_device = has_header && n_blocks > 0 && group >= header_blocks_on_device;
    if (hdr_on_device) {
        all_off_h.assign(n_blocks + 1, 0); all_reads_h.resize(n_blocks);
        for (uint64_t b = 0; b < n_blocks; b++) { all_off_h[b + 1] = all_off_h[b] + thdr[3 * b]; all_reads_h[b] = (uint32_t)tdna[3 * b + 1]; }
        all_pay_h.resize(all_off_h[n_blocks] + 1);
        for (uint64_t b = 0; b < n_blocks; b++) {
            const std::vector<uint8_t> blk = in.getBytes(Container::blockPath(GROUP_HEADER, b));
            if (blk.size() != thdr[3 * b]) throw Exception(_inputFilename + ": block " + std::to_string(b) + " of " + GROUP_HEADER + " has not the size its table says");
            if (!blk.empty()) memcpy(all_pay_h.data() + all_off_h[b], blk.data(), blk.size());
        }
        leon_dna_ctx* hc = hdr_ctx.get();
        const uint8_t* pay = all_pay_h.data(); const uint64_t* off = all_off_h.data(); const uint32_t* nr = all_reads_h.data();
        hdr_symbols = std::async(std::launch::async, [hc, pay, off, nr, n_blocks, hdr_set] {
            if (leon_header_decode_symbols(hc, pay, off, nr, n_blocks, &hdr_set->h) != LEON_OK) throw Exception(std::string("header blocks: ") + leon_last_error(hc));
        }).share();
        drain.all.push_back(hdr_symbols);
    }
    for (uint64_t a0 = 0; a0 < n_blocks; a0 += group * dna_rounds) {
        const uint64_t a1 = std::min(n_blocks, a0 + group * dna_rounds), na = a1 - a0;
        auto tl = std::chrono::steady_clock::now();
        while (pending.size() > dna_rounds) { pending.front().get(); pending.pop_front(); }   // (their errors, and those of their stage B, surface here)
        lap(tl, t_writer_wait);
        auto gather = [&](const char* grp, const std::vector<uint64_t>& tab, uint32_t stride, uint64_t g0, uint64_t nb, RawBytes& pay, std::vector<uint64_t>& off) {
            off.assign(nb + 1, 0);
            for (uint64_t b = 0; b < nb; b++) off[b + 1] = off[b] + tab[stride * (g0 + b)];
            pay.resize(off[nb] + 1);
            for (uint64_t b = 0; b < nb; b++) {
                const std::vector<uint8_t> blk = in.getBytes(Container::blockPath(grp, g0 + b));
                if (blk.size() != tab[stride * (g0 + b)]) throw Exception(_inputFilename + ": block " + std::to_string(g0 + b) + " of " + grp + " has not the size its table says");
                if (!blk.empty()) memcpy(pay.data() + off[b], blk.data(), blk.size());
            }
        };
        // stage A, its host part: the DNA payloads of the call, and every round's header and quality payloads
        auto G = std::make_shared<DnaGroup>();
        RawBytes pay; std::vector<uint64_t> off;
        std::vector<uint32_t> a_reads(na); std::vector<uint64_t> a_bases(na);
        uint64_t ga_reads = 0, ga_bases = 0;
        for (uint64_t b = 0; b < na; b++) { a_reads[b] = (uint32_t)tdna[3 * (a0 + b) + 1]; a_bases[b] = tdna[3 * (a0 + b) + 2]; ga_reads += a_reads[b]; ga_bases += a_bases[b]; }
        gather(GROUP_DNA, tdna, 3, a0, na, pay, off);
        G->bases.resize(ga_bases + 1); G->lens.reset(new uint32_t[ga_reads + 1]);
        std::vector<std::pair<std::shared_ptr<Round>, std::shared_future<void>>> rounds;
        uint64_t base0 = 0, read0 = 0;
        for (uint64_t g0 = a0; g0 < a1; g0 += group) {
            const uint64_t g1 = std::min(a1, g0 + group), nb = g1 - g0;
            auto R = std::make_shared<Round>();
            R->nb = nb; R->blk_reads.assign(a_reads.begin() + (g0 - a0), a_reads.begin() + (g1 - a0)); R->blk_bases.assign(a_bases.begin() + (g0 - a0), a_bases.begin() + (g1 - a0));
            uint64_t g_reads = 0, g_bases = 0;
            for (uint64_t b = 0; b < nb; b++) { g_reads += R->blk_reads[b]; g_bases += R->blk_bases[b]; }
            R->read_index = read_index; R->g_reads = g_reads; R->g_bases = g_bases;
            R->dna = G; R->base0 = base0; R->read0 = read0;
            if (has_header && !hdr_on_device) gather(GROUP_HEADER, thdr, 3, g0, nb, R->pay_h, R->off_h);
            R->block0 = g0;
            if (fastq_out) gather(GROUP_QUAL, tqual, 3, g0, nb, R->pay_q, R->off_q);
            R->hdr_off.assign(g_reads + 1, 0); R->qual_off.assign(g_reads + 1, 0);
            R->hdr_text_bytes = 64;
            if (has_header) for (uint64_t b = 0; b < nb; b++) R->hdr_text_bytes += thdr[3 * (g0 + b) + 2];
            // stage B
            std::shared_future<void> host_job = std::async(std::launch::async, [&, R, host_before] {
                if (host_before.valid()) host_before.wait();
                auto th = std::chrono::steady_clock::now();
                const uint64_t nb = R->nb, g_bases = R->g_bases;
                const bool on_device = hdr_on_device;
                auto decode_quals = [&] {
                    R->qual.resize(g_bases + 1);
                    if (leon_host_qual_decode_blocks(R->pay_q.data(), R->off_q.data(), R->blk_reads.data(), R->blk_bases.data(), nb, R->qual.data(), g_bases, R->qual_off.data(), cores) != LEON_OK)
                        throw Exception(std::string("leon_host_qual_decode_blocks: ") + leon_last_error(nullptr));
                };
                // while the first round waits for the device's header symbols the host threads have nothing to do: the quality blocks go then
                std::future<void> quals_beside;
                if (fastq_out && on_device && hdr_symbols.valid() && hdr_symbols.wait_for(std::chrono::seconds(0)) != std::future_status::ready)
                    quals_beside = std::async(std::launch::async, decode_quals);
                if (has_header) {
                    uint64_t need = 0;
                    R->hdr.resize(R->hdr_text_bytes);             // from the block table (a wrong entry only costs the second call below)
                    // The symbols of every header block come from ONE device call over the whole file (above); a round rebuilds its
                    // blocks' text from them.  Files of a few hundred blocks: the host threads alone are quicker (10 M headers in 200
                    // blocks: 0.62 s against 1.27 s).  A set whose symbols did not fit the device buffer (free text in every header)
                    // sends the rounds to the host decoder, from the same payloads.
                    bool host_decoder = !on_device;
                    if (on_device) {
                        try { hdr_symbols.get(); }
                        catch (...) { if (quals_beside.valid()) { try { quals_beside.get(); } catch (...) {} } throw; }
                    }
                    auto decode = [&]() -> int {
                        if (on_device && !host_decoder) {
                            const int rc = leon_header_text_from_symbols(hdr_set->h, R->block0, nb, R->blk_reads.data(), first_header.data(), first_header.size(),
                                                                         R->hdr.data(), R->hdr.size(), R->hdr_off.data(), &need, cores);
                            if (rc != LEON_E_STATE) return rc;
                            host_decoder = true;
                        }
                        if (on_device)
                            return leon_host_header_decode_blocks(all_pay_h.data(), all_off_h.data() + R->block0, R->blk_reads.data(), nb, first_header.data(), first_header.size(),
                                                                  R->hdr.data(), R->hdr.size(), R->hdr_off.data(), &need, cores);
                        return leon_host_header_decode_blocks(R->pay_h.data(), R->off_h.data(), R->blk_reads.data(), nb, first_header.data(), first_header.size(), R->hdr.data(),
                                                              R->hdr.size(), R->hdr_off.data(), &need, cores);
                    };
                    int rc = decode();
                    if (rc == LEON_E_OVERFLOW) { R->hdr.resize(need + 1); rc = decode(); }
                    if (rc != LEON_OK) {
                        const std::string msg = std::string("header blocks: ") + leon_last_error(nullptr);
                        if (quals_beside.valid()) { try { quals_beside.get(); } catch (...) {} }
                        throw Exception(msg);
                    }
                }
                lap(th, t_hdr);
                if (quals_beside.valid()) quals_beside.get();
                else if (fastq_out) decode_quals();
                lap(th, t_qual);
                R->pay_h = RawBytes(); R->pay_q = RawBytes();     // (the payloads are not needed any more)
            }).share();
            drain.all.push_back(host_job);
            host_before = host_job;
            rounds.emplace_back(R, host_job);
            base0 += g_bases; read0 += g_reads; read_index += g_reads; bases_out += g_bases;
        }
        lap(tl, t_read);
        // stage A, the device part
        if (dict_job.valid()) dict_job.get();
        check(ctx.get(), leon_dna_decode_blocks(ctx.get(), anchors.data(), n_anchors, pay.data(), off.data(), a_reads.data(), a_bases.data(), na, G->bases.data(), ga_bases,
                                                G->lens.get()), "leon_dna_decode_blocks");
        lap(tl, t_dna);
        // stage C
        for (auto& rh : rounds) {
            std::shared_ptr<Round> R = rh.first;
            std::shared_future<void> host_job = rh.second;
            std::shared_future<void> writer = std::async(std::launch::async, [&, R, host_job, writer_before] {
                host_job.get();                                  // (a failed stage B fails this round's stage C with its message)
                if (writer_before.valid()) writer_before.get();  // file order; a failure before this round stops the rounds after it
                const uint64_t g_reads = R->g_reads, g_bases = R->g_bases;
                // the size of this round's text: where the next round's begins
                uint64_t n_text = 0;
                if (has_header && !wrap && plus.exc.empty())     // (the decoder has checked that the lengths add up to the block table's bases)
                    n_text = 2 * g_reads + R->hdr_off[g_reads] + g_bases + g_reads + (fastq_out ? 3 * g_reads + g_bases : 0)
                             + (plus.def == 1 ? R->hdr_off[g_reads] : 0);      // every '+' line repeats its header
                else {
                    auto seq_text_len = [&](uint64_t len) -> uint64_t { return wrap && len > wrap ? len + (len + wrap - 1) / wrap : len + 1; };
                    size_t hint = plus.lower(R->read_index);
                    for (uint64_t r = 0; r < g_reads; r++) {
                        const uint64_t hl = has_header ? R->hdr_off[r + 1] - R->hdr_off[r] : std::to_string(R->read_index + r).size();
                        n_text += 1 + hl + 1 + seq_text_len(R->lens()[r]) + (fastq_out ? 2 + (uint64_t)R->lens()[r] + 1 : 0)
                                  + (plus.trivial() ? 0 : plus_extra(R->read_index + r, hl, &hint));
                    }
                }
                R->n_text = n_text; R->file_off = next_file_off;
                next_file_off += n_text;
                write_round(R);
            }).share();
            drain.all.push_back(writer);
            pending.push_back(writer);
            writer_before = writer;
        }
    }
    {
        auto tl = std::chrono::steady_clock::now();
        while (!pending.empty()) { pending.front().get(); pending.pop_front(); }
        lap(tl, t_write);
    }
    if (::close(ofd.fd) != 0) { ofd.fd = -1; throw Exception("cannot write " + _outputFilename); }
    ofd.fd = -1;
    if (read_index != n_reads) throw Exception("the block tables do not add up to the header's read count");
    ofd.keep = true;
    std::cout << n_reads << " reads, " << bases_out << " bases decoded from " << n_blocks << " blocks, written to " << _outputFilename << std::endl;
    if (_verbose)
        std::cout << "time: " << seconds_since(t_start) << " s (" << (n_blocks + group - 1) / group << " round(s); container reads " << t_read << ", dictionary + DNA blocks on the device " << t_dna
                  << "; beside them, a round behind: header blocks " << t_hdr << " + quality blocks " << t_qual << "; another round behind: formatting + writing "
                  << t_text << "; waited for them " << t_writer_wait + t_write << ")" << std::endl;
    if (_testFile) testDecompressedFile();
}

// -test-file: "check decompressed file against original" (/root/reference/INSTALL:22): X.fastq.d against X.fastq (or X.fastq.gz) beside it
void Leon::testDecompressedFile() {
    std::string orig = _outputFilename.substr(0, _outputFilename.size() - 2);
    struct stat st_o, st_d;
    const bool plain = ::stat(orig.c_str(), &st_o) == 0 && S_ISREG(st_o.st_mode);
    if (plain && ::stat(_outputFilename.c_str(), &st_d) == 0) {
        // both files are plain: compared in slices by all cores (pread at disjoint offsets); the first difference is the lowest one found
        int fa = ::open(orig.c_str(), O_RDONLY), fb = ::open(_outputFilename.c_str(), O_RDONLY);
        if (fa < 0 || fb < 0) { if (fa >= 0) ::close(fa); if (fb >= 0) ::close(fb); throw Exception("-test-file: cannot reopen the files"); }
        const uint64_t na = (uint64_t)st_o.st_size, nb = (uint64_t)st_d.st_size, n = std::min(na, nb);
        const uint32_t n_thr = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(_nbCores > 0 ? (uint64_t)_nbCores : usableCpus(), n / (64ull << 20) + 1));
        std::vector<uint64_t> first_diff(n_thr, ~0ull);
        std::vector<int> io_error(n_thr, 0);
        auto compare = [&](uint32_t t) {
            const uint64_t lo = n * t / n_thr, hi = n * (t + 1) / n_thr;
            std::vector<char> ba(4 << 20), bb(4 << 20);
            for (uint64_t at = lo; at < hi;) {
                const size_t want = (size_t)std::min<uint64_t>(ba.size(), hi - at);
                const ssize_t ga = ::pread(fa, ba.data(), want, (off_t)at), gb = ::pread(fb, bb.data(), want, (off_t)at);
                if (ga <= 0 || gb <= 0) { io_error[t] = 1; return; }
                const size_t m = (size_t)std::min(ga, gb);
                if (memcmp(ba.data(), bb.data(), m) != 0) {
                    size_t i = 0; while (ba[i] == bb[i]) i++;
                    first_diff[t] = at + i; return;
                }
                at += m;
            }
        };
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < n_thr; t++) th.emplace_back(compare, t);
        for (auto& t : th) t.join();
        ::close(fa); ::close(fb);
        for (int e : io_error) if (e) throw Exception("-test-file: read error while comparing " + _outputFilename + " with " + orig);
        uint64_t at = ~0ull;
        for (uint64_t d : first_diff) at = std::min(at, d);
        if (at == ~0ull && na != nb) at = n;
        if (at != ~0ull) throw Exception("-test-file: " + _outputFilename + " differs from " + orig + " at byte " + std::to_string(at));
        std::cout << "test-file: " << _outputFilename << " is identical to " << orig << std::endl;
        return;
    }
    gzFile a = gzopen(orig.c_str(), "rb");
    if (!a) { orig += ".gz"; a = gzopen(orig.c_str(), "rb"); }
    if (!a) throw Exception("-test-file: the original file is not beside " + _outputFilename);
    gzFile b = gzopen(_outputFilename.c_str(), "rb");
    if (!b) { gzclose(a); throw Exception("-test-file: cannot reopen " + _outputFilename); }
    std::vector<char> ba(1 << 20), bb(1 << 20);
    uint64_t at = 0;
    bool same = true;
    for (;;) {
        const int na = gzread(a, ba.data(), (unsigned)ba.size()), nb = gzread(b, bb.data(), (unsigned)bb.size());
        if (na != nb || na < 0 || (na > 0 && memcmp(ba.data(), bb.data(), (size_t)na) != 0)) {
            same = false;
            for (int i = 0; i < std::min(na, nb) && ba[i] == bb[i]; i++) at++;
            break;
        }
        if (na == 0) break;
        at += (uint64_t)na;
    }
    gzclose(a); gzclose(b);
    if (!same) throw Exception("-test-file: " + _outputFilename + " differs from " + orig + " at byte " + std::to_string(at));
    std::cout << "test-file: " << _outputFilename << " is identical to " << orig << std::endl;
}

// ------------------------------------------------------------------------------------------------ self tests (no GPU)
int selftest_container(const std::string& path) {
    std::vector<uint8_t> a(100000), empty;
    for (size_t i = 0; i < a.size(); i++) a[i] = (uint8_t)(i * 2654435761u >> 13);
    std::vector<uint64_t> t = { 1, 2, 3, ~0ull, 0 };
    {
        Container c(path, Container::CREATE);
        c.putBytes(Container::blockPath(GROUP_DNA, 0), a.data(), a.size());
        c.putBytes(Container::blockPath(GROUP_DNA, 1), empty.data(), 0);
        c.putBytes(Container::blockPath(GROUP_HEADER, 0), a.data(), 17);
        c.putBytes(DS_INFOBYTE, a.data(), 1);
        c.putU64(DS_DNA_TABLE, t.data(), t.size());
        c.putU64(DS_PARAMS, t.data(), 0);
        c.close();
    }
    Container c(path, Container::READ);
    bool ok = c.getBytes(Container::blockPath(GROUP_DNA, 0)) == a && c.getBytes(Container::blockPath(GROUP_DNA, 1)).empty() &&
              c.getBytes(Container::blockPath(GROUP_HEADER, 0)) == std::vector<uint8_t>(a.begin(), a.begin() + 17) && c.getU64(DS_DNA_TABLE) == t &&
              c.getU64(DS_PARAMS).empty() && c.exists(DS_INFOBYTE) && !c.exists(Container::blockPath(GROUP_QUAL, 0)) && !c.exists("nothing/here");
    bool threw = false;
    try { (void)c.getBytes("leon/dna/block_7"); } catch (const Exception&) { threw = true; }
    bool threw2 = false;
    try { (void)c.getBytes(DS_DNA_TABLE); } catch (const Exception&) { threw2 = true; }      // a u64 array is not a byte array
    ok = ok && threw && threw2;
    std::cout << (ok ? "container selftest OK" : "container selftest FAILED") << std::endl;
    return ok ? 0 : 1;
}

int selftest_bank(const std::string& path) {
    Bank bank(path);
    ReadBatch b;
    uint64_t n = 0, nb = 0, nh = 0, nq = 0, fnv_b = 1469598103934665603ull, fnv_h = fnv_b, fnv_q = fnv_b;
    auto mix = [](uint64_t& h, const std::string& s) { for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; } };
    const bool fq = bank.isFastq();
    for (;;) {
        b.clear();
        const uint64_t got = bank.next(b, 1000);
        if (!got) break;
        n += got; nb += b.bases.size(); nh += b.headers.size(); nq += b.quals.size();
        mix(fnv_b, b.bases); mix(fnv_h, b.headers); mix(fnv_q, b.quals);
    }
    std::cout << "{\"fastq\": " << (fq ? "true" : "false") << ", \"reads\": " << n << ", \"bases\": " << nb << ", \"header_bytes\": " << nh << ", \"qual_bytes\": " << nq
              << ", \"fasta_line_width\": " << bank.fastaLineWidth() << ", \"fnv_bases\": " << fnv_b << ", \"fnv_headers\": " << fnv_h << ", \"fnv_quals\": " << fnv_q;
    // the '+'-line table as the container stores it, read back the way -d reads it
    const std::string plus = bank.plusLines();
    const PlusLines L = PlusLines::decode(reinterpret_cast<const uint8_t*>(plus.data()), plus.size());
    uint64_t text_bytes = 0;
    for (const PlusLines::Exc& e : L.exc) text_bytes += e.text.size();
    std::cout << ", \"plus\": {\"bytes\": " << plus.size() << ", \"default\": " << (int)L.def << ", \"exceptions\": " << L.exc.size() << ", \"first_exception\": "
              << (L.exc.empty() ? -1ll : (long long)L.exc[0].read) << ", \"text_bytes\": " << text_bytes << "}}" << std::endl;
    return 0;
}

}  // namespace leon_host

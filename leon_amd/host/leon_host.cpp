// leon_host.cpp -- see leon_host.hpp.  Host glue only: every byte of the DNA stream comes from libleon_dna.so.
//
// Interim container written by `-c` and read by `-d` (little endian), until the .leon HDF5 layout row is built:
//   "LEONDNA2" | u32 k | u32 reads_per_block | u64 n_reads | u64 n_blocks | u64 n_anchors | u64 dict_bytes |
//   u64 bloom_tai | u64 bloom_bytes | u32 n_hash | u32 block_nbits |
//   n_blocks x (u64 size, u64 n_reads, u64 n_bases) | dictionary stream | bloom bytes | block payloads
#include "leon_host.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

namespace leon_host {

const char* Leon::STR_COMPRESS = "-c";
const char* Leon::STR_DECOMPRESS = "-d";

namespace {
void check(leon_dna_ctx* ctx, int rc, const char* what) {
    if (rc != LEON_OK) throw Exception(std::string(what) + ": " + leon_last_error(ctx));
}
int sink(void* user, uint64_t block_id, const uint8_t* payload, uint64_t size, uint32_t n_reads) {
    static_cast<Leon*>(user)->writeBlock(payload, size, (int)n_reads, block_id);
    return 0;
}
// FASTA / FASTQ (plain text) -> sequences; stands in for gatb's Bank (out of scope)
std::vector<Sequence> read_bank(const std::string& path) {
    std::ifstream in(path);
    if (!in) throw Exception("cannot open " + path);
    std::vector<Sequence> out;
    std::string line;
    bool fastq = false, first = true;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (first) { fastq = line[0] == '@'; first = false; }
        if (fastq) {
            Sequence s; s.comment = line.substr(1);
            if (!std::getline(in, s.data)) break;
            std::string plus;
            if (!std::getline(in, plus) || !std::getline(in, s.quality)) throw Exception("truncated FASTQ record in " + path);
            s.index = out.size(); out.push_back(std::move(s));
        } else if (line[0] == '>') {
            Sequence s; s.comment = line.substr(1); s.index = out.size(); out.push_back(std::move(s));
        } else {
            if (out.empty()) throw Exception("FASTA data before the first header in " + path);
            out.back().data += line;
        }
    }
    return out;
}
template <typename T> void put(std::ofstream& o, T v) { o.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
}  // namespace

// ------------------------------------------------------------------------------------------------ DnaEncoder
DnaEncoder::DnaEncoder(Leon* leon) : leon_(leon), offsets_(1, 0) {}
DnaEncoder::DnaEncoder(const DnaEncoder& o) : leon_(o.leon_), offsets_(1, 0) {}
DnaEncoder::~DnaEncoder() {
    try { flush(); } catch (...) {}
}
void DnaEncoder::operator()(Sequence& s) {
    bases_.append(s.getDataBuffer(), s.getDataSize());
    offsets_.push_back(bases_.size());
    if (offsets_.size() - 1 == leon_->_batchReads) flush();
}
void DnaEncoder::flush() {
    const uint64_t n = offsets_.size() - 1;
    if (!n) return;
    check(leon_->_ctx, leon_dna_encode_batch(leon_->_ctx, reinterpret_cast<const uint8_t*>(bases_.data()), offsets_.data(), n,
                                             leon_->_nextRead, sink, leon_), "leon_dna_encode_batch");
    leon_->_nextRead += n;
    bases_.clear();
    offsets_.assign(1, 0);
}

// ------------------------------------------------------------------------------------------------ Leon
Leon::Leon() {}
Leon::~Leon() { if (_ctx) leon_dna_ctx_destroy(_ctx); }

void Leon::run(int argc, char* argv[]) {
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](const char* flag) -> std::string {
            if (i + 1 >= argc) throw Exception(std::string("option ") + flag + " needs a value");
            return argv[++i];
        };
        if (a == "-file") _inputFilename = need("-file");
        else if (a == STR_COMPRESS) _compress = true;
        else if (a == STR_DECOMPRESS) _decompress = true;
        else if (a == "-kmer-size") _kmerSize = (size_t)std::stoul(need("-kmer-size"));
        else if (a == "-abundance") _abundance = std::stoi(need("-abundance"));
        else if (a == "-nb-cores") (void)need("-nb-cores");          // accepted; the device replaces the thread pool
        else if (a == "-gpus") _gpus = std::stoi(need("-gpus"));
        else if (a == "-verbose") _verbose = std::stoi(need("-verbose")) != 0;
        else if (a == "-lossless" || a == "-seq-only" || a == "-noheader" || a == "-noqual") {}   // other streams: not built
        else throw Exception("unknown option " + a);
    }
    if (_inputFilename.empty()) throw Exception("option -file is mandatory");
    if (_compress == _decompress) throw Exception("choose one of -c (compress) or -d (decompress)");
    execute();
}

void Leon::execute() {
    if (_compress) executeCompression(); else executeDecompression();
}

void Leon::writeBlock(const uint8_t* data, uint64_t size, int encodedSequenceCount, uint64_t blockID) {
    if (blockID != _blockSizes.size() / 2) throw Exception("blocks arrived out of order");
    _blocks.insert(_blocks.end(), data, data + size);
    _blockSizes.push_back(size);
    _blockSizes.push_back((uint64_t)encodedSequenceCount);
}

void Leon::executeCompression() {
    if (_kmerSize < 3 || _kmerSize > 63) throw Exception("-kmer-size must be in 3..63");
    std::vector<Sequence> bank = read_bank(_inputFilename);
    // solid k-mers: counted on the device (leon_kmer_solid, the stand-in for DSK's SortingCountAlgorithm)
    const uint32_t W = _kmerSize >= 32 ? 2 : 1;
    std::string all_bases; std::vector<uint64_t> all_off(1, 0);
    for (const Sequence& s : bank) { all_bases += s.data; all_off.push_back(all_bases.size()); }
    std::vector<uint64_t> solid(std::max<size_t>(all_bases.size(), 1) * W);
    uint64_t n_solid = 0;
    {
        int rc = leon_kmer_solid(0, reinterpret_cast<const uint8_t*>(all_bases.data()), all_off.data(), bank.size(), (uint32_t)_kmerSize,
                                 (uint32_t)std::max(_abundance, 1), 0, solid.data(), all_bases.size(), &n_solid, nullptr);
        if (rc != LEON_OK) throw Exception(std::string("leon_kmer_solid: ") + leon_last_error(nullptr));
    }
    solid.resize(n_solid * W);
    const uint64_t tai = std::max<uint64_t>(n_solid * 12, 1000);           // NBITS_PER_KMER = 12 [RECALLED]
    leon_dna_cfg cfg = {};
    cfg.struct_size = sizeof(cfg);
    cfg.kmer_size = (uint32_t)_kmerSize; cfg.reads_per_block = READ_PER_BLOCK;
    cfg.bloom_n_hash = 7; cfg.bloom_block_nbits = 12; cfg.bloom_tai = tai; cfg.device_id = 0;
    check(nullptr, leon_dna_ctx_create(&cfg, &_ctx), "leon_dna_ctx_create");
    check(_ctx, leon_dna_bloom_insert(_ctx, solid.data(), n_solid), "leon_dna_bloom_insert");
    {
        DnaEncoder enc(this);                       // upstream: Dispatcher::iterate(itSeq, DnaEncoder(this), READ_PER_BLOCK)
        for (Sequence& s : bank) enc(s);
    }                                               // ~DnaEncoder flushes the last (partial) block
    const uint8_t* dict = nullptr; uint64_t dict_size = 0, n_anchors = 0;
    check(_ctx, leon_dna_finish(_ctx, &dict, &dict_size, &n_anchors), "leon_dna_finish");
    uint64_t bloom_bytes = 0;
    check(_ctx, leon_dna_bloom_nbytes(_ctx, &bloom_bytes), "leon_dna_bloom_nbytes");
    std::vector<uint8_t> bloom(bloom_bytes);
    check(_ctx, leon_dna_bloom_download(_ctx, bloom.data(), bloom_bytes), "leon_dna_bloom_download");

    // output name: strip nothing, append ".leon" (data/toy.fasta -> data/toy.fasta.leon, /root/reference/INSTALL:21-23)
    _outputFilename = _inputFilename + ".leon";
    std::ofstream o(_outputFilename, std::ios::binary);
    if (!o) throw Exception("cannot write " + _outputFilename);
    o.write("LEONDNA2", 8);
    put<uint32_t>(o, (uint32_t)_kmerSize); put<uint32_t>(o, READ_PER_BLOCK);
    put<uint64_t>(o, bank.size()); put<uint64_t>(o, _blockSizes.size() / 2); put<uint64_t>(o, n_anchors);
    put<uint64_t>(o, dict_size); put<uint64_t>(o, tai); put<uint64_t>(o, bloom_bytes); put<uint32_t>(o, 7); put<uint32_t>(o, 12);
    for (size_t b = 0; b < _blockSizes.size() / 2; b++) {          // block table; the bases per block let -d size its output
        uint64_t nb = 0;
        for (size_t r = b * READ_PER_BLOCK; r < std::min(bank.size(), (b + 1) * (size_t)READ_PER_BLOCK); r++) nb += bank[r].getDataSize();
        put<uint64_t>(o, _blockSizes[2 * b]); put<uint64_t>(o, _blockSizes[2 * b + 1]); put<uint64_t>(o, nb);
    }
    o.write(reinterpret_cast<const char*>(dict), (std::streamsize)dict_size);
    o.write(reinterpret_cast<const char*>(bloom.data()), (std::streamsize)bloom_bytes);
    o.write(reinterpret_cast<const char*>(_blocks.data()), (std::streamsize)_blocks.size());
    uint64_t n_bases = 0;
    for (const Sequence& s : bank) n_bases += s.getDataSize();
    std::cout << "DNA stream: " << bank.size() << " reads, " << n_bases << " bases -> " << (_blocks.size() + dict_size)
              << " bytes (" << n_anchors << " anchors, " << _blockSizes.size() / 2 << " blocks), written to " << _outputFilename
              << std::endl;
}

// ------------------------------------------------------------------------------------------------ DnaDecoder
// upstream: DnaDecoder::execute() per block on the dispatcher's threads [RECALLED]; here every block at once on the device
void DnaDecoder::execute(const std::vector<uint64_t>& anchors, const std::vector<uint8_t>& payloads, const std::vector<uint64_t>& payload_off,
                         const std::vector<uint32_t>& block_reads, const std::vector<uint64_t>& block_bases,
                         std::vector<uint8_t>& bases, std::vector<uint32_t>& lengths) {
    uint64_t nb = 0, nr = 0;
    for (uint64_t x : block_bases) nb += x;
    for (uint32_t x : block_reads) nr += x;
    bases.assign(nb + 1, 0);
    lengths.assign(nr + 1, 0);
    const uint32_t W = leon_->_kmerSize >= 32 ? 2 : 1;
    check(leon_->_ctx, leon_dna_decode_blocks(leon_->_ctx, anchors.data(), anchors.size() / W, payloads.data(), payload_off.data(),
                                              block_reads.data(), block_bases.data(), block_reads.size(), bases.data(), nb, lengths.data()),
          "leon_dna_decode_blocks");
    bases.resize(nb);
    lengths.resize(nr);
}

void Leon::executeDecompression() {
    std::ifstream in(_inputFilename, std::ios::binary);
    if (!in) throw Exception("cannot open " + _inputFilename);
    auto get = [&](void* p, size_t n) { if (!in.read(reinterpret_cast<char*>(p), (std::streamsize)n)) throw Exception("truncated container " + _inputFilename); };
    char magic[8];
    get(magic, 8);
    if (std::memcmp(magic, "LEONDNA2", 8) != 0) throw Exception(_inputFilename + " is not a container written by this build's -c");
    uint32_t k, rpb, n_hash, nbits; uint64_t n_reads, n_blocks, n_anchors, dict_bytes, tai, bloom_bytes;
    get(&k, 4); get(&rpb, 4); get(&n_reads, 8); get(&n_blocks, 8); get(&n_anchors, 8); get(&dict_bytes, 8); get(&tai, 8);
    get(&bloom_bytes, 8); get(&n_hash, 4); get(&nbits, 4);
    if (n_blocks > (1ull << 32) || dict_bytes > (1ull << 40) || bloom_bytes > (1ull << 40)) throw Exception("implausible container header");
    std::vector<uint64_t> table(3 * n_blocks);
    if (n_blocks) get(table.data(), table.size() * 8);
    std::vector<uint8_t> dict(dict_bytes + 1), bloom(bloom_bytes);
    if (dict_bytes) get(dict.data(), dict_bytes);
    if (bloom_bytes) get(bloom.data(), bloom_bytes);
    std::vector<uint64_t> pay_off(n_blocks + 1, 0), block_bases(n_blocks);
    std::vector<uint32_t> block_reads(n_blocks);
    for (uint64_t b = 0; b < n_blocks; b++) {
        pay_off[b + 1] = pay_off[b] + table[3 * b];
        block_reads[b] = (uint32_t)table[3 * b + 1];
        block_bases[b] = table[3 * b + 2];
    }
    std::vector<uint8_t> payloads(pay_off[n_blocks] + 1);
    if (pay_off[n_blocks]) get(payloads.data(), pay_off[n_blocks]);

    _kmerSize = k;
    leon_dna_cfg cfg = {};
    cfg.struct_size = sizeof(cfg);
    cfg.kmer_size = k; cfg.reads_per_block = rpb; cfg.bloom_n_hash = n_hash; cfg.bloom_block_nbits = nbits; cfg.bloom_tai = tai; cfg.device_id = 0;
    check(nullptr, leon_dna_ctx_create(&cfg, &_ctx), "leon_dna_ctx_create");
    check(_ctx, leon_dna_bloom_upload(_ctx, bloom.data(), bloom_bytes), "leon_dna_bloom_upload");
    const uint32_t W = k >= 32 ? 2 : 1;
    std::vector<uint64_t> anchors(std::max<uint64_t>(n_anchors * W, 1));
    if (leon_host_anchor_dict_decode(dict.data(), dict_bytes, n_anchors, k, anchors.data()) != LEON_OK)
        throw Exception(std::string("leon_host_anchor_dict_decode: ") + leon_last_error(nullptr));
    anchors.resize(n_anchors * W);
    std::vector<uint8_t> bases; std::vector<uint32_t> lengths;
    DnaDecoder(this).execute(anchors, payloads, pay_off, block_reads, block_bases, bases, lengths);
    if (lengths.size() != n_reads) throw Exception("the block table does not add up to the header's read count");

    // output name: X.fastq.leon -> X.fastq.d (/root/reference/scripts/simple_test.sh:54,62).  Only the DNA stream exists in this
    // build (no header / quality streams), so the output is the sequences, one per line, in file order.
    std::string stem = _inputFilename;
    if (stem.size() > 5 && stem.compare(stem.size() - 5, 5, ".leon") == 0) stem.resize(stem.size() - 5);
    _outputFilename = stem + ".d";
    std::ofstream o(_outputFilename, std::ios::binary);
    if (!o) throw Exception("cannot write " + _outputFilename);
    uint64_t at = 0;
    for (uint32_t len : lengths) {
        o.write(reinterpret_cast<const char*>(bases.data() + at), len);
        o.put('\n');
        at += len;
    }
    std::cout << "DNA stream: " << n_reads << " reads, " << at << " bases decoded from " << n_blocks << " blocks, written to "
              << _outputFilename << std::endl;
}

}  // namespace leon_host

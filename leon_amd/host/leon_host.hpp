// leon_host.hpp -- host-side mirror (C++) of the reference interface AROUND the DNA encode path, above the C-ABI:
//   Leon        : the Tool-shaped class main() drives with `Leon().run(argc, argv)` (/root/reference/src/main.cpp:44),
//                 run() parses the documented flags (/root/reference/README.md:38-58) and calls execute();
//   DnaEncoder  : the functor handed one Sequence at a time (upstream Dispatcher::iterate(bank, DnaEncoder(this)) [RECALLED]);
//                 here it batches whole blocks and calls leon_dna_encode_batch, its destructor flushes the last block;
//   Exception   : gatb::core::system::Exception's getMessage() contract (/root/reference/src/main.cpp:46-49).
//   DnaDecoder  : the inverse (`-d`), all blocks at once through leon_dna_decode_blocks.
// Only the DNA stream is produced (headers, qualities, HDF5 are out of this round's scope: DESIGN.md section 11); the
// output of -c is an interim flat container documented in leon_host.cpp, -d writes the sequences one per line.
#pragma once
#include <stdint.h>
#include <exception>
#include <string>
#include <vector>

#include "../../include/leon_dna.h"

namespace leon_host {

class Exception : public std::exception {
public:
    explicit Exception(const std::string& m) : msg_(m) {}
    const char* getMessage() const { return msg_.c_str(); }
    const char* what() const noexcept override { return msg_.c_str(); }
private:
    std::string msg_;
};

struct Sequence {                         // the slice of gatb's Sequence the path uses
    std::string comment, data, quality;
    size_t index = 0;
    const char* getDataBuffer() const { return data.data(); }
    size_t getDataSize() const { return data.size(); }
    size_t getIndex() const { return index; }
};

class Leon;

class DnaEncoder {
public:
    explicit DnaEncoder(Leon* leon);
    DnaEncoder(const DnaEncoder& o);      // upstream copies the functor per thread; copies share the Leon's one stream
    ~DnaEncoder();                        // flushes what is buffered (upstream: writeBlock of the last partial block)
    void operator()(Sequence& s);
    void flush();
private:
    Leon* leon_;
    std::string bases_;
    std::vector<uint64_t> offsets_;
};

// upstream DnaDecoder::execute() decodes one block; this one hands all blocks to the device decoder at once
class DnaDecoder {
public:
    explicit DnaDecoder(Leon* leon) : leon_(leon) {}
    void execute(const std::vector<uint64_t>& anchors, const std::vector<uint8_t>& payloads, const std::vector<uint64_t>& payload_off,
                 const std::vector<uint32_t>& block_reads, const std::vector<uint64_t>& block_bases,
                 std::vector<uint8_t>& bases, std::vector<uint32_t>& lengths);
private:
    Leon* leon_;
};

class Leon {
public:
    static const int READ_PER_BLOCK = 50000;
    static const char* STR_COMPRESS;      // "-c"
    static const char* STR_DECOMPRESS;    // "-d"
    Leon();
    ~Leon();
    void run(int argc, char* argv[]);     // Tool::run: parse, then execute()
    void execute();
    // Leon::writeBlock(data, size, encodedSequenceCount, blockID): called by the encode path in block order
    void writeBlock(const uint8_t* data, uint64_t size, int encodedSequenceCount, uint64_t blockID);

    // state the functor reads (upstream: public members of Leon)
    size_t _kmerSize = 31;
    int _abundance = 3;
    int _gpus = 1;
    std::string _inputFilename, _outputFilename;
    bool _compress = false, _decompress = false, _verbose = false;
    leon_dna_ctx* _ctx = nullptr;
    uint64_t _nextRead = 0;
    uint64_t _batchReads = 64 * (uint64_t)READ_PER_BLOCK;

private:
    void executeCompression();
    void executeDecompression();
    std::vector<uint8_t> _blocks;                       // concatenated payloads
    std::vector<uint64_t> _blockSizes;                  // (size, nReads) pairs, upstream _blockSizes
};

}  // namespace leon_host

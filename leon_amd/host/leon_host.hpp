// leon_host.hpp -- host-side mirror (C++) of the reference interface AROUND the DNA encode path, above the C-ABI:
//   Leon      : the Tool-shaped class main() drives with `Leon().run(argc, argv)` (/root/reference/src/main.cpp:44);
//               run() parses the documented flags (/root/reference/README.md:38-58, -test-file: /root/reference/INSTALL:22)
//               and calls execute() -> executeCompression() / executeDecompression() [RECALLED names];
//   Exception : gatb::core::system::Exception's getMessage() contract (/root/reference/src/main.cpp:46-49).
// `-c` writes the `.leon` HDF5 container (leon_container.hpp) with the three streams Leon has -- headers, DNA,
// qualities -- and `-d` restores the FASTA / FASTQ file from it (the reference's acceptance test,
// /root/reference/scripts/simple_test.sh:51-62).  Every byte of the DNA and header streams comes from libleon_dna.so
// (HIP kernels); the quality stream is zlib on host threads (lossless) after a device smoothing pass (lossy, the default).
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

class Container;

class Leon {
public:
    static const int READ_PER_BLOCK = 50000;
    static const char* STR_COMPRESS;      // "-c"
    static const char* STR_DECOMPRESS;    // "-d"
    Leon();
    ~Leon();
    void run(int argc, char* argv[]);     // Tool::run: parse, then execute()
    void execute();

    // options (upstream: members of Leon filled from the Tool's properties)
    std::string _inputFilename, _outputFilename;
    bool _compress = false, _decompress = false;
    size_t _kmerSize = 31;
    int _abundance = 0;                   // 0 = automatic (/root/reference/README.md:54)
    int _nbCores = 0;                     // 0 = all
    int _gpus = 1;
    bool _lossless = false, _seqOnly = false, _noHeader = false, _noQual = false, _testFile = false, _verbose = false;
    std::string _qualDeflate;             // -qual-deflate host|device|auto; empty = not given: zlib on the host threads (the reference's bytes)

private:
    void executeCompression();
    void executeDecompression();
    void testDecompressedFile();
};

// `leon -selftest-container <path>` / `leon -selftest-bank <file>`: host logic that runs without a GPU (used by the CPU tests)
int selftest_container(const std::string& path);
int selftest_bank(const std::string& path);

}  // namespace leon_host

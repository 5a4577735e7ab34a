// bank.hpp -- streaming FASTA / FASTQ reader (plain or gzip), the slice of gatb's Bank (bank/impl/BankFasta [RECALLED]) the
// host mirror needs: sequences in file order, each with comment (header text after '>' / '@'), data and quality.
// Out of the hot path's scope ("build writes own minimal parser"); reads the file once, in batches,
// so the host never holds more than one batch of it.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace leon_host {

// one batch of reads as three blobs + offsets, the shape the C-ABI takes
struct ReadBatch {
    std::string bases, headers, quals;
    std::vector<uint64_t> base_off{0}, header_off{0}, qual_off{0};
    uint64_t size() const { return base_off.size() - 1; }
    void clear() { bases.clear(); headers.clear(); quals.clear(); base_off.assign(1, 0); header_off.assign(1, 0); qual_off.assign(1, 0); }
};

class Bank {
public:
    explicit Bank(const std::string& path);      // throws leon_host::Exception when the file cannot be opened
    ~Bank();
    Bank(const Bank&) = delete;
    Bank& operator=(const Bank&) = delete;
    bool isFastq();                               // decided by the first record's first byte ('@' or '>')
    // appends up to max_reads reads to batch; returns the number appended (0 at the end of the file)
    uint64_t next(ReadBatch& batch, uint64_t max_reads);
    uint64_t bytesEstimate() const { return file_bytes_; }   // size of the file on disk (compressed size for .gz)
    // FASTA: the width every sequence line but a record's last has when the file wraps its sequences consistently, else 0
    // (sequences on one line, or lines of differing widths); valid once the whole file has been read
    uint64_t fastaLineWidth() const { return (wrap_seen_ && wrap_ok_ && !single_empty_ && single_max_ <= wrap_) ? wrap_ : 0; }
private:
    bool getline(std::string& line);
    bool view(const char*& p, size_t& n);         // the next line where it lies (in the read buffer when it ends there), else assembled in spill_
    void* gz_ = nullptr;                          // gzFile: reads plain files transparently as well
    std::string path_, pending_, spill_;
    bool have_pending_ = false, decided_ = false, fastq_ = false;
    std::vector<char> buf_;
    size_t buf_pos_ = 0, buf_len_ = 0;
    uint64_t file_bytes_ = 0, n_read_ = 0;
    uint64_t wrap_ = 0;
    bool wrap_seen_ = false, wrap_ok_ = true, single_empty_ = false;
    uint64_t single_max_ = 0;                     // longest sequence that sat on one line: must fit the wrap width too
};

}  // namespace leon_host

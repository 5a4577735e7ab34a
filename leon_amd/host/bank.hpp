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
    // FASTQ: what follows the '+' of every record's third line, as PlusLines::decode reads it back; empty when every record
    // has a bare "+" (the usual case; nothing is stored then).  Valid once the whole file has been read.
    std::string plusLines() const;
private:
    bool getline(std::string& line);
    bool view(const char*& p, size_t& n);         // the next line where it lies (in the read buffer when it ends there), else assembled in spill_
    void* gz_ = nullptr;                          // gzFile (gzip input)
    int fd_ = -1;                                 // plain input: read(2) straight into buf_ (zlib's transparent mode copies every byte once more)
    std::string path_, pending_, spill_;
    bool have_pending_ = false, decided_ = false, fastq_ = false;
    std::vector<char> buf_;
    size_t buf_pos_ = 0, buf_len_ = 0;
    uint64_t file_bytes_ = 0, n_read_ = 0;
    uint64_t wrap_ = 0;
    bool wrap_seen_ = false, wrap_ok_ = true, single_empty_ = false;
    uint64_t single_max_ = 0;                     // longest sequence that sat on one line: must fit the wrap width too
    int plus_default_ = -1;                       // kind of the first record's '+' line (0 bare, 1 repeats the header); -1: none seen
    std::string plus_exc_;                        // the records that differ from it: varint(read index delta), kind, [varint(length), text]
    uint64_t plus_prev_ = 0;
};

// The '+' lines of a FASTQ file (third line of a record): bare, or the header again (older Illumina / SRA dumps), or -- never
// seen, but `diff` would see it -- any other text.  One default kind + the exceptions.
struct PlusLines {
    uint8_t def = 0;
    struct Exc { uint64_t read; uint8_t kind; std::string text; };
    std::vector<Exc> exc;                         // sorted by read
    bool trivial() const { return def == 0 && exc.empty(); }
    static PlusLines decode(const uint8_t* p, uint64_t n);        // throws leon_host::Exception on a malformed blob
    // kind of read `read`'s line (0 bare, 1 header, 2 text); *hint walks the exceptions forwards over increasing reads
    const Exc* find(uint64_t read, size_t* hint) const {
        size_t i = *hint;
        while (i < exc.size() && exc[i].read < read) i++;
        *hint = i;
        return (i < exc.size() && exc[i].read == read) ? &exc[i] : nullptr;
    }
    size_t lower(uint64_t read) const;            // first exception at or after `read`
};

}  // namespace leon_host

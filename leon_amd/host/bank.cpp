#include "bank.hpp"
#include "leon_host.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cerrno>

#include <algorithm>
#include <cstring>

namespace leon_host {

static void put_varint(std::string& s, uint64_t v) { while (v >= 0x80) { s.push_back((char)(v | 0x80)); v >>= 7; } s.push_back((char)v); }

std::string Bank::plusLines() const {
    if (plus_default_ <= 0 && plus_exc_.empty()) return std::string();
    std::string out(1, (char)plus_default_);
    return out + plus_exc_;
}

PlusLines PlusLines::decode(const uint8_t* p, uint64_t n) {
    PlusLines L;
    if (!n) return L;
    if (p[0] > 1) throw Exception("malformed '+'-line table");
    L.def = p[0];
    uint64_t at = 1, read = 0;
    auto varint = [&]() -> uint64_t {
        uint64_t v = 0;
        for (uint32_t sh = 0;; sh += 7) {
            if (at >= n || sh > 63) throw Exception("malformed '+'-line table");
            const uint8_t c = p[at++];
            v |= (uint64_t)(c & 0x7F) << sh;
            if (!(c & 0x80)) return v;
        }
    };
    while (at < n) {
        const uint64_t d = varint();
        if (!L.exc.empty() && d == 0) throw Exception("malformed '+'-line table");
        read += d;
        if (at >= n) throw Exception("malformed '+'-line table");
        const uint8_t kind = p[at++];
        if (kind > 2 || kind == L.def) throw Exception("malformed '+'-line table");
        std::string text;
        if (kind == 2) {
            const uint64_t len = varint();
            if (len > n - at) throw Exception("malformed '+'-line table");
            text.assign(reinterpret_cast<const char*>(p) + at, len);
            at += len;
        }
        L.exc.push_back(Exc{read, kind, std::move(text)});
    }
    return L;
}
size_t PlusLines::lower(uint64_t read) const {
    return (size_t)(std::lower_bound(exc.begin(), exc.end(), read, [](const Exc& e, uint64_t r) { return e.read < r; }) - exc.begin());
}

Bank::Bank(const std::string& path) : path_(path), buf_(4 << 20) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) throw Exception("cannot open " + path);
    file_bytes_ = (uint64_t)st.st_size;
    // gzip input (magic 1f 8b) goes through zlib; anything else is read as it is
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw Exception("cannot open " + path);
    unsigned char magic[2] = {0, 0};
    const ssize_t got = ::pread(fd_, magic, 2, 0);
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        ::close(fd_); fd_ = -1;
        gz_ = gzopen(path.c_str(), "rb");
        if (!gz_) throw Exception("cannot open " + path);
        gzbuffer((gzFile)gz_, 1 << 20);
    }
}
Bank::~Bank() { if (gz_) gzclose((gzFile)gz_); if (fd_ >= 0) ::close(fd_); }

// one line without its terminator ("\n" or "\r\n"); false at the end of the file
bool Bank::getline(std::string& line) {
    line.clear();
    bool any = false;
    for (;;) {
        if (buf_pos_ == buf_len_) {
            long got;
            if (fd_ >= 0) {
                do { got = (long)::read(fd_, buf_.data(), buf_.size()); } while (got < 0 && errno == EINTR);
                if (got < 0) throw Exception(std::string("read error in ") + path_ + ": " + strerror(errno));
                if (got == 0) break;
            } else {
                got = gzread((gzFile)gz_, buf_.data(), (unsigned)buf_.size());
                if (got < 0) { int e = 0; throw Exception(std::string("read error in ") + path_ + ": " + gzerror((gzFile)gz_, &e)); }
                if (got == 0) {
                    // end of the data -- or of a TRUNCATED .gz, which zlib reports as 0 bytes + Z_BUF_ERROR after handing out
                    // what it could inflate: compressing that part and calling it success would lose the rest silently
                    int e = 0;
                    const char* msg = gzerror((gzFile)gz_, &e);
                    if (e != Z_OK && e != Z_STREAM_END) throw Exception(path_ + " is truncated or corrupt (" + (msg && *msg ? msg : "unexpected end of file") + ")");
                    break;
                }
            }
            buf_pos_ = 0; buf_len_ = (size_t)got;
        }
        const char* p = buf_.data() + buf_pos_;
        const char* nl = (const char*)memchr(p, '\n', buf_len_ - buf_pos_);
        any = true;
        if (nl) { line.append(p, (size_t)(nl - p)); buf_pos_ += (size_t)(nl - p) + 1; break; }
        line.append(p, buf_len_ - buf_pos_);
        buf_pos_ = buf_len_;
    }
    if (!line.empty() && line.back() == '\r') line.pop_back();
    return any;
}

bool Bank::isFastq() {
    if (!decided_) {
        std::string l;
        while (getline(l)) {
            if (l.empty()) continue;
            pending_ = l; have_pending_ = true;
            fastq_ = l[0] == '@';
            if (l[0] != '@' && l[0] != '>') throw Exception(path_ + " is neither FASTA nor FASTQ (first record starts with '" + l.substr(0, 1) + "')");
            break;
        }
        decided_ = true;
    }
    return fastq_;
}

// the next line without its terminator, not copied when it ends inside the read buffer (the usual case: a megabyte of
// buffer, lines of a few hundred bytes); valid until the next call
bool Bank::view(const char*& p, size_t& n) {
    if (have_pending_) { have_pending_ = false; p = pending_.data(); n = pending_.size(); return true; }
    if (buf_pos_ < buf_len_) {
        const char* s = buf_.data() + buf_pos_;
        const char* nl = (const char*)memchr(s, '\n', buf_len_ - buf_pos_);
        if (nl) {
            p = s; n = (size_t)(nl - s);
            buf_pos_ += n + 1;
            if (n && p[n - 1] == '\r') n--;
            return true;
        }
    }
    if (!getline(spill_)) return false;                         // crosses a refill (or the file ends without a newline)
    p = spill_.data(); n = spill_.size();
    return true;
}

uint64_t Bank::next(ReadBatch& b, uint64_t max_reads) {
    isFastq();
    uint64_t n = 0;
    const char* p; size_t len;
    while (n < max_reads) {
        if (!view(p, len)) break;
        if (len == 0) continue;
        if (fastq_) {
            if (p[0] != '@') throw Exception("malformed FASTQ record " + std::to_string(n_read_ + 1) + " in " + path_);
            const size_t hdr_at = b.headers.size(), hdr_len = len - 1;
            b.headers.append(p + 1, len - 1);
            if (!view(p, len)) throw Exception("truncated FASTQ record in " + path_);
            const size_t seq_len = len;
            b.bases.append(p, len);
            if (!view(p, len)) throw Exception("truncated FASTQ record in " + path_);
            if (len == 0 || p[0] != '+') throw Exception("malformed FASTQ record " + std::to_string(n_read_ + 1) + " in " + path_);
            {   // the text after the '+': nothing, the header again, or something else
                // (a bare '+' under an empty header is both "bare" and "the header again": whichever the file's default is)
                const int kind = len == 1 ? (hdr_len == 0 && plus_default_ == 1 ? 1 : 0)
                                          : (len - 1 == hdr_len && memcmp(p + 1, b.headers.data() + hdr_at, hdr_len) == 0) ? 1 : 2;
                if (plus_default_ < 0) plus_default_ = kind == 2 ? 0 : kind;
                if (kind != plus_default_) {
                    put_varint(plus_exc_, n_read_ - plus_prev_); plus_prev_ = n_read_;
                    plus_exc_.push_back((char)kind);
                    if (kind == 2) { put_varint(plus_exc_, len - 1); plus_exc_.append(p + 1, len - 1); }
                }
            }
            if (!view(p, len)) throw Exception("truncated FASTQ record in " + path_);
            if (len != seq_len) throw Exception("FASTQ record " + std::to_string(n_read_ + 1) + " of " + path_ + ": quality and sequence lengths differ");
            b.quals.append(p, len);
        } else {
            if (p[0] != '>') throw Exception("FASTA data before the first header in " + path_);
            b.headers.append(p + 1, len - 1);
            uint64_t n_lines = 0, last_len = 0;
            while (view(p, len)) {                                   // sequence lines up to the next header
                if (len && p[0] == '>') { pending_.assign(p, len); have_pending_ = true; break; }
                // wrapped sequences: every line but a record's last must have the same width, the last one 1..width
                if (n_lines) {
                    if (!wrap_seen_) { wrap_seen_ = true; wrap_ = last_len; }
                    if (last_len != wrap_ || wrap_ == 0) wrap_ok_ = false;
                }
                last_len = len; n_lines++;
                b.bases.append(p, len);
            }
            if (n_lines && last_len == 0) wrap_ok_ = false;          // an empty last line would not come back
            if (wrap_seen_ && n_lines && last_len > wrap_) wrap_ok_ = false;
            if (n_lines == 0) single_empty_ = true;
            if (n_lines == 1) single_max_ = std::max<uint64_t>(single_max_, last_len);
        }
        b.header_off.push_back(b.headers.size());
        b.base_off.push_back(b.bases.size());
        b.qual_off.push_back(b.quals.size());
        n++; n_read_++;
    }
    return n;
}

}  // namespace leon_host

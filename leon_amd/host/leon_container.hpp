// leon_container.hpp -- the `.leon` file: ONE HDF5 file (/root/reference/README.md:88 "stored into a single HDF5 binary file:
// '.leon'"), written and read through the HDF5 C API.  Upstream goes through gatb's StorageHDF5 (groups of byte
// collections, Storage::ostream per block, StorageTools::saveBloom) [RECALLED]; the group / dataset NAMES below are
// recalled, not verified -- they all live in the ONE table `layout` so that a maintainer with gatb-core at hand can
// correct them in one place.  libhdf5 is loaded at run time (dlopen, LEON_HDF5_LIB overrides the
// search), so the tool has no link-time dependency on the image's conda tree.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace leon_host {

namespace layout {                                   // [RECALLED] names -- edit here, nowhere else
constexpr const char* GROUP_ROOT = "leon";
constexpr const char* GROUP_METADATA = "leon/metadata";
constexpr const char* GROUP_DNA = "leon/dna";             // block_<id>: one read block's DNA payload (Leon::writeBlock)
constexpr const char* GROUP_HEADER = "leon/header";       // block_<id>: header payloads
constexpr const char* GROUP_QUAL = "leon/qual";           // block_<id>: quality payloads (Leon::writeBlockLena)
constexpr const char* GROUP_ANCHORS = "leon/anchors";
constexpr const char* GROUP_BLOOM = "bloom";              // StorageTools::saveBloom
constexpr const char* DS_INFOBYTE = "leon/metadata/infobyte";          // u8[1]
constexpr const char* DS_PARAMS = "leon/metadata/params";              // u64[PARAM_COUNT]
constexpr const char* DS_FIRST_HEADER = "leon/metadata/firstheader";   // u8[]: the file's first header, plain
constexpr const char* DS_PLUS_LINES = "leon/metadata/pluslines";       // u8[]: FASTQ '+' lines that are not bare (bank.hpp PlusLines); absent = all bare
constexpr const char* DS_DNA_TABLE = "leon/metadata/dna_blocksizes";   // u64[3 n]: payload bytes, reads, bases per block
constexpr const char* DS_HEADER_TABLE = "leon/metadata/header_blocksizes";   // u64[3 n]: payload bytes, reads, bytes of header text
constexpr const char* DS_QUAL_TABLE = "leon/metadata/qual_blocksizes";       // u64[3 n]: payload bytes, reads, quality bytes
constexpr const char* DS_ANCHOR_DICT = "leon/anchors/dict";            // u8[]: the anchor-dictionary stream
constexpr const char* DS_BLOOM_BITS = "bloom/bits";                    // u8[nchar]
constexpr const char* BLOCK_PREFIX = "block_";
enum Param : uint32_t { P_VERSION_MAJOR, P_VERSION_MINOR, P_VERSION_PATCH, P_KMER_SIZE, P_READS_PER_BLOCK, P_N_READS, P_N_ANCHORS,
                        P_ABUNDANCE, P_BLOOM_TAI, P_BLOOM_N_HASH, P_BLOOM_BLOCK_NBITS, P_TOTAL_BASES, P_FASTA_LINE_WIDTH,
                        PARAM_COUNT_REV1,                                  // what containers of revision 1 (rounds 1-3 of this build) hold: 13 words
                        P_CONTAINER_REV = PARAM_COUNT_REV1,                // revision of THIS container layout (not Leon's version): see CONTAINER_REV
                        P_QUAL_ENCODER,                                    // who wrote leon/qual/block_<i>: QUAL_ENC_*
                        PARAM_COUNT };
// Container revisions.  1 (no P_CONTAINER_REV word): the header table has 2 words per block (payload bytes, reads) in files written before
// the text-bytes column existed, 3 in later ones -- told apart by the table's size.  2: 3 words per block, P_QUAL_ENCODER recorded.
constexpr uint64_t CONTAINER_REV = 2;
// leon/qual/block_<i> are zlib streams either way; what differs is whose bytes they are
enum QualEnc : uint64_t { QUAL_ENC_NONE = 0,          // no quality stream (or a revision-1 file: not recorded)
                          QUAL_ENC_ZLIB = 1,          // zlib's compress2 at its default level (what upstream writes [RECALLED]): the default
                          QUAL_ENC_DEVICE_RLE = 2 };  // the device's deflate (runs + dynamic Huffman codes): inflates to the same text, other bytes
// info byte: bit 0 FASTA input (else FASTQ), bit 1 no header stream, bit 2 no quality stream, bit 3 lossless qualities
enum Info : uint8_t { INFO_FASTA = 1, INFO_NO_HEADER = 2, INFO_NO_QUAL = 4, INFO_LOSSLESS = 8 };
}  // namespace layout

class Container {
public:
    enum Mode { READ, CREATE };
    Container(const std::string& path, Mode mode);   // throws leon_host::Exception
    ~Container();
    Container(const Container&) = delete;
    Container& operator=(const Container&) = delete;
    void close();                                    // flushes; throws on failure (the destructor swallows)
    // datasets are addressed by their path from the file root; missing groups are created on the way
    void putBytes(const std::string& path, const void* data, uint64_t size);
    void putU64(const std::string& path, const uint64_t* data, uint64_t count);
    bool exists(const std::string& path);
    std::vector<uint8_t> getBytes(const std::string& path);
    std::vector<uint64_t> getU64(const std::string& path);
    static std::string blockPath(const char* group, uint64_t block_id);
private:
    void ensureGroups(const std::string& dataset_path);
    void put(const std::string& path, const void* data, uint64_t count, bool u64);
    uint64_t get(const std::string& path, bool u64, void* out, uint64_t cap_elems, bool size_only);
    int64_t file_ = -1;
    std::string path_;
    std::vector<std::string> groups_;
    bool writable_ = false;
};

}  // namespace leon_host

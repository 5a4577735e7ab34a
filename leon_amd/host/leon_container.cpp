#include "leon_container.hpp"
#include "leon_host.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>

namespace leon_host {

namespace {

// ---- the slice of the HDF5 1.10 C API the container uses, bound at run time ----
typedef int64_t hid_t;
typedef int herr_t;
typedef int htri_t;
typedef unsigned long long hsize_t;
typedef long long hssize_t;
struct H5 {
    herr_t (*open)(void);
    hid_t (*Fcreate)(const char*, unsigned, hid_t, hid_t);
    hid_t (*Fopen)(const char*, unsigned, hid_t);
    herr_t (*Fclose)(hid_t);
    herr_t (*Fflush)(hid_t, int);
    hid_t (*Gcreate2)(hid_t, const char*, hid_t, hid_t, hid_t);
    herr_t (*Gclose)(hid_t);
    hid_t (*Screate_simple)(int, const hsize_t*, const hsize_t*);
    herr_t (*Sclose)(hid_t);
    hssize_t (*Sget_simple_extent_npoints)(hid_t);
    hid_t (*Dcreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t, hid_t);
    hid_t (*Dopen2)(hid_t, const char*, hid_t);
    herr_t (*Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void*);
    herr_t (*Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void*);
    hid_t (*Dget_space)(hid_t);
    hid_t (*Dget_type)(hid_t);
    size_t (*Tget_size)(hid_t);
    herr_t (*Tclose)(hid_t);
    herr_t (*Dclose)(hid_t);
    htri_t (*Lexists)(hid_t, const char*, hid_t);
    hid_t (*Pcreate)(hid_t);
    herr_t (*Pset_obj_track_times)(hid_t, unsigned);
    herr_t (*Pclose)(hid_t);
    herr_t (*Eset_auto2)(hid_t, void*, void*);
    hid_t T_U8, T_U64, P_DCPL, P_GCPL;
};
H5 g_h5;
std::once_flag g_once;
std::string g_load_error;

void load_hdf5() {
    std::vector<std::string> names;
    if (const char* e = getenv("LEON_HDF5_LIB")) names.push_back(e);
    for (const char* n : { "libhdf5.so.103", "libhdf5_serial.so.103", "/opt/conda/lib/libhdf5.so.103", "libhdf5.so", "libhdf5_serial.so",
                           "/opt/conda/lib/libhdf5.so" }) names.push_back(n);
    void* lib = nullptr;
    for (const std::string& n : names) if ((lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL))) break;
    if (!lib) { g_load_error = "the HDF5 library (libhdf5.so, 1.10) was not found; set LEON_HDF5_LIB to its path"; return; }
    bool ok = true;
    auto sym = [&](const char* n) -> void* { void* p = dlsym(lib, n); if (!p) { ok = false; g_load_error = std::string("libhdf5 lacks ") + n; } return p; };
#define BIND(field, name) g_h5.field = reinterpret_cast<decltype(g_h5.field)>(sym(name))
    BIND(open, "H5open"); BIND(Fcreate, "H5Fcreate"); BIND(Fopen, "H5Fopen"); BIND(Fclose, "H5Fclose"); BIND(Fflush, "H5Fflush");
    BIND(Gcreate2, "H5Gcreate2"); BIND(Gclose, "H5Gclose"); BIND(Screate_simple, "H5Screate_simple"); BIND(Sclose, "H5Sclose");
    BIND(Sget_simple_extent_npoints, "H5Sget_simple_extent_npoints"); BIND(Dcreate2, "H5Dcreate2"); BIND(Dopen2, "H5Dopen2");
    BIND(Dwrite, "H5Dwrite"); BIND(Dread, "H5Dread"); BIND(Dget_space, "H5Dget_space"); BIND(Dget_type, "H5Dget_type");
    BIND(Tget_size, "H5Tget_size"); BIND(Tclose, "H5Tclose"); BIND(Dclose, "H5Dclose"); BIND(Lexists, "H5Lexists");
    BIND(Pcreate, "H5Pcreate"); BIND(Pset_obj_track_times, "H5Pset_obj_track_times"); BIND(Pclose, "H5Pclose"); BIND(Eset_auto2, "H5Eset_auto2");
#undef BIND
    if (!ok) return;
    // the typedefs above are the 1.10+ ABI (64-bit hid_t, the *_ID_g / *_g globals read below): an older library would be
    // called with the wrong types
    {
        auto libversion = reinterpret_cast<herr_t (*)(unsigned*, unsigned*, unsigned*)>(sym("H5get_libversion"));
        unsigned maj = 0, min = 0, rel = 0;
        if (!ok) return;
        if (libversion(&maj, &min, &rel) < 0 || maj != 1 || min < 10) {
            g_load_error = "the HDF5 library found is version " + std::to_string(maj) + "." + std::to_string(min) + "." + std::to_string(rel) +
                           "; 1.10 or later is needed (set LEON_HDF5_LIB to its path)";
            return;
        }
    }
    if (g_h5.open() < 0) { g_load_error = "H5open failed"; return; }
    auto var = [&](const char* n) -> hid_t { void* p = sym(n); return p ? *reinterpret_cast<hid_t*>(p) : -1; };
    g_h5.T_U8 = var("H5T_STD_U8LE_g"); g_h5.T_U64 = var("H5T_STD_U64LE_g");
    g_h5.P_DCPL = var("H5P_CLS_DATASET_CREATE_ID_g"); g_h5.P_GCPL = var("H5P_CLS_GROUP_CREATE_ID_g");
    if (!ok) return;
    g_h5.Eset_auto2(0, nullptr, nullptr);                         // errors are reported through exceptions, not on stderr
}
const H5& h5() {
    std::call_once(g_once, load_hdf5);
    if (!g_load_error.empty()) throw Exception(g_load_error);
    return g_h5;
}
constexpr unsigned ACC_RDONLY = 0x0000u, ACC_TRUNC = 0x0002u;

}  // namespace

Container::Container(const std::string& path, Mode mode) : path_(path), writable_(mode == CREATE) {
    const H5& H = h5();
    file_ = mode == CREATE ? H.Fcreate(path.c_str(), ACC_TRUNC, 0, 0) : H.Fopen(path.c_str(), ACC_RDONLY, 0);
    if (file_ < 0) throw Exception(mode == CREATE ? "cannot write " + path : path + " is not a .leon (HDF5) file this build can open");
}
Container::~Container() { try { close(); } catch (...) {} }
void Container::close() {
    if (file_ < 0) return;
    const hid_t f = file_;
    file_ = -1;
    if (g_h5.Fclose(f) < 0) throw Exception("closing " + path_ + " failed");
}
std::string Container::blockPath(const char* group, uint64_t block_id) {
    return std::string(group) + "/" + layout::BLOCK_PREFIX + std::to_string(block_id);
}
void Container::ensureGroups(const std::string& ds) {
    const H5& H = h5();
    for (size_t at = ds.find('/'); at != std::string::npos; at = ds.find('/', at + 1)) {
        const std::string g = ds.substr(0, at);
        if (std::find(groups_.begin(), groups_.end(), g) != groups_.end()) continue;
        if (H.Lexists(file_, g.c_str(), 0) <= 0) {
            const hid_t gcpl = H.Pcreate(H.P_GCPL);
            H.Pset_obj_track_times(gcpl, 0);                     // no timestamps: two runs give the same file
            const hid_t id = H.Gcreate2(file_, g.c_str(), 0, gcpl, 0);
            H.Pclose(gcpl);
            if (id < 0) throw Exception("cannot create group " + g + " in " + path_);
            H.Gclose(id);
        }
        groups_.push_back(g);
    }
}
void Container::put(const std::string& ds, const void* data, uint64_t count, bool u64) {
    if (!writable_ || file_ < 0) throw Exception(path_ + " is not open for writing");
    const H5& H = h5();
    ensureGroups(ds);
    const hsize_t dims[1] = { count };
    const hid_t space = H.Screate_simple(1, dims, nullptr);
    const hid_t dcpl = H.Pcreate(H.P_DCPL);
    H.Pset_obj_track_times(dcpl, 0);
    const hid_t type = u64 ? H.T_U64 : H.T_U8;
    const hid_t d = space < 0 ? -1 : H.Dcreate2(file_, ds.c_str(), type, space, 0, dcpl, 0);
    herr_t rc = d < 0 ? -1 : 0;
    if (rc == 0 && count) rc = H.Dwrite(d, type, 0, 0, 0, data);
    if (d >= 0) H.Dclose(d);
    H.Pclose(dcpl);
    if (space >= 0) H.Sclose(space);
    if (rc < 0) throw Exception("cannot write dataset " + ds + " of " + path_);
}
void Container::putBytes(const std::string& ds, const void* data, uint64_t size) { put(ds, data, size, false); }
void Container::putU64(const std::string& ds, const uint64_t* data, uint64_t count) { put(ds, data, count, true); }
bool Container::exists(const std::string& ds) {
    const H5& H = h5();
    for (size_t at = ds.find('/'); ; at = ds.find('/', at + 1)) {          // H5Lexists wants every ancestor to exist
        const std::string part = at == std::string::npos ? ds : ds.substr(0, at);
        if (H.Lexists(file_, part.c_str(), 0) <= 0) return false;
        if (at == std::string::npos) return true;
    }
}
uint64_t Container::get(const std::string& ds, bool u64, void* out, uint64_t cap, bool size_only) {
    const H5& H = h5();
    if (file_ < 0) throw Exception(path_ + " is closed");
    if (!exists(ds)) throw Exception(path_ + " has no dataset " + ds);
    const hid_t d = H.Dopen2(file_, ds.c_str(), 0);
    if (d < 0) throw Exception("cannot open dataset " + ds + " of " + path_);
    const hid_t space = H.Dget_space(d), ftype = H.Dget_type(d);
    const hssize_t n = space < 0 ? -1 : H.Sget_simple_extent_npoints(space);
    const size_t esz = ftype < 0 ? 0 : H.Tget_size(ftype);
    if (ftype >= 0) H.Tclose(ftype);
    if (space >= 0) H.Sclose(space);
    herr_t rc = (n < 0 || esz != (u64 ? 8u : 1u)) ? -1 : 0;
    if (rc == 0 && !size_only) {
        if ((uint64_t)n > cap) rc = -1;
        else if (n) rc = H.Dread(d, u64 ? H.T_U64 : H.T_U8, 0, 0, 0, out);
    }
    H.Dclose(d);
    if (rc < 0) throw Exception("dataset " + ds + " of " + path_ + " is not the " + (u64 ? "u64" : "byte") + " array it should be");
    return (uint64_t)n;
}
std::vector<uint8_t> Container::getBytes(const std::string& ds) {
    std::vector<uint8_t> v(get(ds, false, nullptr, 0, true));
    get(ds, false, v.data(), v.size(), false);
    return v;
}
std::vector<uint64_t> Container::getU64(const std::string& ds) {
    std::vector<uint64_t> v(get(ds, true, nullptr, 0, true));
    get(ds, true, v.data(), v.size(), false);
    return v;
}

}  // namespace leon_host

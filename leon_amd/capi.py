"""ctypes binding of include/leon_dna.h.  Mirrors the C-ABI one to one; no compute happens in Python."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

ABI_VERSION = 5          # include/leon_dna.h LEON_DNA_ABI_VERSION this binding (Stats, _EXPORTS) was written for
LEON_F_KEEP_TRACE = 1
LEON_F_DICT_ON_DEVICE = 2


class LeonDnaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("leon_dna error %d: %s" % (code, msg))
        self.code = code


def lib_path():
    return os.path.join(_HERE, "lib", "libleon_dna.so")


class _Cfg(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("kmer_size", C.c_uint32), ("reads_per_block", C.c_uint32),
                ("bloom_n_hash", C.c_uint32), ("bloom_block_nbits", C.c_uint32), ("device_id", C.c_int32),
                ("bloom_tai", C.c_uint64), ("random_values", C.POINTER(C.c_uint64)), ("resolve_window", C.c_uint64),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_reads", "n_bases", "n_blocks", "n_anchors", "n_no_anchor", "n_symbols",
                                           "payload_bytes", "resolve_rounds", "resolve_windows")] + \
               [(n, C.c_float) for n in ("ms_pack", "ms_resolve", "ms_sort", "ms_walk", "ms_symbols", "ms_rangecoder",
                                         "ms_d2h", "ms_total")] + \
               [("walk_launches", C.c_uint32), ("reserved", C.c_uint32), ("ms_anchor_wait", C.c_float),
                ("ms_chain_busy", C.c_float)] + \
               [(n, C.c_float) for n in ("ms_exchange", "ms_exchange_call", "ms_emulated", "ms_emulated_lookups")] + \
               [(n, C.c_uint64) for n in ("xch_words_sent", "xch_words_received", "walk_reads", "resolve_chain_reads", "resolve_chain_windows")] + \
               [("ms_resolve_chain", C.c_float), ("ms_gather_call", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if not n.startswith("reserved")}


SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint8), C.c_uint64, C.c_uint32)
# leon_exchange_fn: (user, d_send, send_counts[world], world, &d_recv, &recv_total) -> 0 on success
EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64))
XCH_OFF, XCH_BY_ANCHOR, XCH_EMULATE = 0, 1, 2
# leon_gather_fn: (user, d_buf, part_bytes, world) -> 0 on success
GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32)

_u8p, _u32p, _i32p, _u64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_int32, C.c_uint64))

_EXPORTS = {
    "leon_dna_abi_version": (C.c_int, []),
    "leon_last_error": (C.c_char_p, [C.c_void_p]),
    "leon_dna_ctx_create": (C.c_int, [C.POINTER(_Cfg), C.POINTER(C.c_void_p)]),
    "leon_dna_ctx_destroy": (None, [C.c_void_p]),
    "leon_dna_bloom_nbytes": (C.c_int, [C.c_void_p, _u64p]),
    "leon_dna_bloom_upload": (C.c_int, [C.c_void_p, _u8p, C.c_uint64]),
    "leon_dna_bloom_download": (C.c_int, [C.c_void_p, _u8p, C.c_uint64]),
    "leon_dna_bloom_clear": (C.c_int, [C.c_void_p]),
    "leon_dna_bloom_insert": (C.c_int, [C.c_void_p, _u64p, C.c_uint64]),
    "leon_dna_bloom_insert_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "leon_dna_bloom_device_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), _u64p]),
    "leon_dna_bloom_upload_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "leon_dna_bloom_download_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "leon_dna_bloom_contains4": (C.c_int, [C.c_void_p, _u64p, C.c_uint64, C.c_int, _u8p]),
    "leon_dna_bloom_contains": (C.c_int, [C.c_void_p, _u64p, C.c_uint64, _u8p]),
    "leon_dna_encode_batch": (C.c_int, [C.c_void_p, C.c_void_p, _u64p, C.c_uint64, C.c_uint64, SINK, C.c_void_p]),
    "leon_dna_encode_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, SINK,
                                                C.c_void_p]),
    "leon_dna_decode_blocks": (C.c_int, [C.c_void_p, _u64p, C.c_uint64, _u8p, _u64p, _u32p, _u64p, C.c_uint64, _u8p, C.c_uint64,
                                          _u32p]),
    "leon_host_anchor_dict_decode": (C.c_int, [_u8p, C.c_uint64, C.c_uint64, C.c_uint32, _u64p]),
    "leon_dna_finish": (C.c_int, [C.c_void_p, C.POINTER(_u8p), _u64p, _u64p]),
    "leon_dna_reset_stream": (C.c_int, [C.c_void_p]),
    "leon_dna_reserve": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "leon_header_decode_symbols": (C.c_int, [C.c_void_p, _u8p, _u64p, _u32p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "leon_header_text_from_symbols": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, _u32p, C.c_char_p, C.c_uint64, _u8p, C.c_uint64, _u64p, _u64p, C.c_uint32]),
    "leon_header_symbols_free": (None, [C.c_void_p]),
    "leon_dna_set_shard": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "leon_dna_set_exchange": (C.c_int, [C.c_void_p, C.c_uint32, EXCHANGE, C.c_void_p]),
    "leon_dna_set_gather": (C.c_int, [C.c_void_p, GATHER, C.c_void_p]),
    "leon_dna_debug_walk_order": (C.c_int, [C.c_void_p, C.c_void_p]),
    "leon_kmer_solid_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64,
                                          C.POINTER(C.c_void_p), _u64p, _u64p]),
    "leon_kmer_solid": (C.c_int, [C.c_int, C.c_char_p, _u64p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, _u64p, C.c_uint64,
                                   _u64p, _u64p]),
    "leon_device_free": (None, [C.c_void_p]),
    "leon_host_anchor_dict_encode": (C.c_int, [_u64p, C.c_uint64, C.c_uint32, _u8p, C.c_uint64, _u64p]),
    "leon_dna_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "leon_dna_trace_anchors": (C.c_int, [C.c_void_p, _i32p, _u32p, _u8p, C.c_uint64]),
    "leon_dna_trace_events": (C.c_int, [C.c_void_p, _u8p, C.c_uint64]),
    "leon_dna_anchor_kmers": (C.c_int, [C.c_void_p, _u64p, C.c_uint64]),
    "leon_rc_encode_streams": (C.c_int, [C.c_void_p, _u8p, _u64p, C.c_uint64, _u8p, C.c_uint64, _u64p]),
    "leon_header_encode_batch": (C.c_int, [C.c_void_p, C.c_char_p, _u64p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint64, SINK,
                                            C.c_void_p]),
    "leon_header_encode_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint64,
                                                   SINK, C.c_void_p]),
    "leon_host_header_decode_blocks": (C.c_int, [_u8p, _u64p, _u32p, C.c_uint64, C.c_char_p, C.c_uint64, _u8p, C.c_uint64, _u64p,
                                                  _u64p, C.c_uint32]),
    "leon_header_decode_blocks": (C.c_int, [C.c_void_p, _u8p, _u64p, _u32p, C.c_uint64, C.c_char_p, C.c_uint64, _u8p, C.c_uint64, _u64p,
                                             _u64p, C.c_uint32]),
    "leon_qual_smooth_batch": (C.c_int, [C.c_void_p, C.c_char_p, _u64p, C.c_uint64, _u8p]),
    "leon_qual_smooth_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "leon_kmer_auto_cutoff": (C.c_int, [_u64p, _u32p]),
    "leon_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "leon_device_alloc": (C.c_int, [C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]),
    "leon_device_upload": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]),
    "leon_device_copy": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]),
    "leon_device_download": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]),
    "leon_host_qual_encode_blocks": (C.c_int, [C.c_char_p, _u64p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, SINK, C.c_void_p,
                                                C.c_uint64]),
    "leon_qual_deflate_blocks_device": (C.c_int, [C.c_int, C.c_void_p, _u64p, C.c_uint64, C.c_uint32, SINK, C.c_void_p, C.c_uint64]),
    "leon_qual_deflate_release": (None, []),
    "leon_device_trim": (None, []),
    "leon_host_qual_decode_blocks": (C.c_int, [_u8p, _u64p, _u32p, _u64p, C.c_uint64, _u8p, C.c_uint64, _u64p, C.c_uint32]),
}
EXPORTED_SYMBOLS = tuple(_EXPORTS)
_lib = None


def load_library():
    """dlopen libleon_dna.so and bind every symbol include/leon_dna.h declares.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise LeonDnaError(-2, "%s is missing: build it with leon_amd.build_library() (hipcc, gfx950); "
                               "there is no CPU fallback" % path)
    lib = C.CDLL(path)
    for name, (res, args) in _EXPORTS.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    if lib.leon_dna_abi_version() != ABI_VERSION:
        # (a stale library under a new binding reads wrong stats and misses symbols; a new one under an old binding overruns Stats)
        raise LeonDnaError(-2, "%s has ABI %d, this binding is written for %d: rebuild it (leon_amd.build_library())"
                           % (path, lib.leon_dna_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def _ptr(a, t):
    return a.ctypes.data_as(t)


def kmer_words(k):
    """64-bit words per k-mer at the C-ABI: 1 below k = 32, 2 from 32 to 63"""
    return 2 if k >= 32 else 1


def anchor_dict_decode(payload, n_anchors, kmer_size):
    """Leon::decodeAnchorDict on the host: the anchors' k-mers (n_anchors * W words) from the dictionary stream"""
    lib = load_library()
    out = np.zeros(max(n_anchors * kmer_words(kmer_size), 1), dtype=np.uint64)
    buf = np.frombuffer(bytes(payload) + b"\0", dtype=np.uint8)
    rc = lib.leon_host_anchor_dict_decode(_ptr(buf, _u8p), len(payload), n_anchors, kmer_size, _ptr(out, _u64p))
    if rc:
        raise LeonDnaError(rc, (lib.leon_last_error(None) or b"").decode())
    return out[:n_anchors * kmer_words(kmer_size)]


def host_anchor_dict_encode(kmers, k):
    """the dictionary stream for a list of anchors (host-only entry point; runs without a GPU).
    kmers: flat uint64 array, kmer_words(k) words per anchor."""
    lib = load_library()
    kmers = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1)
    n = len(kmers) // kmer_words(k)
    cap = n * k + 64
    out = np.zeros(cap, dtype=np.uint8)
    size = C.c_uint64()
    rc = lib.leon_host_anchor_dict_encode(_ptr(kmers, _u64p), n, k, _ptr(out, _u8p), cap, C.byref(size))
    if rc:
        raise LeonDnaError(rc, "leon_host_anchor_dict_encode failed")
    return out[:size.value].tobytes()


def _join_blocks(blocks):
    """[(id, payload, n_reads)] -> (payload bytes array, offsets, n_reads array)"""
    pay = np.frombuffer(b"".join(b[1] for b in blocks) + b"\0", dtype=np.uint8)
    off = np.zeros(len(blocks) + 1, dtype=np.uint64)
    if blocks:
        off[1:] = np.cumsum([len(b[1]) for b in blocks])
    nr = np.array([b[2] for b in blocks] or [0], dtype=np.uint32)
    return pay, off, nr


def host_header_decode_blocks(blocks, first_header, n_threads=0):
    """HeaderDecoder on host threads (no GPU): blocks = [(id, payload, n_reads)] -> list of header bytes"""
    lib = load_library()
    if not blocks:
        return []
    pay, off, nr = _join_blocks(blocks)
    total = int(nr[:len(blocks)].sum())
    out_off = np.zeros(total + 1, dtype=np.uint64)
    need = C.c_uint64()
    cap = max(64, 64 * total)
    for _ in range(2):
        out = np.zeros(cap, dtype=np.uint8)
        rc = lib.leon_host_header_decode_blocks(_ptr(pay, _u8p), _ptr(off, _u64p), _ptr(nr, _u32p), len(blocks), first_header,
                                                len(first_header), _ptr(out, _u8p), cap, _ptr(out_off, _u64p), C.byref(need), n_threads)
        if rc != -5:
            break
        cap = need.value
    if rc:
        raise LeonDnaError(rc, (lib.leon_last_error(None) or b"").decode())
    raw = out.tobytes()
    return [raw[int(out_off[i]):int(out_off[i + 1])] for i in range(total)]


def host_qual_encode_blocks(quals, offsets, reads_per_block, zlib_level=-1, n_threads=0, first_block_id=0):
    """lossless quality stream: [(id, zlib payload, n_reads)] per read block (host-only)"""
    lib = load_library()
    blocks = []
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)

    def cb(user, block_id, payload, size, n_reads):
        blocks.append((block_id, C.string_at(payload, size), n_reads))
        return 0
    rc = lib.leon_host_qual_encode_blocks(bytes(quals), _ptr(offsets, _u64p), len(offsets) - 1, reads_per_block, zlib_level, n_threads,
                                          SINK(cb), None, first_block_id)
    if rc:
        raise LeonDnaError(rc, (lib.leon_last_error(None) or b"").decode())
    return blocks


def qual_deflate_blocks_device(d_quals_ptr, offsets, reads_per_block, device_id=0, first_block_id=0):
    """the lossless quality blocks written by the device (RLE deflate + dynamic Huffman): [(id, zlib payload, n_reads)];
    d_quals_ptr = device pointer to the concatenated qualities, offsets on the host"""
    lib = load_library()
    blocks = []
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)

    def cb(user, block_id, payload, size, n_reads):
        blocks.append((block_id, C.string_at(payload, size), n_reads))
        return 0
    rc = lib.leon_qual_deflate_blocks_device(device_id, C.c_void_p(int(d_quals_ptr)), _ptr(offsets, _u64p), len(offsets) - 1, reads_per_block,
                                             SINK(cb), None, first_block_id)
    if rc:
        raise LeonDnaError(rc, (lib.leon_last_error(None) or b"").decode())
    return blocks


def host_qual_decode_blocks(blocks, block_n_bytes, n_threads=0):
    lib = load_library()
    if not blocks:
        return []
    pay, off, nr = _join_blocks(blocks)
    nb = np.ascontiguousarray(block_n_bytes, dtype=np.uint64)
    total, cap = int(nr[:len(blocks)].sum()), int(nb.sum())
    out = np.zeros(cap + 1, dtype=np.uint8)
    out_off = np.zeros(total + 1, dtype=np.uint64)
    rc = lib.leon_host_qual_decode_blocks(_ptr(pay, _u8p), _ptr(off, _u64p), _ptr(nr, _u32p), _ptr(nb, _u64p), len(blocks),
                                          _ptr(out, _u8p), cap, _ptr(out_off, _u64p), n_threads)
    if rc:
        raise LeonDnaError(rc, (lib.leon_last_error(None) or b"").decode())
    raw = out.tobytes()
    return [raw[int(out_off[i]):int(out_off[i + 1])] for i in range(total)]


def kmer_auto_cutoff(histogram):
    lib = load_library()
    h = np.ascontiguousarray(histogram, dtype=np.uint64)
    assert len(h) == 256
    out = C.c_uint32()
    rc = lib.leon_kmer_auto_cutoff(_ptr(h, _u64p), C.byref(out))
    if rc:
        raise LeonDnaError(rc, "leon_kmer_auto_cutoff failed")
    return out.value


def kmer_solid(bases, offsets, k, min_abundance, device_id=0, with_histogram=False, max_keys_per_pass=0):
    """solid canonical k-mers of the reads, counted on the device (host arrays in and out); flat uint64, kmer_words(k) each"""
    lib = load_library()
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    if not isinstance(bases, (bytes, bytearray)):
        bases = np.ascontiguousarray(bases, dtype=np.uint8).tobytes()
    w = kmer_words(k)
    hist = np.zeros(256, dtype=np.uint64)
    ns = C.c_uint64()
    cap = max(int(offsets[-1] - offsets[0]), 1)
    out = np.zeros(cap * w, dtype=np.uint64)
    rc = lib.leon_kmer_solid(device_id, bases, _ptr(offsets, _u64p), n, k, min_abundance, int(max_keys_per_pass),
                             _ptr(out, _u64p), cap, C.byref(ns), _ptr(hist, _u64p))
    if rc:
        raise LeonDnaError(rc, (lib.leon_last_error(None) or b"").decode())
    res = out[:ns.value * w].copy()
    return (res, hist) if with_histogram else res


def kmer_solid_device(d_bases_ptr, d_offsets_ptr, n_reads, k, min_abundance, device_id=0, max_keys_per_pass=0):
    """device arrays in, device array out: returns (device pointer, n_solid); free with device_free()"""
    lib = load_library()
    p, ns = C.c_void_p(), C.c_uint64()
    rc = lib.leon_kmer_solid_device(device_id, C.c_void_p(int(d_bases_ptr)), C.c_void_p(int(d_offsets_ptr)), int(n_reads), k,
                                    min_abundance, int(max_keys_per_pass), C.byref(p), C.byref(ns), None)
    if rc:
        raise LeonDnaError(rc, (lib.leon_last_error(None) or b"").decode())
    return p.value, ns.value


def device_free(ptr):
    if ptr:
        load_library().leon_device_free(C.c_void_p(int(ptr)))


def device_trim():
    """return the device memory the library parks between calls (the k-mer counter's ~22 GB of sort buffers after a large count) to the
    driver: for hosts that go on allocating through torch or their own hipMalloc with no context's leon_dna_reserve in between"""
    load_library().leon_device_trim()


def device_copy(d_dst, d_src, n_bytes, device_id=0):
    """device to device, n_bytes from d_src to d_dst (raw device pointers)"""
    rc = load_library().leon_device_copy(device_id, C.c_void_p(int(d_dst)), C.c_void_p(int(d_src)), int(n_bytes))
    if rc:
        raise LeonDnaError(rc, (load_library().leon_last_error(None) or b"").decode())


def device_upload_bytes(data, device_id=0):
    """a new device buffer holding `data` (bytes-like); the caller frees it with device_free"""
    lib = load_library()
    data = bytes(data)
    p = C.c_void_p()
    rc = lib.leon_device_alloc(device_id, len(data) + 64, C.byref(p))
    if rc == 0 and data:
        rc = lib.leon_device_upload(device_id, p, C.c_char_p(data), len(data))
    if rc:
        raise LeonDnaError(rc, (lib.leon_last_error(None) or b"").decode())
    return p.value


class DnaEncodeContext:
    """One ordered read stream on one GPU (wraps leon_dna_ctx)."""

    def __init__(self, kmer_size=31, reads_per_block=50000, bloom_tai=0, bloom_n_hash=7, bloom_block_nbits=12,
                 device_id=0, resolve_window=0, keep_trace=False, random_values=None, dict_on_device=False):
        self.lib = load_library()
        cfg = _Cfg()
        cfg.struct_size = C.sizeof(_Cfg)
        cfg.kmer_size, cfg.reads_per_block = kmer_size, reads_per_block
        cfg.bloom_n_hash, cfg.bloom_block_nbits, cfg.device_id = bloom_n_hash, bloom_block_nbits, device_id
        cfg.bloom_tai, cfg.resolve_window = int(bloom_tai), int(resolve_window)
        cfg.flags = (LEON_F_KEEP_TRACE if keep_trace else 0) | (LEON_F_DICT_ON_DEVICE if dict_on_device else 0)
        self._rv = None
        if random_values is not None:
            self._rv = np.ascontiguousarray(random_values, dtype=np.uint64)
            assert len(self._rv) == 256
            cfg.random_values = _ptr(self._rv, _u64p)
        h = C.c_void_p()
        rc = self.lib.leon_dna_ctx_create(C.byref(cfg), C.byref(h))
        if rc:
            raise LeonDnaError(rc, (self.lib.leon_last_error(None) or b"").decode())
        self.h = h
        self.kmer_size, self.reads_per_block = kmer_size, reads_per_block
        self.bloom_tai, self.bloom_n_hash, self.bloom_block_nbits = int(bloom_tai), bloom_n_hash, bloom_block_nbits
        self.next_read = 0
        self._hdr_next = 0
        self._xch_error = None               # what a Python exchange / gather callback raised inside the last call (nothing may cross the C boundary)

    def _chk(self, rc):
        cause, self._xch_error = getattr(self, "_xch_error", None), None
        if rc:
            err = LeonDnaError(rc, (self.lib.leon_last_error(self.h) or b"").decode())
            if cause is not None:            # the callback's own exception is the cause, not lost behind "the callback returned non-zero"
                raise err from cause
            raise err

    def close(self):
        if getattr(self, "h", None):
            self.lib.leon_dna_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    # ---- bloom ----
    @property
    def bloom_nbytes(self):
        n = C.c_uint64()
        self._chk(self.lib.leon_dna_bloom_nbytes(self.h, C.byref(n)))
        return n.value

    def bloom_upload(self, bits):
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        self._chk(self.lib.leon_dna_bloom_upload(self.h, _ptr(bits, _u8p), len(bits)))

    def bloom_download(self):
        out = np.zeros(self.bloom_nbytes, dtype=np.uint8)
        self._chk(self.lib.leon_dna_bloom_download(self.h, _ptr(out, _u8p), len(out)))
        return out

    def bloom_clear(self):
        self._chk(self.lib.leon_dna_bloom_clear(self.h))

    def bloom_insert(self, kmers):
        """kmers: flat uint64 array, kmer_words(k) words per k-mer"""
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1)
        self._chk(self.lib.leon_dna_bloom_insert(self.h, _ptr(kmers, _u64p), len(kmers) // kmer_words(self.kmer_size)))

    def bloom_insert_device(self, dev_ptr, n):
        self._chk(self.lib.leon_dna_bloom_insert_device(self.h, C.c_void_p(int(dev_ptr)), int(n)))

    def bloom_device_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.lib.leon_dna_bloom_device_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def bloom_upload_device(self, dev_ptr, n):
        self._chk(self.lib.leon_dna_bloom_upload_device(self.h, C.c_void_p(int(dev_ptr)), int(n)))

    def bloom_download_device(self, dev_ptr, n):
        self._chk(self.lib.leon_dna_bloom_download_device(self.h, C.c_void_p(int(dev_ptr)), int(n)))

    def bloom_contains4(self, kmers, right):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1)
        n = len(kmers) // kmer_words(self.kmer_size)
        out = np.zeros(n, dtype=np.uint8)
        self._chk(self.lib.leon_dna_bloom_contains4(self.h, _ptr(kmers, _u64p), n, int(right), _ptr(out, _u8p)))
        return out

    def bloom_contains(self, kmers):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1)
        n = len(kmers) // kmer_words(self.kmer_size)
        out = np.zeros(n, dtype=np.uint8)
        self._chk(self.lib.leon_dna_bloom_contains(self.h, _ptr(kmers, _u64p), n, _ptr(out, _u8p)))
        return out

    # ---- encode ----
    def _collect_sink(self, blocks):
        def cb(user, block_id, payload, size, n_reads):
            blocks.append((block_id, C.string_at(payload, size), n_reads))
            return 0
        return SINK(cb)

    def encode_batch(self, bases, offsets, sink=None):
        """bases: bytes / uint8 array, offsets: uint64[n+1].  Returns [(block_id, payload, n_reads)]."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        if isinstance(bases, (bytes, bytearray)):
            keep = (C.c_char * len(bases)).from_buffer_copy(bases) if isinstance(bases, bytes) else (C.c_char * len(bases)).from_buffer(bases)
            ptr = C.cast(keep, C.c_void_p)
        else:                                                   # a uint8 array is handed over in place (no copy)
            keep = np.ascontiguousarray(bases, dtype=np.uint8)
            ptr = C.c_void_p(keep.ctypes.data)
        blocks = []
        cb = sink if sink is not None else self._collect_sink(blocks)
        self._chk(self.lib.leon_dna_encode_batch(self.h, ptr, _ptr(offsets, _u64p), n, self.next_read, cb, None))
        self.next_read += n
        return blocks

    def encode_batch_device(self, d_bases_ptr, d_offsets_ptr, n_reads, sink=None):
        blocks = []
        cb = sink if sink is not None else self._collect_sink(blocks)
        self._chk(self.lib.leon_dna_encode_batch_device(self.h, C.c_void_p(int(d_bases_ptr)), C.c_void_p(int(d_offsets_ptr)),
                                                        int(n_reads), self.next_read, cb, None))
        self.next_read += int(n_reads)
        return blocks

    def header_encode_batch(self, headers, first_header=None, sink=None):
        """HeaderEncoder over a batch of header texts (list of bytes, file order) -> [(block_id, payload, n_reads)]"""
        blob = b"".join(headers)
        off = np.zeros(len(headers) + 1, dtype=np.uint64)
        if headers:
            off[1:] = np.cumsum([len(h) for h in headers])
        if first_header is None:
            first_header = headers[0] if headers else b""
        blocks = []
        cb = sink or self._collect_sink(blocks)
        self._chk(self.lib.leon_header_encode_batch(self.h, blob, _ptr(off, _u64p), len(headers), self._hdr_next, first_header,
                                                    len(first_header), cb, None))
        self._hdr_next += len(headers)
        return blocks

    def header_decode_blocks(self, blocks, first_header, n_threads=0):
        """HeaderDecoder with the symbols decoded on the device and the text rebuilt on host threads:
        blocks = [(id, payload, n_reads)] -> list of header bytes (same results as host_header_decode_blocks)"""
        if not blocks:
            return []
        pay, off, nr = _join_blocks(blocks)
        total = int(nr[:len(blocks)].sum())
        out_off = np.zeros(total + 1, dtype=np.uint64)
        need = C.c_uint64()
        cap = max(64, 64 * total)
        for _ in range(2):
            out = np.zeros(cap, dtype=np.uint8)
            rc = self.lib.leon_header_decode_blocks(self.h, _ptr(pay, _u8p), _ptr(off, _u64p), _ptr(nr, _u32p), len(blocks), first_header,
                                                    len(first_header), _ptr(out, _u8p), cap, _ptr(out_off, _u64p), C.byref(need), n_threads)
            if rc != -5:
                break
            cap = need.value
        self._chk(rc)
        raw = out.tobytes()
        return [raw[int(out_off[i]):int(out_off[i + 1])] for i in range(total)]

    def header_symbol_set(self, blocks):
        """the device half of header_decode_blocks alone: the symbols of ALL of `blocks` in one device call.  Returns an object whose
        .text(first_block, n_blocks, first_header) gives those blocks' headers (host threads) and .close() frees the set"""
        ctx = self
        pay, off, nr = _join_blocks(blocks)
        h = C.c_void_p()
        self._chk(self.lib.leon_header_decode_symbols(self.h, _ptr(pay, _u8p), _ptr(off, _u64p), _ptr(nr, _u32p), len(blocks), C.byref(h)))

        class Set:
            def text(self, first_block, n_blocks, first_header, n_threads=0):
                counts = np.ascontiguousarray(nr[first_block:first_block + n_blocks], dtype=np.uint32)
                total = int(counts.sum())
                out_off = np.zeros(total + 1, dtype=np.uint64)
                need = C.c_uint64()
                cap = max(64, 64 * total)
                for _ in range(2):
                    out = np.zeros(cap, dtype=np.uint8)
                    rc = ctx.lib.leon_header_text_from_symbols(h, first_block, n_blocks, _ptr(counts, _u32p), first_header, len(first_header),
                                                               _ptr(out, _u8p), cap, _ptr(out_off, _u64p), C.byref(need), n_threads)
                    if rc != -5:
                        break
                    cap = need.value
                if rc:
                    raise LeonDnaError(rc, (ctx.lib.leon_last_error(None) or b"").decode())
                raw = out.tobytes()
                return [raw[int(out_off[i]):int(out_off[i + 1])] for i in range(total)]

            def close(self):
                ctx.lib.leon_header_symbols_free(h)
        return Set()

    def qual_smooth_batch(self, bases, offsets, quals):
        """DnaEncoder::smoothQuals (lossy qualities) over a batch: returns the smoothed quality bytes (same offsets)"""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        q = np.frombuffer(bytes(quals), dtype=np.uint8).copy()
        if len(offsets) > 1:
            self._chk(self.lib.leon_qual_smooth_batch(self.h, bytes(bases), _ptr(offsets, _u64p), len(offsets) - 1, _ptr(q, _u8p)))
        return q.tobytes()

    def finish(self, copy=True):
        """(dictionary stream, number of anchors); copy=False returns its size instead of a bytes copy (the C caller
        gets a pointer into the context and writes from there: a 100 M-read file's stream is 110 MB)"""
        p, sz, na = _u8p(), C.c_uint64(), C.c_uint64()
        self._chk(self.lib.leon_dna_finish(self.h, C.byref(p), C.byref(sz), C.byref(na)))
        return (C.string_at(p, sz.value) if copy else sz.value), na.value

    def decode_blocks(self, anchors, blocks, block_n_bases):
        """DnaDecoder over read blocks: blocks = [(block_id, payload, n_reads)] as the encoder's sink delivered them,
        anchors = the dictionary (anchor_dict_decode), block_n_bases = bases per block.  Returns the list of reads (bytes)."""
        out, lens = self.decode_blocks_raw(anchors, blocks, block_n_bases)
        ends = np.cumsum(lens, dtype=np.uint64)
        raw = out.tobytes()
        return [raw[int(e) - int(l):int(e)] for e, l in zip(ends, lens)]

    def decode_blocks_raw(self, anchors, blocks, block_n_bases):
        """same, returning (bases back to back as a uint8 array, read lengths as a uint32 array)"""
        blocks = sorted(blocks)
        nb = len(blocks)
        anchors = np.ascontiguousarray(anchors, dtype=np.uint64)
        W = kmer_words(self.kmer_size)
        pay = np.frombuffer(b"".join(b[1] for b in blocks) + b"\0", dtype=np.uint8)
        off = np.zeros(nb + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(b[1]) for b in blocks], dtype=np.uint64)
        nreads = np.array([b[2] for b in blocks], dtype=np.uint32)
        nbases = np.ascontiguousarray(block_n_bases, dtype=np.uint64)
        total, n_total = int(nbases.sum()), int(nreads.sum())
        out = np.zeros(total + 1, dtype=np.uint8)
        lens = np.zeros(n_total + 1, dtype=np.uint32)
        self._chk(self.lib.leon_dna_decode_blocks(self.h, _ptr(anchors, _u64p), len(anchors) // W, _ptr(pay, _u8p), _ptr(off, _u64p),
                                                  _ptr(nreads, _u32p), _ptr(nbases, _u64p), nb, _ptr(out, _u8p), total,
                                                  _ptr(lens, _u32p)))
        return out[:total], lens[:n_total]

    def reserve(self, max_reads, max_bases):
        self._chk(self.lib.leon_dna_reserve(self.h, int(max_reads), int(max_bases)))

    def set_shard(self, rank, world):
        self._chk(self.lib.leon_dna_set_shard(self.h, rank, world))

    def set_exchange(self, mode, fn=None):
        """how the walk is divided among the ranks of set_shard: XCH_OFF (by block range), XCH_BY_ANCHOR with `fn(d_send, send_counts) ->
        (d_recv, recv_total)` doing the all-to-all of the 64-bit words in device memory, XCH_EMULATE (this context plays every rank's slice)"""
        if fn is None:
            self._xch_cb = C.cast(None, EXCHANGE)
        else:
            def thunk(user, d_send, counts, world, d_recv, recv_total):
                try:
                    ptr, total = fn(int(d_send or 0), [int(counts[i]) for i in range(world)])
                    d_recv[0] = C.c_void_p(ptr)
                    recv_total[0] = int(total)
                    return 0
                except Exception as e:                 # noqa: BLE001 -- nothing may cross the C boundary
                    self._xch_error = e
                    return 1
            self._xch_cb = EXCHANGE(thunk)
        self._chk(self.lib.leon_dna_set_exchange(self.h, mode, self._xch_cb, None))

    def set_gather(self, fn=None):
        """the window look-ups of the anchor resolution divided among the ranks too: `fn(d_buf, part_bytes, world)` all-gathers the
        device buffer in place (part r at d_buf + r * part_bytes, this rank's part filled on entry)"""
        if fn is None:
            self._gather_cb = C.cast(None, GATHER)
        else:
            def thunk(user, d_buf, part_bytes, world):
                try:
                    fn(int(d_buf or 0), int(part_bytes), int(world))
                    return 0
                except Exception as e:                 # noqa: BLE001 -- nothing may cross the C boundary
                    self._xch_error = e
                    return 1
            self._gather_cb = GATHER(thunk)
        self._chk(self.lib.leon_dna_set_gather(self.h, self._gather_cb, None))

    def reset_stream(self):
        self._hdr_next = 0
        self._chk(self.lib.leon_dna_reset_stream(self.h))
        self.next_read = 0

    def stats(self):
        s = Stats()
        self._chk(self.lib.leon_dna_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    # ---- traces ----
    def trace_anchors(self, n_reads):
        pos = np.zeros(n_reads, dtype=np.int32)
        addr = np.zeros(n_reads, dtype=np.uint32)
        flags = np.zeros(n_reads, dtype=np.uint8)
        self._chk(self.lib.leon_dna_trace_anchors(self.h, _ptr(pos, _i32p), _ptr(addr, _u32p), _ptr(flags, _u8p), n_reads))
        return pos, addr, flags

    def trace_events(self, n_bases):
        ev = np.zeros(n_bases, dtype=np.uint8)
        self._chk(self.lib.leon_dna_trace_events(self.h, _ptr(ev, _u8p), n_bases))
        return ev

    def anchor_kmers(self, n):
        w = kmer_words(self.kmer_size)
        out = np.zeros(max(n, 1) * w, dtype=np.uint64)
        self._chk(self.lib.leon_dna_anchor_kmers(self.h, _ptr(out, _u64p), n))
        return out[:n * w]

    def rc_encode_streams(self, syms, begin):
        """syms: uint8[2*n] (model, value) pairs; begin: uint64[n_streams+1].  Returns list of payload bytes."""
        syms = np.ascontiguousarray(syms, dtype=np.uint8)
        begin = np.ascontiguousarray(begin, dtype=np.uint64)
        ns = len(begin) - 1
        cap = 3 * (len(syms) // 2) + 64 * (ns + 1)
        out = np.zeros(cap, dtype=np.uint8)
        sizes = np.zeros(max(ns, 1), dtype=np.uint64)
        self._chk(self.lib.leon_rc_encode_streams(self.h, _ptr(syms, _u8p), _ptr(begin, _u64p), ns, _ptr(out, _u8p), cap,
                                                  _ptr(sizes, _u64p)))
        res, w = [], 0
        for i in range(ns):
            res.append(out[w:w + int(sizes[i])].tobytes())
            w += int(sizes[i])
        return res

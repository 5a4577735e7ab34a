"""ctypes binding of oracle/_build/libleon_oracle.so (the CPU restatement; test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "libleon_oracle.so")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


def _load():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_ROOT, "oracle", "leon_oracle.c")):
        build()
    L = C.CDLL(_SO)
    sig = {
        "lo_revcomp": (C.c_uint64, [C.c_uint64, C.c_uint32]),
        "lo_canonical": (C.c_uint64, [C.c_uint64, C.c_uint32]),
        "lo_hash64": (C.c_uint64, [C.c_uint64, C.c_uint64]),
        "lo_hash_seed": (C.c_uint64, [C.c_uint32]),
        "lo_random_value": (C.c_uint64, [C.c_uint32]),
        "lo_bloom_new": (C.c_void_p, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]),
        "lo_bloom_free": (None, [C.c_void_p]),
        "lo_bloom_insert": (None, [C.c_void_p, u64p, C.c_uint64]),
        "lo_bloom_contains": (C.c_int, [C.c_void_p, C.c_uint64]),
        "lo_bloom_contains4": (C.c_uint, [C.c_void_p, C.c_uint64, C.c_int]),
        "lo_bloom_contains_w": (C.c_int, [C.c_void_p, u64p]),
        "lo_bloom_contains4_w": (C.c_uint, [C.c_void_p, u64p, C.c_int]),
        "lo_bloom_bits": (u8p, [C.c_void_p]),
        "lo_bloom_nbytes": (C.c_uint64, [C.c_void_p]),
        "lo_bloom_tai": (C.c_uint64, [C.c_void_p]),
        "lo_bloom_reduced_tai": (C.c_uint64, [C.c_void_p]),
        "lo_encoder_new": (C.c_void_p, [C.c_uint32, C.c_uint32, C.c_void_p]),
        "lo_encoder_free": (None, [C.c_void_p]),
        "lo_encoder_add_reads": (C.c_int, [C.c_void_p, C.c_char_p, u64p, C.c_uint64]),
        "lo_encoder_finish": (C.c_int, [C.c_void_p]),
        "lo_encoder_n_reads": (C.c_uint64, [C.c_void_p]),
        "lo_encoder_n_blocks": (C.c_uint64, [C.c_void_p]),
        "lo_encoder_block": (u8p, [C.c_void_p, C.c_uint64, u64p, u32p]),
        "lo_encoder_anchor_dict": (u8p, [C.c_void_p, u64p, u64p]),
        "lo_encoder_anchor_kmers": (u64p, [C.c_void_p]),
        "lo_encoder_read_anchor_pos": (i32p, [C.c_void_p]),
        "lo_encoder_read_anchor_addr": (u32p, [C.c_void_p]),
        "lo_encoder_read_flags": (u8p, [C.c_void_p]),
        "lo_encoder_events": (u8p, [C.c_void_p, u64p]),
        "lo_encoder_n_symbols": (C.c_uint64, [C.c_void_p]),
        "lo_decode_anchor_dict": (C.c_int, [u8p, C.c_uint64, C.c_uint64, C.c_uint32, u64p]),
        "lo_decode_block": (C.c_int64, [C.c_uint32, C.c_void_p, u64p, C.c_uint64, u8p, C.c_uint64, C.c_uint32,
                                        C.c_char_p, C.c_uint64, u32p]),
        "lo_count_solid": (C.c_uint64, [C.c_char_p, u64p, C.c_uint64, C.c_uint32, C.c_uint32, u64p, C.c_uint64]),
        "lo_header_encode_block": (C.c_int, [C.c_char_p, u64p, C.c_uint64, C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p), u64p,
                                             C.POINTER(C.c_void_p), u64p]),
        "lo_header_decode_block": (C.c_int64, [u8p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, u64p]),
        "lo_free": (None, [C.c_void_p]),
        "lo_qual_smooth": (None, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32, u8p]),
        "lo_rc_new": (C.c_void_p, []),
        "lo_rc_free": (None, [C.c_void_p]),
        "lo_rc_encode_stream": (C.c_int, [C.c_void_p, u8p, u8p, C.c_uint64, u32p, C.c_uint32]),
        "lo_rc_bytes": (u8p, [C.c_void_p, u64p]),
        "lo_rc_decode_stream": (C.c_int, [u8p, C.c_uint64, u8p, u8p, C.c_uint64, u32p, C.c_uint32]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    return L


lib = _load()


def _p(a, t):
    return a.ctypes.data_as(t)


def _copy(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def kwords(k):
    """64-bit words per k-mer: upstream's LargeInt<1> below k = 32, LargeInt<2> for 32 <= k < 64"""
    return 2 if k >= 32 else 1


def kmer_words(x, k):
    """python int k-mer -> uint64 array of kwords(k) words (low word first)"""
    x = int(x)
    return np.array([x & 0xFFFFFFFFFFFFFFFF, x >> 64][:kwords(k)], dtype=np.uint64)


def kmers_to_ints(words, k):
    """flat word array (kwords(k) per k-mer) -> list of python ints"""
    w = kwords(k)
    a = np.asarray(words, dtype=np.uint64).reshape(-1, w)
    return [int(r[0]) | (int(r[1]) << 64 if w == 2 else 0) for r in a]


def reads_to_arrays(reads):
    """list of str/bytes -> (bases bytes, offsets uint64[n+1])"""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs])
    return b"".join(bs), off


class Bloom:
    def __init__(self, tai_bloom, k, n_hash=7, block_nbits=12):
        self.h = lib.lo_bloom_new(int(tai_bloom), k, n_hash, block_nbits)
        if not self.h:
            raise ValueError("bad bloom parameters")
        self.k, self.n_hash, self.block_nbits, self.tai_bloom = k, n_hash, block_nbits, int(tai_bloom)

    def insert(self, kmers):
        """kmers: flat uint64 array, kwords(k) words per k-mer"""
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1)
        lib.lo_bloom_insert(self.h, _p(kmers, u64p), len(kmers) // kwords(self.k))

    def contains(self, kmer):
        w = kmer_words(kmer, self.k)
        return bool(lib.lo_bloom_contains_w(self.h, _p(w, u64p)))

    def contains4(self, kmer, right):
        w = kmer_words(kmer, self.k)
        return int(lib.lo_bloom_contains4_w(self.h, _p(w, u64p), int(right)))

    @property
    def bits(self):
        n = lib.lo_bloom_nbytes(self.h)
        return _copy(lib.lo_bloom_bits(self.h), n, np.uint8)

    def set_bits(self, arr):
        n = lib.lo_bloom_nbytes(self.h)
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        assert len(arr) == n
        C.memmove(lib.lo_bloom_bits(self.h), arr.ctypes.data, n)

    @property
    def tai(self):
        return lib.lo_bloom_tai(self.h)

    @property
    def reduced_tai(self):
        return lib.lo_bloom_reduced_tai(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib.lo_bloom_free(self.h)
            self.h = None


class EncodeResult:
    pass


def encode(bases, offsets, k, reads_per_block, bloom, trace=True):
    """Sequential (-nb-cores 1) DNA encode of all reads; returns blocks, anchor dict and traces."""
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    e = lib.lo_encoder_new(k, reads_per_block, bloom.h)
    if not e:
        raise ValueError("bad encoder parameters")
    try:
        rc = lib.lo_encoder_add_reads(e, bases, _p(offsets, u64p), n)
        assert rc == 0
        lib.lo_encoder_finish(e)
        res = EncodeResult()
        res.n_reads = lib.lo_encoder_n_reads(e)
        nb = lib.lo_encoder_n_blocks(e)
        res.blocks, res.block_nreads = [], []
        for i in range(nb):
            sz, nr = C.c_uint64(), C.c_uint32()
            p = lib.lo_encoder_block(e, i, C.byref(sz), C.byref(nr))
            res.blocks.append(_copy(p, sz.value, np.uint8).tobytes())
            res.block_nreads.append(nr.value)
        sz, na = C.c_uint64(), C.c_uint64()
        p = lib.lo_encoder_anchor_dict(e, C.byref(sz), C.byref(na))
        res.anchor_dict = _copy(p, sz.value, np.uint8).tobytes()
        res.n_anchors = na.value
        res.anchor_kmers = _copy(lib.lo_encoder_anchor_kmers(e), na.value * kwords(k), np.uint64)
        res.n_symbols = lib.lo_encoder_n_symbols(e)
        if trace:
            res.anchor_pos = _copy(lib.lo_encoder_read_anchor_pos(e), n, np.int32)
            res.anchor_addr = _copy(lib.lo_encoder_read_anchor_addr(e), n, np.uint32)
            res.flags = _copy(lib.lo_encoder_read_flags(e), n, np.uint8)
            tot = C.c_uint64()
            p = lib.lo_encoder_events(e, C.byref(tot))
            res.events = _copy(p, tot.value, np.uint8)
        return res
    finally:
        lib.lo_encoder_free(e)


def decode_anchor_dict(payload, n_anchors, k):
    w = kwords(k)
    out = np.zeros(max(n_anchors, 1) * w, dtype=np.uint64)
    buf = np.frombuffer(payload, dtype=np.uint8)
    lib.lo_decode_anchor_dict(_p(buf, u8p), len(buf), n_anchors, k, _p(out, u64p))
    return out[:n_anchors * w]


def decode_block(k, bloom, anchors, payload, n_reads, max_bases):
    anchors = np.ascontiguousarray(anchors, dtype=np.uint64).reshape(-1)
    if len(anchors) == 0:
        anchors = np.zeros(2, dtype=np.uint64)
        na = 0
    else:
        na = len(anchors) // kwords(k)
    buf = np.frombuffer(payload, dtype=np.uint8)
    out = C.create_string_buffer(int(max_bases) + 1)
    lens = np.zeros(max(n_reads, 1), dtype=np.uint32)
    w = lib.lo_decode_block(k, bloom.h, _p(anchors, u64p), na, _p(buf, u8p), len(buf), n_reads,
                            out, int(max_bases), _p(lens, u32p))
    if w < 0:
        raise ValueError("decode error %d" % w)
    raw = out.raw[:w]
    reads, o = [], 0
    for i in range(n_reads):
        reads.append(raw[o:o + int(lens[i])])
        o += int(lens[i])
    return reads


def count_solid(bases, offsets, k, min_abundance):
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    w = kwords(k)
    ns = lib.lo_count_solid(bases, _p(offsets, u64p), n, k, min_abundance, None, 0)
    out = np.zeros(max(ns, 1) * w, dtype=np.uint64)
    lib.lo_count_solid(bases, _p(offsets, u64p), n, k, min_abundance, _p(out, u64p), ns)
    return out[:ns * w]                      # kwords(k) words per k-mer


def rc_encode_stream(models, syms, model_sizes):
    models = np.ascontiguousarray(models, dtype=np.uint8)
    syms = np.ascontiguousarray(syms, dtype=np.uint8)
    sizes = np.ascontiguousarray(model_sizes, dtype=np.uint32)
    r = lib.lo_rc_new()
    try:
        rc = lib.lo_rc_encode_stream(r, _p(models, u8p), _p(syms, u8p), len(syms), _p(sizes, u32p), len(sizes))
        if rc:
            raise ValueError("bad symbol stream")
        sz = C.c_uint64()
        p = lib.lo_rc_bytes(r, C.byref(sz))
        return _copy(p, sz.value, np.uint8).tobytes()
    finally:
        lib.lo_rc_free(r)


def rc_decode_stream(payload, models, model_sizes):
    models = np.ascontiguousarray(models, dtype=np.uint8)
    sizes = np.ascontiguousarray(model_sizes, dtype=np.uint32)
    buf = np.frombuffer(payload, dtype=np.uint8)
    out = np.zeros(len(models), dtype=np.uint8)
    lib.lo_rc_decode_stream(_p(buf, u8p), len(buf), _p(models, u8p), _p(out, u8p), len(models), _p(sizes, u32p),
                            len(sizes))
    return out


def header_encode_block(headers, first, with_trace=False):
    """HeaderEncoder over one block: list of bytes -> payload bytes (and the (model id, symbol) trace as an (n, 2) array)"""
    blob, off = reads_to_arrays(headers)
    pay, tr = C.c_void_p(), C.c_void_p()
    sz, tsz = C.c_uint64(), C.c_uint64()
    lib.lo_header_encode_block(blob, _p(off, u64p), len(headers), first, len(first), C.byref(pay), C.byref(sz),
                               C.byref(tr) if with_trace else None, C.byref(tsz))
    payload = C.string_at(pay, sz.value) if sz.value else b""
    lib.lo_free(pay)
    if not with_trace:
        return payload
    trace = np.frombuffer(C.string_at(tr, tsz.value), dtype=np.uint8).reshape(-1, 2).copy() if tsz.value else np.zeros((0, 2), np.uint8)
    lib.lo_free(tr)
    return payload, trace


def header_decode_block(payload, n, first, max_bytes):
    buf = np.frombuffer(payload, dtype=np.uint8)
    out = C.create_string_buffer(int(max_bytes) + 1)
    off = np.zeros(n + 1, dtype=np.uint64)
    w = lib.lo_header_decode_block(_p(buf, u8p), len(buf), n, first, len(first), out, int(max_bytes), _p(off, u64p))
    if w < 0:
        raise ValueError("header block does not decode")
    raw = out.raw[:w]
    return [raw[int(off[i]):int(off[i + 1])] for i in range(n)]


def qual_smooth(bloom, k, seq, qual):
    """DnaEncoder::smoothQuals on one read: bytes in, bytes out"""
    q = np.frombuffer(bytes(qual), dtype=np.uint8).copy()
    if len(q):
        lib.lo_qual_smooth(bloom.h, k, bytes(seq), len(seq), _p(q, u8p))
    return q.tobytes()

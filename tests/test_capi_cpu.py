"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/leon_dna.h declares,
and fails loudly without a GPU (no CPU fallback).  No compute is run here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "leon_dna.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(leon_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if n != "leon_block_sink"))


@pytest.fixture(scope="module")
def lib():
    import leon_amd
    if not os.path.exists(leon_amd.lib_path()):
        leon_amd.build_library()
    return leon_amd.load_library()


def test_every_declared_symbol_is_exported(lib):
    from leon_amd import capi
    declared = _declared_functions()
    assert len(declared) >= 20
    raw = ctypes.CDLL(capi.lib_path())
    for name in declared:
        assert hasattr(raw, name), "libleon_dna.so does not export " + name
    assert sorted(capi.EXPORTED_SYMBOLS) == declared, "the Python binding and the header disagree"
    assert lib.leon_dna_abi_version() == 5 == capi.ABI_VERSION


def test_no_cpu_fallback(lib):
    import torch
    import leon_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(leon_amd.LeonDnaError) as e:
        leon_amd.DnaEncodeContext(kmer_size=31, bloom_tai=1000)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_config_validation_messages(lib):
    import leon_amd
    for kw in (dict(kmer_size=64), dict(kmer_size=2), dict(reads_per_block=0), dict(bloom_n_hash=0)):
        with pytest.raises(leon_amd.LeonDnaError) as e:
            leon_amd.DnaEncodeContext(bloom_tai=1000, **kw)
        assert e.value.code == -1
    # bloom_tai = 0 would make BloomNeighborCoherent's modulus non-positive (every probe out of bounds): refused up front
    for tai in (0, 1 << 50):
        with pytest.raises(leon_amd.LeonDnaError) as e:
            leon_amd.DnaEncodeContext(bloom_tai=tai)
        assert e.value.code == -1 and "bloom_tai" in str(e.value)


def test_product_does_not_touch_the_oracle():
    # the product tree must not import, link or call anything under oracle/
    for d, _, files in os.walk(os.path.join(ROOT, "leon_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "leon_oracle" not in txt and "oracle_lib" not in txt, os.path.join(d, f)
    assert "oracle" not in open(os.path.join(ROOT, "include", "leon_dna.h")).read().lower()


def test_host_anchor_dict_stream_matches_oracle(lib):
    """leon_dna_finish's dictionary stream is coded on a host thread (one serial chain per file); its coder runs
    without a GPU and must equal the oracle's range coder over the anchors' bases on _anchorDictModel(5)."""
    import numpy as np
    import oracle_lib as O
    from leon_amd import capi
    rng = np.random.default_rng(3)
    sizes = [2, 5, 5, 2, 3, 3, 3, 2] + [256] * 72
    for k, n in [(31, 0), (31, 1), (31, 700), (21, 3000), (5, 10), (32, 50), (33, 4000), (47, 2500), (63, 900), (31, 30000), (63, 20000)]:   # the last two: 29 and 39 segments of records through the feed (its ring of 512; smaller rings below)
        ints = [(int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 1 << 62)) << 62)) & ((1 << (2 * k)) - 1) for _ in range(n)]
        w = O.kwords(k)
        kmers = np.array([[x & 0xFFFFFFFFFFFFFFFF, x >> 64][:w] for x in ints], dtype=np.uint64).reshape(-1)
        syms = np.array([(x >> (2 * (k - 1 - i))) & 3 for x in ints for i in range(k)], dtype=np.uint8)
        exp = O.rc_encode_stream(np.ones(len(syms), dtype=np.uint8), syms, sizes)   # model 1: alphabet 5
        assert capi.host_anchor_dict_encode(kmers, k) == exp
        assert O.kmers_to_ints(O.decode_anchor_dict(exp, n, k), k) == ints


@pytest.mark.parametrize("ring,helpers,spares", [("8", "3", "3"), ("8", "1", "0"), ("64", "1", "5"), ("4", "2", "6")])
def test_host_anchor_dict_stream_under_other_feed_shapes(lib, monkeypatch, ring, helpers, spares):
    """the feed between the helpers and the chain -- ring size, helpers at work all the time, spares that join while the look-ahead
    is thin -- decides who makes which record when, never the bytes: 29 and 39 segments several times round small rings, one helper with
    five spares, == the oracle's stream"""
    import numpy as np
    import oracle_lib as O
    from leon_amd import capi
    monkeypatch.setenv("LEON_CHAIN_RING", ring)
    monkeypatch.setenv("LEON_CHAIN_HELPERS", helpers)
    monkeypatch.setenv("LEON_CHAIN_SPARES", spares)
    rng = np.random.default_rng(17)
    sizes = [2, 5, 5, 2, 3, 3, 3, 2] + [256] * 72
    for k, n in [(31, 30000), (63, 20000)]:
        ints = [(int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 1 << 62)) << 62)) & ((1 << (2 * k)) - 1) for _ in range(n)]
        w = O.kwords(k)
        kmers = np.array([[x & 0xFFFFFFFFFFFFFFFF, x >> 64][:w] for x in ints], dtype=np.uint64).reshape(-1)
        syms = np.array([(x >> (2 * (k - 1 - i))) & 3 for x in ints for i in range(k)], dtype=np.uint8)
        assert capi.host_anchor_dict_encode(kmers, k) == O.rc_encode_stream(np.ones(len(syms), dtype=np.uint8), syms, sizes)


@pytest.mark.parametrize("kind", ["polyA", "skewed", "two_letter", "uniform_long"])
def test_host_anchor_dict_chain_rare_paths(lib, kind):
    """the host chain's branch-free form covers 'no byte' / 'one byte' per symbol; skewed streams drive the other
    outcomes (several bytes at once, the range < BOTTOM reset, quotient fix-up) and long ones cross many reciprocal chunks"""
    import numpy as np
    import oracle_lib as O
    from leon_amd import capi
    rng = np.random.default_rng(11)
    k = 31
    n = {"polyA": 40000, "skewed": 40000, "two_letter": 40000, "uniform_long": 200000}[kind]
    if kind == "polyA":
        syms = np.zeros(n * k, dtype=np.uint8)
        syms[rng.integers(0, n * k, 50)] = 3
    elif kind == "skewed":
        syms = rng.choice(4, size=n * k, p=[0.97, 0.01, 0.01, 0.01]).astype(np.uint8)
    elif kind == "two_letter":
        syms = (rng.integers(0, 2, n * k) * 2).astype(np.uint8)
    else:
        syms = rng.integers(0, 4, n * k).astype(np.uint8)
    s2 = syms.reshape(n, k).astype(np.uint64)
    kmers = np.zeros(n, dtype=np.uint64)
    for i in range(k):
        kmers = (kmers << np.uint64(2)) | s2[:, i]
    sizes = [2, 5, 5, 2, 3, 3, 3, 2] + [256] * 72
    exp = O.rc_encode_stream(np.ones(len(syms), dtype=np.uint8), syms, sizes)
    assert capi.host_anchor_dict_encode(kmers, k) == exp


@pytest.mark.parametrize("k,n", [(31, 5000), (63, 1200), (9, 40), (31, 0)])
def test_host_anchor_dict_decode_inverts_encode(lib, k, n):
    """Leon::decodeAnchorDict on the host (the decoder's first step, no GPU): inverse of the dictionary stream, == oracle"""
    import numpy as np
    import oracle_lib as O
    from leon_amd import capi
    rng = np.random.default_rng(5)
    ints = [(int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 1 << 62)) << 62)) & ((1 << (2 * k)) - 1) for _ in range(n)]
    w = O.kwords(k)
    kmers = np.array([[x & 0xFFFFFFFFFFFFFFFF, x >> 64][:w] for x in ints], dtype=np.uint64).reshape(-1)
    stream = capi.host_anchor_dict_encode(kmers, k)
    got = capi.anchor_dict_decode(stream, n, k)
    assert np.array_equal(got, kmers)
    assert O.kmers_to_ints(O.decode_anchor_dict(stream, n, k), k) == ints


@pytest.mark.parametrize("kind", ["uniform", "poly_a", "skewed", "two_letter", "long"])
def test_host_anchor_dict_decode_on_skewed_streams(lib, kind):
    """the decoder's renormalisation is branch-free for "no byte" / "one byte" and a loop for the rest: streams that spend most
    of their time in the rest (poly-A: many symbols per byte, then several bytes at once) and a long one (past the reciprocal
    stream's first chunks) decode back to what was coded, and to what the oracle's decoder gives"""
    import numpy as np
    import oracle_lib as O
    from leon_amd import capi
    rng = np.random.default_rng(11)
    k, n = 31, 300000 if kind == "long" else 20000
    if kind == "poly_a":
        syms = np.zeros(n * k, dtype=np.uint8)
        syms[rng.integers(0, n * k, 200)] = rng.integers(1, 4, 200)
    elif kind == "skewed":
        syms = rng.choice(4, size=n * k, p=[0.97, 0.01, 0.01, 0.01]).astype(np.uint8)
    elif kind == "two_letter":
        syms = (rng.integers(0, 2, n * k) * 2).astype(np.uint8)
    else:
        syms = rng.integers(0, 4, n * k).astype(np.uint8)
    s2 = syms.reshape(n, k).astype(np.uint64)
    kmers = np.zeros(n, dtype=np.uint64)
    for i in range(k):
        kmers = (kmers << np.uint64(2)) | s2[:, i]
    stream = capi.host_anchor_dict_encode(kmers, k)
    assert np.array_equal(capi.anchor_dict_decode(stream, n, k), kmers)
    if kind != "long":
        assert np.array_equal(np.asarray(O.decode_anchor_dict(stream, n, k), dtype=np.uint64).reshape(-1)[:n], kmers)
    for cut in (len(stream) // 2, 3):                              # a truncated stream decodes to something or is refused: never a crash, never a hang
        try:
            capi.anchor_dict_decode(stream[:cut], n, k)
        except capi.LeonDnaError:
            pass

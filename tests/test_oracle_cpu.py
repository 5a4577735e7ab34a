"""CPU tests (no GPU): the oracle against its committed self-golden vectors, its own decoder, and a
pure-Python restatement of the primitives (independent second implementation for small cases)."""
import hashlib
import json
import os

import numpy as np
import pytest

import common
import oracle_lib as O

GOLD = json.load(open(os.path.join(common.GOLDEN, "self_golden.json")))
M64 = (1 << 64) - 1


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def test_toy_fixture_is_the_reference_file():
    # sha256 recorded in SURVEY.md section 2a for /root/reference/data/toy.fasta
    data = open(os.path.join(common.GOLDEN, "toy.fasta"), "rb").read()
    assert sha(data) == "73eef1cda4ac89ad381b862ac15c956491a23c138c7f21d89c4e82365af53199"
    reads = [l for l in data.decode().splitlines() if not l.startswith(">")]
    assert len(reads) == 200 and all(len(r) == 100 for r in reads)


def _inputs(name):
    if name.startswith("toy"):
        return common.toy_reads()
    if name.startswith("synthetic 3000"):
        return common.synthetic(3000, 150, 12000, seed=101, n_rate=0.002)
    return common.synthetic(2000, 100, 6000, seed=102, ragged=True)


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_oracle_matches_self_golden(case):
    bases, off = _inputs(case["name"])
    assert sha(bases) == case["input_sha256"]
    bl, solid, tai = common.make_bloom(bases, off, case["k"], case["min_abundance"])
    assert len(solid) == case["n_solid"] and tai == case["bloom_tai"]
    assert sha(bl.bits.tobytes()) == case["bloom_sha256"]
    res = O.encode(bases, off, case["k"], case["reads_per_block"], bl)
    assert res.n_anchors == case["n_anchors"] and res.n_symbols == case["n_symbols"]
    assert [len(b) for b in res.blocks] == case["block_sizes"]
    assert res.block_nreads == case["block_nreads"]
    assert [sha(b) for b in res.blocks] == case["block_sha256"]
    assert sha(res.anchor_dict) == case["anchor_dict_sha256"]
    assert sha(res.anchor_pos.tobytes()) == case["anchor_pos_sha256"]
    assert sha(res.events.tobytes()) == case["events_sha256"]
    # the reference's own acceptance criterion (scripts/simple_test.sh:62): decompress(compress(x)) == x
    anchors = O.decode_anchor_dict(res.anchor_dict, res.n_anchors, case["k"])
    assert np.array_equal(anchors, res.anchor_kmers)
    r = 0
    for payload, nr in zip(res.blocks, res.block_nreads):
        for j, d in enumerate(O.decode_block(case["k"], bl, anchors, payload, nr, len(bases) + 16)):
            assert d == bases[int(off[r + j]):int(off[r + j + 1])]
        r += nr
    assert r == case["n_reads"]


# ---- pure-Python second implementation of the primitives (small cases only) ----
def py_hash64(key, seed):
    h = seed
    h ^= ((h << 7) & M64) ^ ((key * (h >> 3)) & M64) ^ (~(((h << 11) + (key ^ (h >> 5))) & M64) & M64)
    h = ((~h & M64) + ((h << 21) & M64)) & M64
    h ^= h >> 24
    h = (h + ((h << 3) & M64) + ((h << 8) & M64)) & M64
    h ^= h >> 14
    h = (h + ((h << 2) & M64) + ((h << 4) & M64)) & M64
    h ^= h >> 28
    h = (h + ((h << 31) & M64)) & M64
    return h


def py_revcomp(x, k):
    r = 0
    for _ in range(k):
        r = (r << 2) | ((x & 3) ^ 2)
        x >>= 2
    return r


class PyModel:
    def __init__(self, n):
        self.n, self.r = n, list(range(n + 1))

    def update(self, c):
        for i in range(c + 1, self.n + 1):
            self.r[i] += 1


def py_rc_encode(models, syms, sizes):
    TOP, BOTTOM = 1 << 56, 1 << 48
    ms = [PyModel(s) for s in sizes]
    low, rng, out = 0, M64, bytearray()
    for m, c in zip(models, syms):
        mod = ms[m]
        rng //= mod.r[mod.n]
        low = (low + mod.r[c] * rng) & M64
        rng = (rng * (mod.r[c + 1] - mod.r[c])) & M64
        while True:
            if (low ^ ((low + rng) & M64)) < TOP:
                pass
            elif rng < BOTTOM:
                rng = (-low) & (BOTTOM - 1)
            else:
                break
            out.append(low >> 56)
            rng = (rng << 8) & M64
            low = (low << 8) & M64
        mod.update(c)
    for _ in range(8):
        out.append(low >> 56)
        low = (low << 8) & M64
    return bytes(out)


def test_known_answers_and_python_primitives():
    ka = GOLD["known_answers"]
    assert hex(O.lib.lo_hash_seed(0)) == ka["hash_seed0"]
    assert O.lib.lo_hash_seed(0) == (0xAAAAAAAA55555555 * 0xB5B5B5B54B4B4B4B) & M64
    for k_, s_, h_ in ka["hash64"]:
        assert hex(O.lib.lo_hash64(int(k_, 16), int(s_, 16))) == h_
        assert py_hash64(int(k_, 16), int(s_, 16)) == int(h_, 16)
    assert [hex(O.lib.lo_random_value(i)) for i in range(4)] == ka["random_values_head"]
    for x, r in ka["revcomp_k31"]:
        assert O.lib.lo_revcomp(int(x, 16), 31) == int(r, 16) == py_revcomp(int(x, 16), 31)
    # cano2 is min(v, revcomp of the (first,last) base pair)
    cano2 = [0, 1, 2, 3, 4, 5, 3, 7, 8, 9, 0, 4, 9, 13, 1, 5]
    for v in range(16):
        p, s = v >> 2, v & 3
        assert cano2[v] == min(v, ((s ^ 2) << 2) | (p ^ 2))
    sizes = [2, 5, 5, 2, 3, 3, 3, 2] + [256] * 72
    syms = ka["rc_stream"]["symbols"]
    enc = O.rc_encode_stream([m for m, _ in syms], [v for _, v in syms], sizes)
    assert enc.hex() == ka["rc_stream"]["payload_hex"]
    assert py_rc_encode([m for m, _ in syms], [v for _, v in syms], sizes) == enc
    kb = ka["bloom_5000_k31"]
    bl = O.Bloom(5000, 31)
    bl.insert(np.array([int(x, 16) for x in kb["kmers"]], dtype=np.uint64))
    assert len(bl.bits) == kb["nbytes"]
    assert [int(i) for i in np.flatnonzero(np.unpackbits(bl.bits, bitorder="little"))] == kb["set_bits"]
    assert [bl.contains4(int(x, 16), 1) for x in kb["kmers"]] == kb["contains4_right"]
    assert [bl.contains4(int(x, 16), 0) for x in kb["kmers"]] == kb["contains4_left"]


def test_range_coder_random_streams_roundtrip_and_python_agreement():
    rng = np.random.default_rng(5)
    sizes = [2, 5, 5, 2, 3, 3, 3, 2] + [256] * 72
    for n in [0, 1, 7, 300, 5000]:
        m = rng.integers(0, 80, size=n).astype(np.uint8)
        v = np.array([rng.integers(0, sizes[x]) for x in m], dtype=np.uint8)
        enc = O.rc_encode_stream(m, v, sizes)
        assert np.array_equal(O.rc_decode_stream(enc, m, sizes), v)
        if n <= 300:
            assert py_rc_encode(m.tolist(), v.tolist(), sizes) == enc


def test_bloom_is_strand_symmetric_and_contains4_consistent():
    k = 31
    rng = np.random.default_rng(9)
    kmers = rng.integers(0, 1 << 62, size=400, dtype=np.uint64)
    bl = O.Bloom(400 * 12, k)
    bl.insert(kmers)
    for x in kmers[:100]:
        x = int(x)
        assert bl.contains(x) and bl.contains(O.lib.lo_revcomp(x, k))        # revcomp-invariant positions
        mask = (1 << 62) - 1
        for right in (0, 1):
            res = bl.contains4(x, right)
            for nt in range(4):
                nb = ((x << 2) & mask) | nt if right else (x >> 2) | (nt << 60)
                assert bool(res >> nt & 1) == bl.contains(nb)


def test_edge_reads_roundtrip():
    k = 31
    bases, off = common.synthetic(500, 150, 4000, seed=8)
    bl, _, _ = common.make_bloom(bases, off, k)
    reads = [b"", b"A", b"ACGT" * 7 + b"AC", b"N" * 40, b"ACGTN" * 30, bases[:150], bases[150:300], b"T" * 31]
    b2, off2 = O.reads_to_arrays(reads)
    res = O.encode(b2, off2, k, 3, bl)
    assert res.block_nreads == [3, 3, 2]
    anchors = O.decode_anchor_dict(res.anchor_dict, res.n_anchors, k)
    out = []
    for payload, nr in zip(res.blocks, res.block_nreads):
        out += O.decode_block(k, bl, anchors, payload, nr, 10000)
    assert out == reads


@pytest.mark.parametrize("k", [32, 47, 63])
def test_two_word_kmers_roundtrip(k):
    """32 <= k <= 63: LargeInt<2> k-mers (config 5 of BASELINE.json uses k = 63)"""
    bases, off = common.synthetic(1500, 250, 12000, seed=70 + k, n_rate=0.001)
    bl, solid, tai = common.make_bloom(bases, off, k)
    assert len(solid) % 2 == 0 and O.kwords(k) == 2
    ints = O.kmers_to_ints(solid, k)
    for x in ints[:50]:
        assert x < (1 << (2 * k)) and bl.contains(x)
        rc = 0
        y = x
        for _ in range(k):
            rc = (rc << 2) | ((y & 3) ^ 2)
            y >>= 2
        assert bl.contains(rc) and x <= rc                          # canonical, strand-symmetric bloom
    res = O.encode(bases, off, k, 400, bl)
    assert (res.anchor_pos >= 0).sum() > 1000
    anchors = O.decode_anchor_dict(res.anchor_dict, res.n_anchors, k)
    assert np.array_equal(anchors, res.anchor_kmers)
    r = 0
    for payload, nr in zip(res.blocks, res.block_nreads):
        for j, d in enumerate(O.decode_block(k, bl, anchors, payload, nr, len(bases) + 16)):
            assert d == bases[int(off[r + j]):int(off[r + j + 1])]
        r += nr
    assert r == 1500


@pytest.mark.parametrize("k,nreads", [(31, 200), (15, 120), (33, 60)])
def test_oracle_equals_independent_python_restatement(k, nreads):
    """tests/py_leon.py restates the path from DESIGN.md's rules in pure Python; the C oracle must agree byte for byte
    (blocks, dictionary stream, anchors, bloom bits)"""
    import py_leon
    if k == 31:
        bases, off = common.toy_reads()
    else:
        bases, off = common.synthetic(nreads, 90, 1500, seed=300 + k, n_rate=0.004, err=0.02)
    reads = [bases[int(off[i]):int(off[i + 1])].decode() for i in range(nreads)]
    b2, off2 = O.reads_to_arrays(reads)
    bl, solid, tai = common.make_bloom(b2, off2, k)
    pb = py_leon.Bloom(tai, k)
    for x in O.kmers_to_ints(solid, k):
        pb.insert(x)
    assert bytes(pb.bits) == bl.bits.tobytes()
    rpb = 70
    ref = O.encode(b2, off2, k, rpb, bl, trace=False)
    blocks, dstream, anchors = py_leon.encode(reads, k, rpb, pb)
    assert anchors == O.kmers_to_ints(ref.anchor_kmers, k)
    assert blocks == ref.blocks
    assert dstream == ref.anchor_dict

"""-m "not gpu": the header and quality streams' CPU sides.  The oracle's header coder (oracle/leon_oracle.c) against the
independent Python implementation (tests/py_header.py), its own decoder, and the PRODUCT's host decoder
(leon_host_header_decode_blocks); the lossless quality codec (leon_host_qual_*) against Python's zlib."""
import zlib

import numpy as np
import pytest

import hdr_samples as H
import oracle_lib as O
import py_header as P


@pytest.fixture(scope="module")
def capi():
    import leon_amd
    from leon_amd import capi as c
    leon_amd.build_library()
    c.load_library()
    return c


@pytest.mark.parametrize("name,make,n", [("sra", H.sra, 1500), ("toy", H.toy_like, 800), ("nasty", H.nasty, 600)])
def test_header_oracle_equals_python_restatement_and_round_trips(name, make, n):
    hs = make(n)
    pay, trace = O.header_encode_block(hs, hs[0], with_trace=True)
    assert pay == P.encode_block(hs, hs[0]), "C oracle and the Python restatement disagree"
    assert O.header_decode_block(pay, len(hs), hs[0], sum(map(len, hs)) + 64) == hs
    assert len(trace) > 0 and trace[:, 0].max() <= 21
    # a different first header (the file's, not the block's) must be honoured
    pay2 = O.header_encode_block(hs[5:], hs[0])
    assert pay2 == P.encode_block(hs[5:], hs[0])
    assert O.header_decode_block(pay2, len(hs) - 5, hs[0], sum(map(len, hs)) + 64) == hs[5:]


def test_header_cost_on_regular_headers():
    """sanity of the model: SRA-style headers (two deltas and an end mark per header) cost a few bytes each"""
    hs = H.sra(20000)
    pay = O.header_encode_block(hs, hs[0])
    assert len(pay) < 0.2 * sum(map(len, hs))


@pytest.mark.parametrize("make,n,rpb", [(H.sra, 5300, 1000), (H.nasty, 700, 64), (H.toy_like, 10, 50000)])
def test_product_host_header_decoder_inverts_the_oracle_encoder(capi, make, n, rpb):
    hs = make(n)
    blocks = [(b // rpb, O.header_encode_block(hs[b:b + rpb], hs[0]), len(hs[b:b + rpb])) for b in range(0, n, rpb)]
    for threads in (1, 4):
        assert capi.host_header_decode_blocks(blocks, hs[0], n_threads=threads) == hs
    # a corrupted block is reported with its number, never crashed on
    bad = list(blocks)
    bad[-1] = (bad[-1][0], bytes(255 - x for x in bad[-1][1]), bad[-1][2])
    try:
        out = capi.host_header_decode_blocks(bad, hs[0])
        assert out != hs
    except capi.LeonDnaError as e:
        assert e.code == -1 and "does not decode" in str(e)
    assert capi.host_header_decode_blocks([], b"") == []


@pytest.mark.parametrize("L,n,rpb,level", [(100, 2500, 1000, -1), (0, 300, 50, 9), (150, 10, 50000, 1)])
def test_lossless_quality_blocks_are_zlib_of_the_joined_lines(capi, L, n, rpb, level):
    qs = H.fastq_quals(n, L)
    blob, off = O.reads_to_arrays(qs)
    blocks = capi.host_qual_encode_blocks(blob, off, rpb, zlib_level=level, n_threads=3)
    assert [b[0] for b in blocks] == list(range((n + rpb - 1) // rpb))
    for bid, pay, nr in blocks:
        text = b"".join(q + b"\n" for q in qs[bid * rpb:(bid + 1) * rpb])
        assert nr == len(qs[bid * rpb:(bid + 1) * rpb])
        assert zlib.decompress(pay) == text                       # any zlib reader inverts it
        assert pay == zlib.compress(text, level)                   # and it IS compress2 at that level
    nbytes = [sum(len(q) for q in qs[b * rpb:(b + 1) * rpb]) for b in range(len(blocks))]
    assert capi.host_qual_decode_blocks(blocks, nbytes, n_threads=2) == qs
    with pytest.raises(capi.LeonDnaError):
        capi.host_qual_decode_blocks(blocks, [x + 1 for x in nbytes])
    bad = [(blocks[0][0], blocks[0][1][:-3] + b"abc", blocks[0][2])] + blocks[1:]
    with pytest.raises(capi.LeonDnaError):
        capi.host_qual_decode_blocks(bad, nbytes)


def test_stream_oracles_match_the_committed_self_golden_vectors():
    """tests/golden/self_golden.json "streams": the header coder and the lossy quality rule, frozen (self-golden, not reference output)"""
    import hashlib
    import json
    import os
    import common
    sha = lambda b: hashlib.sha256(bytes(b)).hexdigest()
    gold = json.load(open(os.path.join(common.GOLDEN, "self_golden.json")))["streams"]
    toy = [l[1:].rstrip("\n").encode() for l in open(os.path.join(common.GOLDEN, "toy.fasta")) if l.startswith(">")]
    sets = {"toy.fasta headers rpb50000": toy, "toy.fasta headers rpb64": toy, "sra 3000 rpb1000": H.sra(3000, seed=1), "nasty 600 rpb100": H.nasty(600, seed=3)}
    for c in gold["header"]:
        heads, rpb = sets[c["name"]], c["reads_per_block"]
        assert sha(b"\n".join(heads)) == c["input_sha256"]
        blocks = [O.header_encode_block(heads[i:i + rpb], heads[0]) for i in range(0, len(heads), rpb)]
        assert [sha(b) for b in blocks] == c["block_sha256"] and blocks[0][:32].hex() == c["first_block_head_hex"], c["name"]
    bases, off = common.synthetic(800, 120, 5000, seed=103, n_rate=0.003, err=0.02)
    reads = [bases[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
    quals = [q[:len(r)].ljust(len(r), b"J") for q, r in zip(H.fastq_quals(len(reads), 130, seed=9), reads)]
    for c in gold["qual_smooth"]:
        bl, solid, tai = common.make_bloom(bases, off, c["k"])
        assert tai == c["bloom_tai"] and sha(bl.bits.tobytes()) == c["bloom_sha256"] and sha(b"".join(quals)) == c["quals_sha256"]
        assert sha(b"".join(O.qual_smooth(bl, c["k"], r, q) for r, q in zip(reads, quals))) == c["smoothed_sha256"], c["name"]

"""The C++ host mirror (leon_amd/host: Leon / DnaEncoder / main) above the C-ABI."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import common
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEON = os.path.join(ROOT, "leon_amd", "lib", "leon")


@pytest.fixture(scope="module")
def leon_bin():
    if not os.path.exists(LEON):
        import leon_amd
        leon_amd.build_library()
    return LEON


def test_cli_error_contract(leon_bin):
    # /root/reference/src/main.cpp:38-41: -v prints the banner and returns EXIT_FAILURE
    r = subprocess.run([leon_bin, "-v"], capture_output=True, text=True)
    assert r.returncode == 1 and "C-ABI version" in r.stdout
    # main.cpp:46-49: exceptions become "EXCEPTION: <msg>" on stderr and EXIT_FAILURE
    for args in (["-c"], ["-file", "x", "-c", "-d"], ["-file", "x", "-bogus"], ["-file", "/nonexistent/x.leon", "-d"]):
        r = subprocess.run([leon_bin] + args, capture_output=True, text=True)
        assert r.returncode == 1 and r.stderr.startswith("EXCEPTION: "), (args, r.stderr)


def _read_container(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"LEONDNA2"
    k, rpb, n_reads, n_blocks, n_anchors, dict_bytes, tai, bloom_bytes, n_hash, nbits = struct.unpack_from("<IIQQQQQQII", raw, 8)
    o = 8 + struct.calcsize("<IIQQQQQQII")
    table3 = struct.unpack_from("<%dQ" % (3 * n_blocks), raw, o)
    table = [x for b in range(n_blocks) for x in table3[3 * b:3 * b + 2]]
    o += 24 * n_blocks
    d = raw[o:o + dict_bytes]; o += dict_bytes
    bloom = raw[o:o + bloom_bytes]; o += bloom_bytes
    blocks = []
    for b in range(n_blocks):
        blocks.append(raw[o:o + table[2 * b]]); o += table[2 * b]
    assert o == len(raw)
    return dict(k=k, rpb=rpb, n_reads=n_reads, n_anchors=n_anchors, tai=tai, dict=d, bloom=bloom, blocks=blocks,
                nreads=[table[2 * b + 1] for b in range(n_blocks)])


@pytest.mark.gpu
def test_cli_compress_toy_matches_oracle(leon_bin, tmp_path):
    src = os.path.join(common.GOLDEN, "toy.fasta")
    dst = str(tmp_path / "toy.fasta")
    shutil.copy(src, dst)
    r = subprocess.run([leon_bin, "-file", dst, "-c", "-kmer-size", "31", "-abundance", "3", "-nb-cores", "4"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    c = _read_container(dst + ".leon")                       # INSTALL:21-23: data/toy.fasta -> data/toy.fasta.leon
    bases, off = common.toy_reads()
    bl, solid, tai = common.make_bloom(bases, off, 31, 3)
    assert c["tai"] == tai and np.array_equal(np.frombuffer(c["bloom"], dtype=np.uint8), bl.bits)
    ref = O.encode(bases, off, 31, 50000, bl, trace=False)
    assert c["blocks"] == ref.blocks and c["nreads"] == ref.block_nreads
    assert c["dict"] == ref.anchor_dict and c["n_anchors"] == ref.n_anchors
    # and the stream decodes back to the input with the oracle's decoder
    anchors = O.decode_anchor_dict(c["dict"], c["n_anchors"], 31)
    dec = O.decode_block(31, bl, anchors, c["blocks"][0], c["nreads"][0], len(bases) + 16)
    assert b"".join(dec) == bases


@pytest.mark.gpu
def test_cli_round_trip(leon_bin, tmp_path):
    """the reference's own acceptance test (scripts/simple_test.sh:51-62): compress, decompress, compare -- on the DNA stream"""
    src = os.path.join(common.GOLDEN, "toy.fasta")
    dst = str(tmp_path / "toy.fasta")
    shutil.copy(src, dst)
    r = subprocess.run([leon_bin, "-file", dst, "-c", "-kmer-size", "31", "-abundance", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([leon_bin, "-file", dst + ".leon", "-d"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = open(dst + ".d").read().split("\n")[:-1]             # X.fasta.leon -> X.fasta.d
    want = [l.strip() for l in open(src) if l.strip() and not l.startswith(">")]
    assert got == want
    # a FASTQ with N and ragged lengths, k = 21
    bases, off = common.synthetic(1200, 120, 5000, seed=5, n_rate=0.004, ragged=True, err=0.02)
    fq = str(tmp_path / "x.fastq")
    with open(fq, "w") as f:
        for i in range(len(off) - 1):
            s = bases[int(off[i]):int(off[i + 1])].decode()
            f.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
    r = subprocess.run([leon_bin, "-file", fq, "-c", "-kmer-size", "21", "-abundance", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([leon_bin, "-file", fq + ".leon", "-d"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = open(fq + ".d").read().split("\n")[:-1]
    assert got == [bases[int(off[i]):int(off[i + 1])].decode() for i in range(len(off) - 1)]

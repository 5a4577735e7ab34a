"""The C++ host mirror (leon_amd/host: Leon, the .leon HDF5 container, the FASTA/FASTQ reader, main) above the C-ABI.
CPU part: the error contract of /root/reference/src/main.cpp:38-49, the container layer through HDF5's own tools
(h5ls / h5dump / h5diff of /opt/conda/bin), the reader.  GPU part: the reference's acceptance test
(/root/reference/scripts/simple_test.sh:51-62: `-c -lossless`, `-d`, diff) and the flags of /root/reference/README.md:52-58."""
import gzip
import json
import os
import shutil
import subprocess
import zlib

import numpy as np
import pytest

import common
import hdr_samples as H
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEON = os.path.join(ROOT, "leon_amd", "lib", "leon")
H5BIN = "/opt/conda/bin"


@pytest.fixture(scope="module")
def leon_bin():
    import leon_amd
    leon_amd.build_library()
    return LEON


def run(*args, **kw):
    return subprocess.run(list(args), capture_output=True, text=True, **kw)


def h5_dataset(path, name, dtype=np.uint8):
    out = path + ".dump"
    r = run(os.path.join(H5BIN, "h5dump"), "-d", "/" + name, "-b", "LE", "-o", out, path)
    assert r.returncode == 0, r.stderr
    data = np.fromfile(out, dtype=dtype)
    os.remove(out)
    return data


def h5_names(path):
    r = run(os.path.join(H5BIN, "h5ls"), "-r", path)
    assert r.returncode == 0, r.stderr
    return {l.split()[0]: l for l in r.stdout.splitlines()}


# ---------------------------------------------------------------------------------------------------------------- CPU
def test_cli_error_contract(leon_bin, tmp_path):
    # /root/reference/src/main.cpp:38-41: -v prints the banner and returns EXIT_FAILURE
    r = run(leon_bin, "-v")
    assert r.returncode == 1 and "C-ABI version" in r.stdout
    # main.cpp:46-49: exceptions become "EXCEPTION: <msg>" on stderr and EXIT_FAILURE
    junk = str(tmp_path / "junk.leon")
    open(junk, "wb").write(b"not hdf5 at all" * 100)
    for args in (["-c"], ["-file", "x", "-c", "-d"], ["-file", "x", "-bogus"], ["-file", "/nonexistent/x.leon", "-d"],
                 ["-file", "x", "-c", "-kmer-size", "abc"], ["-file", "x", "-c", "-kmer-size"], ["-file", "x", "-c", "-abundance", "0"],
                 ["-file", "x", "-c", "-gpus", "0"], ["-file", "/nonexistent/reads.fa", "-c"], ["-file", junk, "-d"]):
        r = run(leon_bin, *args)
        assert r.returncode == 1 and r.stderr.startswith("EXCEPTION: "), (args, r.stderr)
    assert not os.path.exists("/nonexistent/reads.fa.leon")


def test_container_layer_through_hdf5_tools(leon_bin, tmp_path):
    a, b = str(tmp_path / "a.leon"), str(tmp_path / "b.leon")
    for p in (a, b):
        r = run(leon_bin, "-selftest-container", p)
        assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    names = h5_names(a)
    for n in ("/leon/dna/block_0", "/leon/dna/block_1", "/leon/header/block_0", "/leon/metadata/infobyte", "/leon/metadata/dna_blocksizes"):
        assert n in names, names
    assert "{100000}" in names["/leon/dna/block_0"] and "{0}" in names["/leon/dna/block_1"]
    blk = h5_dataset(a, "leon/dna/block_0")
    assert len(blk) == 100000 and blk[5] == ((5 * 2654435761 & 0xFFFFFFFF) >> 13) & 0xFF
    assert list(h5_dataset(a, "leon/metadata/dna_blocksizes", np.uint64)) == [1, 2, 3, 2 ** 64 - 1, 0]
    # parity of containers is dataset-level (h5diff); this writer also drops the timestamps, so files are equal
    assert run(os.path.join(H5BIN, "h5diff"), a, b).returncode == 0
    assert open(a, "rb").read() == open(b, "rb").read()


def test_bank_reader(leon_bin, tmp_path):
    reads = [("r1 first", "ACGTNACGT", "IIIIIIIII"), ("r2", "A", "#"), ("", "GGGTTT", "ABCDEF"), ("r4 x/1", "ACGT" * 40, "5" * 160)]
    fq = str(tmp_path / "x.fastq")
    open(fq, "w").write("".join("@%s\n%s\n+\n%s\n" % r for r in reads))
    fa = str(tmp_path / "x.fa")
    open(fa, "w").write("".join(">%s\n%s\n" % (h, "\n".join(s[i:i + 7] for i in range(0, len(s), 7))) for h, s, _ in reads))     # wrapped lines
    crlf = str(tmp_path / "crlf.fastq")
    open(crlf, "wb").write("".join("@%s\r\n%s\r\n+%s\r\n%s\r\n" % (h, s, h, q) for h, s, q in reads).encode())
    gz = str(tmp_path / "x.fastq.gz")
    with gzip.open(gz, "wb") as f:
        f.write(open(fq, "rb").read())
    out = {}
    for p in (fq, fa, crlf, gz):
        r = run(leon_bin, "-selftest-bank", p)
        assert r.returncode == 0, r.stderr
        out[p] = json.loads(r.stdout)
    # the '+' lines: bare in x.fastq (nothing to store), the header again in crlf.fastq (one byte: the default kind)
    plus = {p: out[p].pop("plus") for p in out}
    assert plus[fq] == {"bytes": 0, "default": 0, "exceptions": 0, "first_exception": -1, "text_bytes": 0} and plus[gz] == plus[fq]
    assert plus[crlf] == {"bytes": 1, "default": 1, "exceptions": 0, "first_exception": -1, "text_bytes": 0}
    nb = sum(len(s) for _, s, _ in reads)
    assert out[fq]["fastq"] and out[fq]["reads"] == 4 and out[fq]["bases"] == nb and out[fq]["qual_bytes"] == nb
    assert out[fq]["header_bytes"] == sum(len(h) for h, _, _ in reads)
    assert out[gz] == out[fq] and out[crlf] == out[fq]
    assert not out[fa]["fastq"] and out[fa]["qual_bytes"] == 0 and out[fa]["fasta_line_width"] == 7 and out[fq]["fasta_line_width"] == 0
    ragged = str(tmp_path / "ragged.fa")
    open(ragged, "w").write(">a\nACGTACG\nACG\n>b\nACGTA\nACGTACG\n")          # lines of differing widths: not reproducible, so 0
    r = run(leon_bin, "-selftest-bank", ragged)
    assert r.returncode == 0 and json.loads(r.stdout)["fasta_line_width"] == 0
    assert (out[fa]["fnv_bases"], out[fa]["fnv_headers"], out[fa]["bases"]) == (out[fq]["fnv_bases"], out[fq]["fnv_headers"], nb)
    # mixed '+' lines: bare, the header again, other text -- the exceptions are kept against the first record's kind
    mixed = str(tmp_path / "mixed.fastq")
    open(mixed, "w").write("".join("@%s\n%s\n+%s\n%s\n" % (h, s, ("", h, "", "something else")[i], q) for i, (h, s, q) in enumerate(reads)))
    r = run(leon_bin, "-selftest-bank", mixed)
    assert r.returncode == 0 and json.loads(r.stdout)["plus"] == {"bytes": 1 + 2 + 2 + 1 + 14, "default": 0, "exceptions": 2, "first_exception": 1, "text_bytes": 14}
    # a truncated .gz is an error, not a shorter file (zlib hands out what it could inflate, then 0 bytes + Z_BUF_ERROR)
    big = str(tmp_path / "big.fastq.gz")
    import random
    rnd = random.Random(3)
    with gzip.open(big, "wb") as f:
        for i in range(20000):
            sq = "".join(rnd.choice("ACGT") for _ in range(80))
            f.write(("@r%d\n%s\n+\n%s\n" % (i, sq, "I" * 80)).encode())
    assert json.loads(run(leon_bin, "-selftest-bank", big).stdout)["reads"] == 20000
    blob = open(big, "rb").read()
    open(big, "wb").write(blob[:len(blob) * 2 // 3])
    r = run(leon_bin, "-selftest-bank", big)
    assert r.returncode == 1 and r.stderr.startswith("EXCEPTION: ") and "truncated" in r.stderr, r.stderr
    for bad in ("@r\nACGT\n+\nIII\n", "@r\nACGT\n", "ACGT\n", "@r\nACGT\nIIII\nIIII\n"):
        p = str(tmp_path / "bad.fq")
        open(p, "w").write(bad)
        r = run(leon_bin, "-selftest-bank", p)
        assert r.returncode == 1 and r.stderr.startswith("EXCEPTION: "), (bad, r.stderr)


def test_automatic_abundance_threshold():
    """leon_kmer_auto_cutoff (host-only): first local minimum of the abundance spectrum, never below 2"""
    from leon_amd import capi
    capi.load_library()

    def cut(pairs):
        h = np.zeros(256, dtype=np.uint64)
        for a, v in pairs.items():
            h[a] = v
        return capi.kmer_auto_cutoff(h)
    # a 30x read set with 1 % errors (numbers of the shape DESIGN.md section 4.6 derives): valley at 4
    assert cut({1: 9_000_000_000, 2: 75_000_000, 3: 2_300_000, 4: 100_000, 5: 160_000, 6: 500_000, 17: 40_000_000, 30: 1_000_000}) == 4
    assert cut({1: 1000, 2: 10, 3: 50, 4: 100}) == 2                 # a valley at 2 is kept
    assert cut({1: 10, 2: 100, 3: 50}) == 2                          # never below 2
    assert cut({1: 1000, 2: 100, 3: 10, 4: 1}) == 2                  # no valley: coverage too low to separate errors
    assert cut({1: 6000, 2: 30, 3: 2, 60: 5, 70: 80, 80: 20}) == 4   # a gap of empty abundances: the valley starts where it ends falling
    assert cut({}) == 2


# ---------------------------------------------------------------------------------------------------------------- GPU
def _write_fastq(path, reads, heads, quals):
    with open(path, "wb") as f:
        for h, s, q in zip(heads, reads, quals):
            f.write(b"@" + h + b"\n" + s + b"\n+\n" + q + b"\n")


def _synthetic_fastq(tmp_path, n=2600, L=110, seed=5, **kw):
    bases, off = common.synthetic(n, L, 6000, seed=seed, **kw)
    reads = [bases[int(off[i]):int(off[i + 1])] for i in range(n)]
    heads = H.sra(n, seed=seed)
    quals = H.fastq_quals(n, 0, seed=seed)
    quals = [(q * (len(r) // max(len(q), 1) + 1))[:len(r)] if q else b"I" * len(r) for q, r in zip(quals, reads)]
    return reads, heads, quals


@pytest.mark.gpu
def test_cli_toy_fasta_streams_match_the_oracle(leon_bin, tmp_path):
    src = os.path.join(common.GOLDEN, "toy.fasta")
    dst = str(tmp_path / "toy.fasta")
    shutil.copy(src, dst)
    r = run(leon_bin, "-file", dst, "-c", "-kmer-size", "31", "-abundance", "3", "-nb-cores", "4")
    assert r.returncode == 0, r.stderr
    leon = dst + ".leon"                                           # /root/reference/INSTALL:21-23: data/toy.fasta -> data/toy.fasta.leon
    names = h5_names(leon)
    for n in ("/leon/dna/block_0", "/leon/header/block_0", "/leon/anchors/dict", "/bloom/bits", "/leon/metadata/params", "/leon/metadata/firstheader"):
        assert n in names, names
    assert "/leon/qual" not in names
    bases, off = common.toy_reads()
    bl, solid, tai = common.make_bloom(bases, off, 31, 3)
    ref = O.encode(bases, off, 31, 50000, bl, trace=False)
    assert h5_dataset(leon, "leon/dna/block_0").tobytes() == ref.blocks[0]
    assert h5_dataset(leon, "leon/anchors/dict").tobytes() == ref.anchor_dict
    assert np.array_equal(h5_dataset(leon, "bloom/bits"), bl.bits)
    heads = [l[1:].rstrip("\n").encode() for l in open(src) if l.startswith(">")]
    assert h5_dataset(leon, "leon/header/block_0").tobytes() == O.header_encode_block(heads, heads[0])
    assert h5_dataset(leon, "leon/metadata/firstheader").tobytes() == heads[0]
    params = h5_dataset(leon, "leon/metadata/params", np.uint64)
    assert list(params[:4]) == [1, 1, 0, 31] and params[5] == 200 and params[6] == ref.n_anchors and params[8] == tai
    assert list(h5_dataset(leon, "leon/metadata/dna_blocksizes", np.uint64)) == [len(ref.blocks[0]), 200, len(bases)]
    # two runs give the same container (h5diff)
    other = str(tmp_path / "again" / "toy.fasta")
    os.makedirs(os.path.dirname(other))
    shutil.copy(src, other)
    assert run(leon_bin, "-file", other, "-c", "-abundance", "3").returncode == 0
    assert run(os.path.join(H5BIN, "h5diff"), leon, other + ".leon").returncode == 0
    # INSTALL:21-23: -d restores the file; -test-file compares it with the original beside it
    r = run(leon_bin, "-file", leon, "-d", "-test-file")
    assert r.returncode == 0 and "identical" in r.stdout, r.stdout + r.stderr
    assert open(dst + ".d", "rb").read() == open(src, "rb").read()
    # a FASTA whose sequences are wrapped at a fixed width comes back wrapped
    wrapped = str(tmp_path / "wrapped.fa")
    with open(wrapped, "w") as f:
        for l in open(src):
            f.write(l if l.startswith(">") else "".join(l.strip()[i:i + 60] + "\n" for i in range(0, len(l.strip()), 60)))
    assert run(leon_bin, "-file", wrapped, "-c", "-abundance", "3").returncode == 0
    r = run(leon_bin, "-file", wrapped + ".leon", "-d", "-test-file")
    assert r.returncode == 0 and "identical" in r.stdout, r.stdout + r.stderr
    # automatic abundance (the default): still a lossless round trip of the file
    assert run(leon_bin, "-file", other, "-c").returncode == 0
    assert run(leon_bin, "-file", other + ".leon", "-d", "-test-file").returncode == 0


@pytest.mark.gpu
def test_reference_acceptance_test_lossless_fastq_gz(leon_bin, tmp_path):
    """/root/reference/scripts/simple_test.sh:51,54,62: leon -c -lossless -file X.fastq.gz; leon -d -file X.fastq.leon; diff X.fastq X.fastq.d"""
    reads, heads, quals = _synthetic_fastq(tmp_path, n=120000, L=100, seed=8, n_rate=0.002, err=0.02)       # three read blocks
    fq = str(tmp_path / "SRR.fastq")
    _write_fastq(fq, reads, heads, quals)
    with open(fq, "rb") as f, gzip.open(fq + ".gz", "wb", compresslevel=1) as g:
        shutil.copyfileobj(f, g)
    r = run(leon_bin, "-c", "-lossless", "-file", fq + ".gz", "-kmer-size", "25")
    assert r.returncode == 0, r.stderr
    assert os.path.exists(fq + ".leon") and not os.path.exists(fq + ".gz.leon")          # X.fastq.gz -> X.fastq.leon
    r = run(leon_bin, "-d", "-file", fq + ".leon")
    assert r.returncode == 0, r.stderr
    assert run("diff", fq, fq + ".d").returncode == 0
    # the same file encoded in batches of one block (the path a file above 120 M reads takes) gives the same container
    shutil.move(fq + ".leon", fq + ".whole")
    r = run(leon_bin, "-c", "-lossless", "-file", fq + ".gz", "-kmer-size", "25", env=dict(os.environ, LEON_BATCH_BLOCKS="1"))
    assert r.returncode == 0, r.stderr
    assert run(os.path.join(H5BIN, "h5diff"), fq + ".whole", fq + ".leon").returncode == 0
    # the lossy default on the same file: its qualities smoothed in ONE call over the whole file, or in calls of one read block each (what a
    # file beyond the DNA stream's batch size, or a device short of memory, takes -- ADVICE r3): the same container
    shutil.copy(fq + ".leon", fq + ".lossless")
    r = run(leon_bin, "-c", "-file", fq + ".gz", "-kmer-size", "25")
    assert r.returncode == 0, r.stderr
    shutil.move(fq + ".leon", fq + ".lossy_whole")
    r = run(leon_bin, "-c", "-file", fq + ".gz", "-kmer-size", "25", env=dict(os.environ, LEON_BATCH_BLOCKS="1"))
    assert r.returncode == 0, r.stderr
    assert run(os.path.join(H5BIN, "h5diff"), fq + ".lossy_whole", fq + ".leon").returncode == 0
    shutil.move(fq + ".lossless", fq + ".leon")
    # and decoded in rounds of one block (the path a host short of memory takes) it gives the same file
    r = run(leon_bin, "-d", "-file", fq + ".leon", env=dict(os.environ, LEON_DECODE_BLOCKS="1"))
    assert r.returncode == 0, r.stderr
    assert run("diff", fq, fq + ".d").returncode == 0
    # ... and with the DNA blocks of two rounds decoded by one device call (what files of 800 blocks or more do)
    r = run(leon_bin, "-d", "-file", fq + ".leon", env=dict(os.environ, LEON_DECODE_BLOCKS="1", LEON_DECODE_DNA_ROUNDS="2", LEON_HEADER_DEVICE_BLOCKS="0"))
    assert r.returncode == 0, r.stderr
    assert run("diff", fq, fq + ".d").returncode == 0
    # header blocks through the device (what rounds of hundreds of blocks do) and through the host threads alone: the same file
    for where in ("0", "1000000"):
        r = run(leon_bin, "-d", "-file", fq + ".leon", env=dict(os.environ, LEON_HEADER_DEVICE_BLOCKS=where))
        assert r.returncode == 0, r.stderr
        assert run("diff", fq, fq + ".d").returncode == 0
    r = run(leon_bin, "-c", "-noheader", "-noqual", "-file", fq + ".gz", "-kmer-size", "25")
    assert r.returncode == 0, r.stderr
    r = run(leon_bin, "-d", "-file", fq + ".leon", env=dict(os.environ, LEON_DECODE_BLOCKS="2"))
    assert r.returncode == 0, r.stderr
    got = open(fq + ".d", "rb").read().split(b"\n")
    assert got[0] == b">0" and got[2 * 119999] == b">119999" and len(got) == 2 * 120000 + 1      # the read index runs on across rounds
    shutil.copy(fq + ".whole", fq + ".leon")
    # the quality blocks are zlib over the block's quality lines
    q0 = h5_dataset(fq + ".leon", "leon/qual/block_0").tobytes()
    assert zlib.decompress(q0) == b"".join(q + b"\n" for q in quals[:50000])
    assert os.path.getsize(fq + ".leon") < 0.6 * os.path.getsize(fq)
    # -test-file finds the original as X.fastq or X.fastq.gz
    os.remove(fq)
    r = run(leon_bin, "-d", "-test-file", "-file", fq + ".leon")
    assert r.returncode == 0 and "identical" in r.stdout, r.stdout + r.stderr
    # '+' lines that repeat the header (older Illumina / SRA dumps; simple_test.sh:62 is a byte diff): every record, then a mix of
    # bare / repeated / other text, in rounds of one block so that the exceptions are found from the middle of the table
    for name, plus in (("all", lambda i, h: h), ("mixed", lambda i, h: (b"", h, b"", b"", h, b"free text %d" % i)[i % 6] if i % 1000 < 6 else h)):
        fp = str(tmp_path / ("plus_%s.fastq" % name))
        with open(fp, "wb") as f:
            for i, (h, sq, q) in enumerate(zip(heads[:60000], reads, quals)):               # two read blocks
                f.write(b"@" + h + b"\n" + sq + b"\n+" + plus(i, h) + b"\n" + q + b"\n")
        r = run(leon_bin, "-c", "-lossless", "-file", fp, "-kmer-size", "25")
        assert r.returncode == 0, r.stderr
        assert "/leon/metadata/pluslines" in h5_names(fp + ".leon")
        for env in (os.environ, dict(os.environ, LEON_DECODE_BLOCKS="1")):
            r = run(leon_bin, "-d", "-test-file", "-file", fp + ".leon", env=env)
            assert r.returncode == 0 and "identical" in r.stdout, r.stdout + r.stderr
            assert run("cmp", fp, fp + ".d").returncode == 0
    assert "/leon/metadata/pluslines" not in h5_names(fq + ".leon")          # bare '+' lines store nothing


@pytest.mark.gpu
def test_cli_stream_selection_flags_and_lossy_qualities(leon_bin, tmp_path):
    import leon_amd
    reads, heads, quals = _synthetic_fastq(tmp_path, n=3000, L=120, seed=11, n_rate=0.003, ragged=True, err=0.02)
    fq = str(tmp_path / "x.fastq")
    _write_fastq(fq, reads, heads, quals)
    k = 21

    def cycle(*flags):
        for f in (fq + ".leon", fq + ".d"):
            if os.path.exists(f):
                os.remove(f)
        r = run(leon_bin, "-file", fq, "-c", "-kmer-size", str(k), "-abundance", "2", *flags)
        assert r.returncode == 0, r.stderr
        r = run(leon_bin, "-file", fq + ".leon", "-d")
        assert r.returncode == 0, r.stderr
        return open(fq + ".d", "rb").read(), h5_names(fq + ".leon")
    norm = [bytes(c if c in b"ACGT" else ord("N") for c in r) for r in reads]
    # default = lossy qualities (README.md:55): DnaEncoder::smoothQuals against the file's bloom
    text, names = cycle()
    bases, off = O.reads_to_arrays(reads)
    bl, solid, tai = common.make_bloom(bases, off, k, 2)
    smooth = [O.qual_smooth(bl, k, r, q) for r, q in zip(reads, quals)]
    assert any(s != q for s, q in zip(smooth, quals))
    assert text == b"".join(b"@" + h + b"\n" + s + b"\n+\n" + q + b"\n" for h, s, q in zip(heads, norm, smooth))
    # -test-file on a lossy round trip reports the first byte that differs (both files plain: compared in slices by all cores)
    r = run(leon_bin, "-file", fq + ".leon", "-d", "-test-file")
    orig_text = open(fq, "rb").read()
    first = next(i for i in range(min(len(text), len(orig_text))) if text[i] != orig_text[i])
    assert r.returncode == 1 and r.stderr.startswith("EXCEPTION: ") and ("differs from %s at byte %d" % (fq, first)) in r.stderr, r.stderr
    # the lossy qualities wait for the bloom in HBM; a file too large for that goes through the file a second time
    # (LEON_QUAL_RESIDENT_MB=0 forces it): the same container either way
    shutil.move(fq + ".leon", fq + ".resident")
    r = run(leon_bin, "-file", fq, "-c", "-kmer-size", str(k), "-abundance", "2", env=dict(os.environ, LEON_QUAL_RESIDENT_MB="0"))
    assert r.returncode == 0, r.stderr
    assert run(os.path.join(H5BIN, "h5diff"), fq + ".resident", fq + ".leon").returncode == 0
    # -noheader: headers discarded, the read index stands in; -noqual: decompresses to FASTA (README.md:56-58)
    text, names = cycle("-noheader", "-lossless")
    assert "/leon/header" not in names and "/leon/qual/block_0" in names
    assert text == b"".join(b"@%d\n" % i + s + b"\n+\n" + q + b"\n" for i, (s, q) in enumerate(zip(norm, quals)))
    text, names = cycle("-noqual")
    assert "/leon/qual" not in names and "/leon/header/block_0" in names
    assert text == b"".join(b">" + h + b"\n" + s + b"\n" for h, s in zip(heads, norm))
    text, names = cycle("-seq-only")
    assert "/leon/qual" not in names and "/leon/header" not in names
    assert text == b"".join(b">%d\n" % i + s + b"\n" for i, s in enumerate(norm))
    # -gpus 2 (two contexts sharing the one device of the test box) writes the same container as -gpus 1
    assert run(leon_bin, "-file", fq, "-c", "-kmer-size", str(k), "-abundance", "2", "-lossless").returncode == 0
    shutil.move(fq + ".leon", fq + ".one")
    r = run(leon_bin, "-file", fq, "-c", "-kmer-size", str(k), "-abundance", "2", "-lossless", "-gpus", "2", env=dict(os.environ, LEON_SHARE_GPU="1"))
    assert r.returncode == 0, r.stderr
    assert run(os.path.join(H5BIN, "h5diff"), fq + ".one", fq + ".leon").returncode == 0
    r = run(leon_bin, "-file", fq, "-c", "-gpus", "9")
    assert r.returncode == 1 and r.stderr.startswith("EXCEPTION: ") and "device" in r.stderr


@pytest.mark.gpu
def test_cli_quality_encoder_choice(leon_bin, tmp_path):
    """`leon -c` writes zlib's own quality blocks (compress2 at the default level: the bytes upstream writes [RECALLED], and the one stream
    of the file a third-party library pins) unless the user asks otherwise: `-qual-deflate device` hands them to the device's deflate
    (runs + dynamic Huffman codes: inflates to the same text, other bytes), `-qual-deflate auto` lets a sample of the first lines decide;
    LEON_QUAL_DEFLATE is the same choice for tests when the flag is absent.  The container records who wrote the blocks
    (leon/metadata/params, word 14); whichever did, `leon -d` gives the file back"""
    import random
    import synth
    rnd = random.Random(3)
    g = synth.make_genome(30000, seed=21)
    bases, off = synth.make_reads(g, 2600, 120, seed=22, err=0.01)
    reads = [bytes(bases[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    noisy = [bytes(rnd.choice(b"#,-5:<>?@ABCDEFGHIJ") for _ in r) for r in reads]             # nothing to match: runs are enough
    stair = [bytes(74 - min(41, (j * 41) // len(r)) for j in range(len(r))) for r in reads]     # every line the same staircase: earlier lines match
    env0 = {k: v for k, v in os.environ.items() if k != "LEON_QUAL_DEFLATE"}
    P_REV, P_QUAL_ENCODER, ZLIB, DEVICE = 13, 14, 1, 2
    for name, quals, auto_picks_device in (("noisy", noisy, True), ("stair", stair, False)):
        fq = str(tmp_path / (name + ".fastq"))
        text = b"".join(b"@r%d\n" % i + r + b"\n+\n" + q + b"\n" for i, (r, q) in enumerate(zip(reads, quals)))
        qtext = b"".join(q + b"\n" for q in quals)
        open(fq, "wb").write(text)
        sizes = {}
        for how, flags, env in (("default", (), env0), ("host", ("-qual-deflate", "host"), env0), ("device", ("-qual-deflate", "device"), env0),
                                ("auto", ("-qual-deflate", "auto"), env0), ("env-device", (), dict(env0, LEON_QUAL_DEFLATE="device")),
                                ("flag-over-env", ("-qual-deflate", "host"), dict(env0, LEON_QUAL_DEFLATE="device"))):
            r = run(leon_bin, "-file", fq, "-c", "-lossless", "-kmer-size", "21", "-abundance", "2", *flags, env=env)
            assert r.returncode == 0, r.stderr
            line = next(l for l in r.stdout.splitlines() if l.startswith("quality stream"))
            on_device = how in ("device", "env-device") or (how == "auto" and auto_picks_device)
            assert ("deflated on the device" in line) == on_device, (name, how, line)
            sizes[how] = int(line.split("->")[1].split()[0])
            params = h5_dataset(fq + ".leon", "leon/metadata/params", np.uint64)
            assert len(params) == 15 and params[P_REV] == 2 and params[P_QUAL_ENCODER] == (DEVICE if on_device else ZLIB), (name, how, list(params))
            q0 = h5_dataset(fq + ".leon", "leon/qual/block_0").tobytes()
            assert zlib.decompress(q0) == qtext, (name, how)               # either way a zlib stream of the block's lines ...
            assert (q0 == zlib.compress(qtext)) == (not on_device), (name, how)     # ... zlib's own bytes unless the device was asked for
            r = run(leon_bin, "-file", fq + ".leon", "-d", env=env0)
            assert r.returncode == 0, r.stderr
            assert open(fq + ".d", "rb").read() == text, (name, how)
        assert sizes["default"] == sizes["host"] == sizes["flag-over-env"] and sizes["device"] == sizes["env-device"]
        assert sizes["auto"] <= 1.02 * min(sizes["device"], sizes["host"]) + 64, (name, sizes)       # the sample picked the smaller one
    # the lossy default takes the same road: smoothed lines through zlib unless asked otherwise
    r = run(leon_bin, "-file", fq, "-c", "-kmer-size", "21", "-abundance", "2", env=env0)
    assert r.returncode == 0 and "zlib on the host threads" in r.stdout, r.stdout + r.stderr
    assert h5_dataset(fq + ".leon", "leon/metadata/params", np.uint64)[P_QUAL_ENCODER] == ZLIB
    r = run(leon_bin, "-file", fq, "-c", "-kmer-size", "21", "-abundance", "2", "-qual-deflate", "device", env=env0)
    assert r.returncode == 0 and "deflated on the device" in r.stdout, r.stdout + r.stderr
    assert h5_dataset(fq + ".leon", "leon/metadata/params", np.uint64)[P_QUAL_ENCODER] == DEVICE
    r = run(leon_bin, "-file", fq, "-c", "-noqual", env=env0)
    assert r.returncode == 0 and h5_dataset(fq + ".leon", "leon/metadata/params", np.uint64)[P_QUAL_ENCODER] == 0
    r = run(leon_bin, "-file", fq, "-c", "-lossless", env=dict(env0, LEON_QUAL_DEFLATE="gpu"))
    assert r.returncode == 1 and r.stderr.startswith("EXCEPTION: ") and "LEON_QUAL_DEFLATE" in r.stderr
    r = run(leon_bin, "-file", fq, "-c", "-lossless", "-qual-deflate", "gpu", env=env0)
    assert r.returncode == 1 and r.stderr.startswith("EXCEPTION: ") and "-qual-deflate" in r.stderr
    r = run(leon_bin, "-file", fq, "-c", "-lossless", "-qual-deflate", env=env0)
    assert r.returncode == 1 and r.stderr.startswith("EXCEPTION: ")


@pytest.mark.gpu
def test_container_of_the_previous_layout_still_decodes(leon_bin, tmp_path):
    """the header block table went from 2 to 3 words per block without the container saying so.  Containers now carry their
    own revision (leon/metadata/params word 13; absent = revision 1) and revision 1 is read in both of its shapes.
    tests/golden/r2_layout_toy.fasta.leon was written by the round-2 build (git 12ee041: 13 parameter words, 2-word header table) from
    tests/golden/toy.fasta (tests/golden/make_r2_layout.sh); a file from a LATER revision is refused by name, not mis-parsed"""
    old = os.path.join(ROOT, "tests", "golden", "r2_layout_toy.fasta.leon")
    params = h5_dataset(old, "leon/metadata/params", np.uint64)
    n_blocks = 1
    assert len(params) == 13 and len(h5_dataset(old, "leon/metadata/header_blocksizes", np.uint64)) == 2 * n_blocks
    f = str(tmp_path / "toy.fasta.leon")
    shutil.copy(old, f)
    shutil.copy(os.path.join(ROOT, "tests", "golden", "toy.fasta"), str(tmp_path / "toy.fasta"))
    for env in (os.environ, dict(os.environ, LEON_HEADER_DEVICE_BLOCKS="0")):
        r = run(leon_bin, "-file", f, "-d", "-test-file", env=env)
        assert r.returncode == 0 and "identical" in r.stdout, r.stdout + r.stderr
    # today's writer on the same input: revision 2, 3-word table, the same reads back
    r = run(leon_bin, "-file", str(tmp_path / "toy.fasta"), "-c")
    assert r.returncode == 0, r.stderr
    params = h5_dataset(f, "leon/metadata/params", np.uint64)
    assert len(params) == 15 and params[13] == 2 and len(h5_dataset(f, "leon/metadata/header_blocksizes", np.uint64)) == 3 * n_blocks
    assert run(leon_bin, "-file", f, "-d", "-test-file").returncode == 0


@pytest.mark.gpu
def test_cli_failures_leave_no_container(leon_bin, tmp_path):
    """a failed compression ends with EXCEPTION: and a non-zero status, and leaves no .leon behind (a swallowed
    encode error used to produce a container with missing blocks and exit 0)"""
    fa = str(tmp_path / "long.fa")
    open(fa, "w").write(">r\n" + "ACGT" * 10 + "\n")
    r = run(leon_bin, "-file", fa, "-c", "-kmer-size", "64")
    assert r.returncode == 1 and r.stderr.startswith("EXCEPTION: ") and not os.path.exists(fa + ".leon")
    ro = tmp_path / "ro"
    ro.mkdir()
    shutil.copy(fa, str(ro / "x.fa"))
    os.chmod(str(ro), 0o555)
    try:
        r = run(leon_bin, "-file", str(ro / "x.fa"), "-c")
        if os.geteuid() != 0:                                      # root writes anywhere
            assert r.returncode == 1 and r.stderr.startswith("EXCEPTION: ")
    finally:
        os.chmod(str(ro), 0o755)
    # a container whose tables disagree with its metadata is refused, not trusted
    assert run(leon_bin, "-file", fa, "-c", "-kmer-size", "11").returncode == 0
    raw = bytearray(open(fa + ".leon", "rb").read())
    r = run(leon_bin, "-file", fa + ".leon", "-d")
    assert r.returncode == 0, r.stderr
    assert open(fa + ".d").read() == open(fa).read()
    empty = str(tmp_path / "empty.fa")
    open(empty, "w").write("")
    assert run(leon_bin, "-file", empty, "-c").returncode == 0
    assert run(leon_bin, "-file", empty + ".leon", "-d", "-test-file").returncode == 0
    assert open(empty + ".d").read() == ""


@pytest.mark.gpu
def test_cli_corrupted_containers_are_reported_not_crashed_on(leon_bin, tmp_path):
    """`leon -d` on damaged .leon files (random byte flips and truncations anywhere in the HDF5 file: metadata, tables, payloads,
    the bloom): every run must end by itself with status 0 (damage that happens not to matter, or that only changes the
    output) or 1 with `EXCEPTION:` -- never by a signal, never by a hang"""
    import random
    reads, heads, quals = _synthetic_fastq(tmp_path, n=400, L=80, seed=21, n_rate=0.01, err=0.03)
    fq = str(tmp_path / "f.fastq")
    _write_fastq(fq, reads, heads, quals)
    assert run(leon_bin, "-file", fq, "-c", "-kmer-size", "15", "-abundance", "2", "-lossless").returncode == 0
    good = open(fq + ".leon", "rb").read()
    rnd = random.Random(5)
    outcomes = {0: 0, 1: 0}
    for trial in range(40):
        raw = bytearray(good)
        if trial % 8 == 7:
            raw = raw[:rnd.randrange(len(raw))]
        else:
            for _ in range(rnd.choice([1, 1, 4, 64])):
                raw[rnd.randrange(len(raw))] = rnd.randrange(256)
        bad = str(tmp_path / ("bad%d.fastq.leon" % trial))
        open(bad, "wb").write(raw)
        r = subprocess.run([leon_bin, "-file", bad, "-d"], capture_output=True, text=True, timeout=120)
        assert r.returncode in (0, 1), (trial, r.returncode, r.stderr[-300:])
        if r.returncode == 1:
            assert r.stderr.startswith("EXCEPTION: "), (trial, r.stderr[-300:])
        outcomes[r.returncode] += 1
        for f in (bad, bad[:-5] + ".d"):
            if os.path.exists(f):
                os.remove(f)
    assert outcomes[1] > 0

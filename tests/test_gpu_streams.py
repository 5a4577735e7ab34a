"""-m gpu: the header stream through the C-ABI (leon_header_encode_batch: records on the device, one lane per header, then
k_rc_encode with the header model set) against the oracle's HeaderEncoder, block payloads byte for byte; and back
through the product's host decoder."""
import numpy as np
import pytest

import hdr_samples as H
import oracle_lib as O

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("rc_chains")]


def _ctx(rpb, **kw):
    import leon_amd
    return leon_amd.DnaEncodeContext(kmer_size=31, reads_per_block=rpb, bloom_tai=100000, **kw)


def _oracle_blocks(hs, rpb, first):
    return [O.header_encode_block(hs[b:b + rpb], first) for b in range(0, len(hs), rpb)]


@pytest.mark.parametrize("name,make,n,rpb", [("sra", H.sra, 12000, 5000), ("toy", H.toy_like, 3000, 700), ("nasty", H.nasty, 2000, 150),
                                             ("one_block", H.sra, 900, 50000), ("single", H.toy_like, 1, 10)])
def test_header_blocks_bit_exact(name, make, n, rpb):
    from leon_amd import capi
    hs = make(n)
    ctx = _ctx(rpb)
    blocks = ctx.header_encode_batch(hs)
    ref = _oracle_blocks(hs, rpb, hs[0])
    assert [b[0] for b in blocks] == list(range(len(ref)))
    assert [b[2] for b in blocks] == [len(hs[i:i + rpb]) for i in range(0, n, rpb)]
    for i, (b, r) in enumerate(zip(blocks, ref)):
        assert b[1] == r, "header block %d differs from the oracle" % i
    assert capi.host_header_decode_blocks(blocks, hs[0]) == hs
    # the decoder with the arithmetic decoding on the device (symbols there, text on the host threads): the same headers;
    # "nasty" has headers with hundreds of symbols, more than a block's share of the device buffer: the host decodes instead
    assert ctx.header_decode_blocks(blocks, hs[0]) == hs
    assert ctx.header_decode_blocks(blocks, hs[0], n_threads=3) == hs
    assert ctx.header_decode_blocks([], b"") == []
    # the same in its two halves, as `leon -d` uses them: the symbols of ALL blocks in one device call, then the text of any run of blocks
    S = ctx.header_symbol_set(blocks)
    try:
        for b0, nb in ((0, len(blocks)), (len(blocks) - 1, 1), (1, max(len(blocks) - 2, 0)), (0, 0)):
            if b0 > len(blocks):
                continue
            got = S.text(b0, nb, hs[0])
            assert got == hs[b0 * rpb:(b0 + nb) * rpb], (name, b0, nb)
    except capi.LeonDnaError as e:                                # "nasty": the set says its symbols did not fit -- the caller decodes the payloads on the host
        assert name == "nasty" and e.code == -4, e
    with pytest.raises(capi.LeonDnaError):
        S.text(len(blocks), 1, hs[0])                             # beyond the set
    S.close()
    if len(blocks) > 1:                                            # a payload that is not a header stream is reported, by both decoders
        bad = list(blocks)
        bad[1] = (bad[1][0], bytes(255 - x for x in bad[1][1]), bad[1][2])
        for dec in (lambda: ctx.header_decode_blocks(bad, hs[0]), lambda: capi.host_header_decode_blocks(bad, hs[0])):
            try:
                assert dec()[rpb:2 * rpb] != hs[rpb:2 * rpb]
            except capi.LeonDnaError as e:
                assert "does not decode" in str(e)
    ctx.close()


def test_header_stream_batches_shards_and_state():
    import leon_amd
    hs = H.sra(7000) + H.nasty(1300, seed=9)
    rpb, first = 500, b"some other first header 12"
    ref = _oracle_blocks(hs, rpb, first)
    # several batches of whole blocks, the last one partial
    ctx = _ctx(rpb)
    got = []
    for lo, hi in ((0, 2000), (2000, 2500), (2500, len(hs))):
        got += ctx.header_encode_batch(hs[lo:hi], first_header=first)
    assert [b[1] for b in got] == ref and [b[0] for b in got] == list(range(len(ref)))
    with pytest.raises(leon_amd.LeonDnaError) as e:                  # a partial block must be the last batch
        ctx.header_encode_batch(hs[:10], first_header=first)
    assert e.value.code == -4
    ctx.reset_stream()                                               # a new file on the same context
    assert [b[1] for b in ctx.header_encode_batch(hs, first_header=first)] == ref
    ctx.close()
    # ranks of a sharded job produce disjoint block ranges whose union is the stream
    union = []
    for r in range(3):
        c = _ctx(rpb)
        c.set_shard(r, 3)
        union += c.header_encode_batch(hs, first_header=first)
        c.close()
    assert sorted(b[0] for b in union) == list(range(len(ref)))
    assert [b[1] for b in sorted(union)] == ref
    # degenerate batches
    ctx = _ctx(rpb)
    assert ctx.header_encode_batch([]) == []
    assert [b[1] for b in ctx.header_encode_batch([b"", b"", b""])] == _oracle_blocks([b"", b"", b""], rpb, b"")
    ctx.close()


@pytest.mark.parametrize("k,kw", [(31, dict(n_rate=0.003, err=0.02)), (21, dict(ragged=True, n_rate=0.001)), (47, dict(err=0.03)), (63, dict())])
def test_lossy_quality_smoothing_equals_the_oracle(k, kw, monkeypatch):
    """leon_qual_smooth_batch (DnaEncoder::smoothQuals on the device) == the CPU restatement, every byte, whichever way the device goes through the reads"""
    import common
    bases, off = common.synthetic(1500, 150 if k < 60 else 220, 9000, seed=31 + k, **kw)
    n = len(off) - 1
    reads = [bases[int(off[i]):int(off[i + 1])] for i in range(n)] + [b"ACGT", b"", b"N" * 200, b"ACGTTGCA" * 40]
    bases, off = O.reads_to_arrays(reads)
    bl, solid, tai = common.make_bloom(bases, off, k)
    quals = [q[:len(r)].ljust(len(r), b"J") for q, r in zip(H.fastq_quals(len(reads), 260, seed=k), reads)]
    import leon_amd
    ctx = leon_amd.DnaEncodeContext(kmer_size=k, reads_per_block=500, bloom_tai=tai)
    ctx.bloom_upload(bl.bits)
    got = ctx.qual_smooth_batch(bases, off, b"".join(quals))        # the probes shared between the reads of a locus (minimizer order)
    want = b"".join(O.qual_smooth(bl, k, r, q) for r, q in zip(reads, quals))
    assert len(got) == len(want)
    assert got == want
    assert want != b"".join(quals)                                  # something was smoothed
    monkeypatch.setenv("LEON_QUAL_ORDER", "0")                      # every read probing for itself, in file order: the same bytes
    assert ctx.qual_smooth_batch(bases, off, b"".join(quals)) == want
    monkeypatch.delenv("LEON_QUAL_ORDER")
    ctx.reset_stream()                                              # after an encode of the same reads the device aligns them on their ANCHORS instead
    ctx.encode_batch(bases, off)
    ctx.finish()
    assert ctx.qual_smooth_batch(bases, off, b"".join(quals)) == want
    ctx.reset_stream()
    few = 100                                                       # a batch too small for the sort: file order again
    assert ctx.qual_smooth_batch(bases, off[:few + 1], b"".join(quals[:few])) == b"".join(O.qual_smooth(bl, k, r, q) for r, q in zip(reads[:few], quals[:few]))
    assert ctx.qual_smooth_batch(b"", np.zeros(1, dtype=np.uint64), b"") == b""
    ctx.close()


def test_automatic_abundance_on_the_device():
    """min_abundance = 0: the counter keeps the k-mers at or above the threshold leon_kmer_auto_cutoff derives from the spectrum"""
    import common
    from leon_amd import capi
    bases, off = common.synthetic(30000, 120, 100000, seed=77, err=0.01)      # 36x
    for k, maxkeys in ((31, 0), (47, 400000)):
        auto = capi.kmer_solid(bases, off, k, 0, max_keys_per_pass=maxkeys)
        hist = capi.kmer_solid(bases, off, k, 1, with_histogram=True, max_keys_per_pass=maxkeys)[1]
        cutoff = capi.kmer_auto_cutoff(hist)
        assert 2 <= cutoff <= 6, cutoff
        want = O.count_solid(bases, off, k, cutoff)
        w = capi.kmer_words(k)
        as_set = lambda a: set(map(tuple, np.asarray(a, dtype=np.uint64).reshape(-1, w)))
        assert as_set(auto) == as_set(want) and len(auto) == len(want)


def test_hip_streams_match_the_committed_golden_fixtures():
    """the device header path and the device smoothing against tests/golden/self_golden.json, without the oracle in the loop"""
    import hashlib
    import json
    import os
    import common
    import leon_amd
    sha = lambda b: hashlib.sha256(bytes(b)).hexdigest()
    gold = json.load(open(os.path.join(common.GOLDEN, "self_golden.json")))["streams"]
    toy = [l[1:].rstrip("\n").encode() for l in open(os.path.join(common.GOLDEN, "toy.fasta")) if l.startswith(">")]
    sets = {"toy.fasta headers rpb50000": toy, "toy.fasta headers rpb64": toy, "sra 3000 rpb1000": H.sra(3000, seed=1), "nasty 600 rpb100": H.nasty(600, seed=3)}
    for c in gold["header"]:
        ctx = _ctx(c["reads_per_block"])
        blocks = ctx.header_encode_batch(sets[c["name"]])
        ctx.close()
        assert [sha(b[1]) for b in blocks] == c["block_sha256"], c["name"]
    bases, off = common.synthetic(800, 120, 5000, seed=103, n_rate=0.003, err=0.02)
    reads = [bases[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
    quals = b"".join(q[:len(r)].ljust(len(r), b"J") for q, r in zip(H.fastq_quals(len(reads), 130, seed=9), reads))
    for c in gold["qual_smooth"]:
        bl, solid, tai = common.make_bloom(bases, off, c["k"])
        ctx = leon_amd.DnaEncodeContext(kmer_size=c["k"], reads_per_block=500, bloom_tai=tai)
        ctx.bloom_insert(solid)
        assert sha(ctx.bloom_download().tobytes()) == c["bloom_sha256"]
        assert sha(ctx.qual_smooth_batch(bases, off, quals)) == c["smoothed_sha256"], c["name"]
        ctx.close()


@pytest.mark.parametrize("seed", range(3))
def test_header_stream_fuzz(seed):
    """seeded random header sets -- mutated copies of the previous header (digit fields bumped, fields inserted / dropped /
    replaced by arbitrary bytes, leading zeros, very long runs), random block sizes: device blocks == oracle, and decode back"""
    import random
    from leon_amd import capi
    rnd = random.Random(1000 + seed)
    alphabet = [b"A", b"z", b"7", b"0", b" ", b":", b"/", b"_", b"\t", b"\x00", b"\xff", b"=", b"."]

    def mutate(h):
        fields = [h[i:i + rnd.randint(1, 9)] for i in range(0, len(h), 7)] or [b""]
        for _ in range(rnd.randint(0, 3)):
            op, i = rnd.randrange(6), rnd.randrange(len(fields))
            if op == 0:
                fields[i] = b"%d " % rnd.randint(0, 10 ** rnd.randint(1, 20))
            elif op == 1:
                fields[i] = b"0" * rnd.randint(1, 5) + b"%d:" % rnd.randint(0, 999)
            elif op == 2:
                fields.insert(i, b"".join(rnd.choice(alphabet) for _ in range(rnd.randint(0, 12))))
            elif op == 3 and len(fields) > 1:
                del fields[i]
            elif op == 4:
                fields[i] = bytes(rnd.randrange(256) for _ in range(rnd.randint(0, 30)))
            else:
                fields[i] = b"x" * rnd.randint(200, 600)
        return b"".join(fields)[:2000]
    hs = [b"SRR1.1 first 0001 length=100"]
    for _ in range(1500):
        hs.append(mutate(hs[-1]) if rnd.random() < 0.9 else hs[rnd.randrange(len(hs))])
    rpb = rnd.choice([1, 7, 100, 512])
    first = hs[rnd.randrange(len(hs))] if seed else hs[0]
    ctx = _ctx(rpb)
    blocks = ctx.header_encode_batch(hs, first_header=first)
    assert [b[1] for b in blocks] == _oracle_blocks(hs, rpb, first)
    assert capi.host_header_decode_blocks(blocks, first, n_threads=3) == hs
    assert ctx.header_decode_blocks(blocks, first, n_threads=3) == hs
    ctx.close()


def _quality_sets():
    import random
    rnd = random.Random(11)
    sets = {}
    sets["uniform19"] = H.fastq_quals(6000, 150, seed=5)                                   # near-incompressible: 19 symbols, no runs
    binned = []                                                                            # four levels with long runs (modern instruments)
    for _ in range(5000):
        q, cur = bytearray(), 70
        for _ in range(150):
            if rnd.random() < 0.08:
                cur = rnd.choice(b"#,:F")
            q.append(cur)
        binned.append(bytes(q))
    sets["binned_runs"] = binned
    sets["constant"] = [b"I" * 151] * 3000                                                 # runs far longer than 258, across chunks and threads
    sets["ragged"] = [bytes(rnd.choice(b"ABCDEFGH!~") for _ in range(rnd.choice((0, 1, 2, 3, 40, 257, 258, 259, 260, 700)))) for _ in range(2500)]
    sets["all_bytes"] = [bytes((i * 7 + j) % 256 if (i * 7 + j) % 256 != 10 else 11 for j in range(300)) for i in range(400)]   # 255 literal symbols in use
    sets["empty_lines"] = [b""] * 1000
    sets["one_read"] = [b"#"]
    sets["skewed"] = [b"F" * 140 + bytes(rnd.choice(b"#,:") for _ in range(10)) for _ in range(4000)]
    return sets


@pytest.mark.parametrize("name", ["uniform19", "binned_runs", "constant", "ragged", "all_bytes", "empty_lines", "one_read", "skewed"])
def test_device_deflate_of_quality_blocks(name):
    """leon_qual_deflate_blocks_device: every block is a zlib stream that inflates to the block's quality lines (any inflate reads
    it: Python's here, the product's own decoder below), about as small as zlib's own RLE strategy makes it, and the Adler-32
    the host combines from the chunks' is the text's"""
    import zlib
    from leon_amd import capi
    qs = _quality_sets()[name]
    rpb = 1000 if len(qs) > 1 else 50
    blob, off = O.reads_to_arrays(qs)
    d = capi.device_upload_bytes(blob)
    try:
        blocks = capi.qual_deflate_blocks_device(d, off, rpb, first_block_id=7)
    finally:
        capi.device_free(d)
    assert [b[0] for b in blocks] == [7 + i for i in range((len(qs) + rpb - 1) // rpb)]
    tot_dev = tot_rle = tot_def = 0
    for i, (bid, pay, nr) in enumerate(blocks):
        text = b"".join(q + b"\n" for q in qs[i * rpb:(i + 1) * rpb])
        assert nr == len(qs[i * rpb:(i + 1) * rpb])
        assert zlib.decompress(pay) == text, (name, i)            # (checks the Adler-32 too)
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_RLE)
        tot_rle += len(c.compress(text) + c.flush())
        tot_def += len(zlib.compress(text, 6))
        tot_dev += len(pay)
    # 32 KB deflate blocks against zlib's ~100+ KB ones: a few per cent of headers, never more than 5 % + a few bytes per chunk
    assert tot_dev <= 1.05 * tot_rle + 64 * len(blocks) + 12 * (len(blob) // 32768 + len(blocks)), (name, tot_dev, tot_rle, tot_def)
    nbytes = [sum(len(q) for q in qs[b * rpb:(b + 1) * rpb]) for b in range(len(blocks))]
    assert capi.host_qual_decode_blocks([(b[0] - 7, b[1], b[2]) for b in blocks], nbytes, n_threads=2) == qs
    print("%s: device %d, zlib RLE %d, zlib default %d bytes" % (name, tot_dev, tot_rle, tot_def))


def test_device_deflate_fuzz():
    """random shapes through leon_qual_deflate_blocks_device: alphabets of 1 to 200 symbols, run lengths from none to thousands, lines
    from empty to longer than a deflate block, block texts that end exactly on, one before and one after a 32 KB boundary"""
    import random
    import zlib
    from leon_amd import capi
    rnd = random.Random(2026)
    for case in range(40):
        nsym = rnd.choice((1, 2, 3, 5, 16, 41, 94, 200))
        alphabet = bytes(rnd.sample([c for c in range(1, 256) if c != 10], nsym))
        p_run = rnd.choice((0.0, 0.3, 0.9, 0.99, 0.999))
        rpb = rnd.choice((1, 7, 100, 1000))
        n = rnd.choice((1, 3, 50, 800))
        qs = []
        for _ in range(n):
            L = rnd.choice((0, 1, 2, 150, 151, 5000)) if case % 5 else rnd.choice((32767 - 1, 32768 - 1, 32769 - 1, 70000))
            q, cur = bytearray(), alphabet[0]
            for _ in range(L):
                if rnd.random() >= p_run:
                    cur = rnd.choice(alphabet)
                q.append(cur)
            qs.append(bytes(q))
            if sum(len(x) for x in qs) > 3_000_000:
                break
        blob, off = O.reads_to_arrays(qs)
        d = capi.device_upload_bytes(blob)
        try:
            blocks = capi.qual_deflate_blocks_device(d, off, rpb)
        finally:
            capi.device_free(d)
        assert len(blocks) == (len(qs) + rpb - 1) // rpb
        for i, (bid, pay, nr) in enumerate(blocks):
            assert zlib.decompress(pay) == b"".join(q + b"\n" for q in qs[i * rpb:(i + 1) * rpb]), (case, i)


def test_device_deflate_code_length_limit():
    """Fibonacci frequencies in one 32 KB deflate block would give Huffman codes of 19 bits: the encoder must stay within deflate's 15
    (it halves the frequencies and builds again), and within 7 for the code-length alphabet"""
    import heapq
    import zlib
    from leon_amd import capi
    fib = [1, 1]
    while sum(fib) + fib[-1] + fib[-2] < 32000:
        fib.append(fib[-1] + fib[-2])
    syms = [c for c in range(33, 33 + len(fib))]
    heap = [(-f, s) for f, s in zip(fib, syms)]
    heapq.heapify(heap)
    out, prev = bytearray(), None
    while heap:                                                   # never two equal neighbours: no runs, every byte a literal
        f, s = heapq.heappop(heap)
        if s == prev and heap:
            f2, s2 = heapq.heappop(heap)
            out.append(s2); prev = s2
            if f2 + 1 < 0:
                heapq.heappush(heap, (f2 + 1, s2))
            heapq.heappush(heap, (f, s))
            continue
        out.append(s); prev = s
        if f + 1 < 0:
            heapq.heappush(heap, (f + 1, s))
    line = bytes(out)
    assert len(line) < 32767 and len(set(line)) == len(fib) >= 20
    for qs in ([line], [line] * 3, [line[:20000], line[20000:]]):
        blob, off = O.reads_to_arrays(qs)
        d = capi.device_upload_bytes(blob)
        try:
            blocks = capi.qual_deflate_blocks_device(d, off, 50)
        finally:
            capi.device_free(d)
        assert zlib.decompress(blocks[0][1]) == b"".join(q + b"\n" for q in qs)

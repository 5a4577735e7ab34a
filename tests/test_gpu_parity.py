"""-m gpu: the HIP path, called through the C-ABI, against the CPU oracle on the same seeded inputs.
Bit-exact everywhere (integer / byte work)."""
import ctypes as C
import os

import numpy as np
import pytest

import common
import oracle_lib as O

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("rc_chains")]


def _ctx(k, rpb, tai, **kw):
    import leon_amd
    return leon_amd.DnaEncodeContext(kmer_size=k, reads_per_block=rpb, bloom_tai=tai, keep_trace=True, **kw)


def _kmers_of(bases, off, k, limit=20000):
    """python-int k-mers of the reads (first base in the highest bits)"""
    code = {65: 0, 67: 1, 84: 2, 71: 3}
    out = []
    mask = (1 << (2 * k)) - 1
    for r in range(len(off) - 1):
        s, e = int(off[r]), int(off[r + 1])
        km, valid = 0, 0
        for i in range(s, e):
            c = code.get(bases[i])
            if c is None:
                km, valid = 0, 0
                continue
            km = ((km << 2) | c) & mask
            valid += 1
            if valid >= k:
                out.append(km)
        if len(out) >= limit:
            break
    return out[:limit]


def _words(ints, k):
    """python ints -> flat uint64 array, kwords(k) words per k-mer (low word first)"""
    w = O.kwords(k)
    a = np.zeros((len(ints), w), dtype=np.uint64)
    for i, x in enumerate(ints):
        a[i, 0] = x & 0xFFFFFFFFFFFFFFFF
        if w == 2:
            a[i, 1] = x >> 64
    return a.reshape(-1)


@pytest.mark.parametrize("k", [31, 21, 9, 32, 47, 63])
def test_bloom_build_and_probe(k):
    bases, off = common.synthetic(3000, 120, 30000, seed=5)
    bl, solid, tai = common.make_bloom(bases, off, k)
    ctx = _ctx(k, 1000, tai)
    assert ctx.bloom_nbytes == len(bl.bits)
    ctx.bloom_insert(solid)
    assert np.array_equal(ctx.bloom_download(), bl.bits)          # insert kernel == oracle insert, bit for bit
    q = _kmers_of(bases, off, k, limit=6000)
    rng = np.random.default_rng(0)
    q += [int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 1 << 62)) << 62) & ((1 << (2 * k)) - 1) for _ in range(3000)]
    q = [x & ((1 << (2 * k)) - 1) for x in q]
    exp_c = np.array([bl.contains(x) for x in q], dtype=np.uint8)
    exp_l = np.array([bl.contains4(x, 0) for x in q], dtype=np.uint8)
    exp_r = np.array([bl.contains4(x, 1) for x in q], dtype=np.uint8)
    qw = _words(q, k)
    assert np.array_equal(ctx.bloom_contains(qw), exp_c)
    assert np.array_equal(ctx.bloom_contains4(qw, 0), exp_l)
    assert np.array_equal(ctx.bloom_contains4(qw, 1), exp_r)
    ctx.close()


def _random_symbol_streams(rng, n_streams, max_len):
    sizes = [2, 5, 5, 2, 3, 3, 3, 2] + [256] * 72
    syms, begin = [], [0]
    for s in range(n_streams):
        n = int(rng.integers(0, max_len))
        # a realistic mix: mostly small models and the low numeric models, some rare high ones
        m = rng.choice(80, size=n, p=_model_probs())
        v = np.array([rng.integers(0, sizes[x]) if rng.random() < 0.3 else min(sizes[x] - 1, int(rng.geometric(0.3)) - 1)
                      for x in m], dtype=np.uint8)
        syms.append(np.stack([m.astype(np.uint8), v], axis=1).reshape(-1))
        begin.append(begin[-1] + n)
    return (np.concatenate(syms) if syms else np.zeros(0, np.uint8)), np.array(begin, dtype=np.uint64), sizes


def _model_probs():
    p = np.full(80, 0.002)
    p[:8] = 0.05
    for g in range(8):
        p[8 + 9 * g] = 0.03
        p[8 + 9 * g + 1] = 0.03
    return p / p.sum()


def test_range_coder_streams():
    rng = np.random.default_rng(7)
    syms, begin, sizes = _random_symbol_streams(rng, 12, 6000)
    ctx = _ctx(31, 1000, 1000)
    got = ctx.rc_encode_streams(syms, begin)
    for i in range(len(begin) - 1):
        a, b = int(begin[i]), int(begin[i + 1])
        exp = O.rc_encode_stream(syms[2 * a:2 * b:2], syms[2 * a + 1:2 * b:2], sizes)
        assert got[i] == exp, "stream %d differs" % i
    ctx.close()


def test_host_chains_equal_the_device_coder_and_the_oracle(monkeypatch):
    """the same random symbol streams through the device's coder waves (k_rc_encode) and through the host chains fed by the device's
    modelers (k_rc_records -> host_blocks.h), whose tiles go through in chunks with the models' state parked in global memory in
    between: one chunk, three, more chunks than some streams have tiles; empty streams; streams of one symbol"""
    rng = np.random.default_rng(19)
    syms, begin, sizes = _random_symbol_streams(rng, 14, 7000)
    extra_m = rng.choice(80, size=20000, p=_model_probs()).astype(np.uint8)         # one long stream on top: many tiles per chunk
    extra_v = np.array([rng.integers(0, sizes[x]) for x in extra_m], dtype=np.uint8)
    syms = np.concatenate([syms, np.stack([extra_m, extra_v], axis=1).reshape(-1), np.array([0, 1], dtype=np.uint8)])
    begin = np.concatenate([begin, [begin[-1] + 20000, begin[-1] + 20000, begin[-1] + 20001]]).astype(np.uint64)   # + an empty stream, + a one-symbol stream
    want = [O.rc_encode_stream(syms[2 * int(a):2 * int(b):2], syms[2 * int(a) + 1:2 * int(b):2], sizes) for a, b in zip(begin[:-1], begin[1:])]
    ctx = _ctx(31, 1000, 1000)
    assert ctx.rc_encode_streams(syms, begin) == want
    monkeypatch.setenv("LEON_RC_STREAMS_ON_HOST", "1")
    for chunks in ("1", "3", "64"):
        monkeypatch.setenv("LEON_RC_HOST_CHUNKS", chunks)
        for threads in ("1", "5"):
            monkeypatch.setenv("LEON_RC_HOST_THREADS", threads)
            assert ctx.rc_encode_streams(syms, begin) == want, (chunks, threads)
    ctx.close()


def test_range_coder_exact_division_path(monkeypatch):
    """totals of 2^30 and more (a block with a billion symbols on one model: long reads) leave the coder's multiply-high +
    32-bit fix-up for an exact division, tile by tile.  The switch-over total is lowered here (test hook) so that ordinary
    streams and a whole encode take that path after their first few hundred symbols: same bytes as the oracle."""
    monkeypatch.setenv("LEON_RC_FAST_TOTAL_LOG2", "8")
    rng = np.random.default_rng(11)
    syms, begin, sizes = _random_symbol_streams(rng, 9, 9000)
    ctx = _ctx(31, 1000, 1000)
    got = ctx.rc_encode_streams(syms, begin)
    for i in range(len(begin) - 1):
        a, b = int(begin[i]), int(begin[i + 1])
        assert got[i] == O.rc_encode_stream(syms[2 * a:2 * b:2], syms[2 * a + 1:2 * b:2], sizes), "stream %d differs" % i
    ctx.close()
    bases, off = common.synthetic(3000, 150, 9000, seed=29, n_rate=0.001)
    _full_compare(bases, off, 31, 700)


def _full_compare(bases, off, k, rpb, window=0, batches=1, bloom=None):
    bl, solid, tai = bloom if bloom is not None else common.make_bloom(bases, off, k)
    ref = O.encode(bases, off, k, rpb, bl)
    ctx = _ctx(k, rpb, tai, resolve_window=window)
    ctx.bloom_upload(bl.bits)
    n = len(off) - 1
    blocks = []
    # split into `batches` calls on block boundaries
    nb = (n + rpb - 1) // rpb
    per = max(1, nb // batches) * rpb
    r0 = 0
    pos_all, addr_all, flags_all, ev_all = [], [], [], []
    while r0 < n:
        r1 = min(n, r0 + per)
        blocks += ctx.encode_batch(bases, off[r0:r1 + 1])
        p, a, f = ctx.trace_anchors(r1 - r0)
        pos_all.append(p); addr_all.append(a); flags_all.append(f)
        ev_all.append(ctx.trace_events(int(off[r1] - off[r0])))
        r0 = r1
    dict_payload, n_anchors = ctx.finish()
    pos, addr, flags, ev = (np.concatenate(x) for x in (pos_all, addr_all, flags_all, ev_all))
    # stage-wise: anchors, then events, then bytes
    assert np.array_equal(pos, ref.anchor_pos), "anchor positions differ"
    anchored = ref.anchor_pos >= 0
    assert np.array_equal(addr[anchored], ref.anchor_addr[anchored]), "anchor addresses differ"
    assert np.array_equal(flags[anchored], ref.flags[anchored]), "revcomp/inserted flags differ"
    assert n_anchors == ref.n_anchors
    assert np.array_equal(ctx.anchor_kmers(n_anchors), ref.anchor_kmers)
    assert np.array_equal(ev, ref.events), "walk events differ"
    assert [b[0] for b in blocks] == list(range(len(ref.blocks)))
    assert [b[2] for b in blocks] == ref.block_nreads
    for i, (b, r) in enumerate(zip(blocks, ref.blocks)):
        assert b[1] == r, "block %d payload differs" % i
    assert dict_payload == ref.anchor_dict
    st = ctx.stats()
    ctx.close()
    return ref, st


def test_toy_fasta_bit_exact():
    bases, off = common.toy_reads()
    _full_compare(bases, off, 31, 50)


@pytest.mark.parametrize("group", ["1", "2", "4", "8"])
@pytest.mark.parametrize("counts_apart", ["1", "0"])
def test_block_coder_every_group_size_and_layout(monkeypatch, group, counts_apart):
    """k_rc_encode codes 1, 2, 4 or 8 read blocks per workgroup (the launcher picks by the number of blocks; 8 only beyond 1 024 blocks, which no small
    test reaches) and keeps, for the read blocks' symbols, the byte-count models apart from the 256-symbol slots (13 slots per block at 8 per workgroup,
    so reads with N, errors, ragged lengths and no anchor here push some numeric models to the global overflow area): every combination against the
    oracle's bytes."""
    monkeypatch.setenv("LEON_RC_HOST_BLOCKS", "0")
    monkeypatch.setenv("LEON_RC_GROUP", group)
    monkeypatch.setenv("LEON_RC_CMP", counts_apart)
    bases, off = common.synthetic(2500, 150, 12000, seed=23, junk_reads=40, ragged=True, n_rate=0.002, err=0.02)
    _full_compare(bases, off, 31, 100)


def test_toy_fasta_default_block_size():
    bases, off = common.toy_reads()
    _full_compare(bases, off, 31, 50000)


@pytest.mark.parametrize("kw", [dict(), dict(n_rate=0.003), dict(ragged=True, n_rate=0.001), dict(err=0.05)])
def test_synthetic_bit_exact(kw):
    bases, off = common.synthetic(6000, 150, 20000, seed=11, **kw)
    _full_compare(bases, off, 31, 1000)


def test_small_windows_and_batches():
    # tiny resolution windows force many fixpoint rounds and cross-window dictionary reuse
    bases, off = common.synthetic(5000, 100, 8000, seed=3)
    _full_compare(bases, off, 31, 500, window=64)
    _full_compare(bases, off, 31, 500, window=1000, batches=3)


@pytest.mark.parametrize("k0", [5, 17, 29, 41, 53])
def test_final_key_filter_under_every_kmer_size(k0):
    # The filter of the final keys is addressed by a minimizer whose geometry (m-mer size, window, offset) depends on k
    # (kernels.h minimizer_geometry): a k-mer it failed to find again in a later window would anchor a read elsewhere than the
    # oracle does.  Every k of 5 .. 63 with windows of 48 reads, on reads that share their k-mers many times over -- and
    # on low-complexity reads, whose m-mers repeat inside one k-mer.
    for k in range(k0, min(k0 + 12, 64)):
        bases, off = common.synthetic(600, max(90, 2 * k + 20), 1500, seed=100 + k)
        _full_compare(bases, off, k, 200, window=48)
    k = min(k0 + 11, 63)
    rnd = np.random.default_rng(k)
    units = [b"A", b"AC", b"ACG", b"AACCGGTT", b"ACGTTGCATG"]
    reads = []
    for i in range(300):
        u = units[int(rnd.integers(len(units)))]
        r = bytearray((u * (200 // len(u) + 1))[:140 + int(rnd.integers(20))])
        for j in rnd.integers(0, len(r), size=2):
            r[int(j)] = b"ACGT"[int(rnd.integers(4))]
        reads.append(bytes(r))
    b2, off2 = O.reads_to_arrays(reads)
    _full_compare(b2, off2, k, 100, window=32)


def test_other_kmer_sizes_and_lengths():
    bases, off = common.synthetic(3000, 250, 15000, seed=9, n_rate=0.001)
    _full_compare(bases, off, 21, 700)
    bases, off = common.synthetic(2000, 40, 3000, seed=10)
    _full_compare(bases, off, 15, 300)


def test_edge_cases():
    k = 31
    bases, off = common.synthetic(2000, 150, 10000, seed=21)
    bl = common.make_bloom(bases, off, k)
    # reads shorter than k, exactly k, all-N, garbage reads with no solid k-mer, empty read
    extra = [b"ACGT", b"A" * 31, b"N" * 50, b"ACGTTGCA" * 20, b"", b"ACGTNNNNACGT" * 10]
    arr = np.frombuffer(bases, dtype=np.uint8)
    reads = [arr[int(off[i]):int(off[i + 1])].tobytes() for i in range(300)]
    mixed = []
    for i, r in enumerate(reads):
        mixed.append(r)
        if i % 50 == 0:
            mixed.append(extra[(i // 50) % len(extra)])
    b2, off2 = O.reads_to_arrays(mixed)
    _full_compare(b2, off2, k, 100, bloom=bl)
    # a single read, and an empty bloom (nothing anchors)
    b3, off3 = O.reads_to_arrays([reads[0]])
    _full_compare(b3, off3, k, 100, bloom=bl)
    empty = O.Bloom(1000, k)
    _full_compare(b2, off2, k, 100, bloom=(empty, None, 1000))


def test_roundtrip_through_oracle_decoder():
    k, rpb = 31, 1000
    bases, off = common.synthetic(4000, 150, 20000, seed=31, n_rate=0.002)
    bl, solid, tai = common.make_bloom(bases, off, k)
    ctx = _ctx(k, rpb, tai)
    ctx.bloom_insert(solid)
    blocks = ctx.encode_batch(bases, off)
    dict_payload, n_anchors = ctx.finish()
    anchors = O.decode_anchor_dict(dict_payload, n_anchors, k)
    r = 0
    for _, payload, nr in blocks:
        dec = O.decode_block(k, bl, anchors, payload, nr, 10 ** 7)
        for j, d in enumerate(dec):
            assert d == bases[int(off[r + j]):int(off[r + j + 1])]
        r += nr
    assert r == len(off) - 1
    ctx.close()


def test_hip_path_matches_committed_golden_fixtures():
    """the HIP path against tests/golden/self_golden.json (sha256 of every block), without the oracle encoder"""
    import hashlib
    import json
    import os
    gold = json.load(open(os.path.join(common.GOLDEN, "self_golden.json")))
    for case in gold["cases"]:
        if case["name"].startswith("toy"):
            bases, off = common.toy_reads()
        elif case["name"].startswith("synthetic 3000"):
            bases, off = common.synthetic(3000, 150, 12000, seed=101, n_rate=0.002)
        else:
            bases, off = common.synthetic(2000, 100, 6000, seed=102, ragged=True)
        solid = O.count_solid(bases, off, case["k"], case["min_abundance"])
        ctx = _ctx(case["k"], case["reads_per_block"], case["bloom_tai"])
        ctx.bloom_insert(solid)
        assert hashlib.sha256(ctx.bloom_download().tobytes()).hexdigest() == case["bloom_sha256"]
        blocks = ctx.encode_batch(bases, off)
        d, na = ctx.finish()
        assert [hashlib.sha256(b[1]).hexdigest() for b in blocks] == case["block_sha256"]
        assert hashlib.sha256(d).hexdigest() == case["anchor_dict_sha256"] and na == case["n_anchors"]
        assert ctx.stats()["n_symbols"] == case["n_symbols"]
        ctx.close()


def test_long_reads_two_byte_numerics():
    # read length > 255: sizes, anchor positions and error-position deltas need two-byte numerics,
    # the k-mer lanes need several passes per read
    bases, off = common.synthetic(1500, 700, 30000, seed=41, err=0.004, n_rate=0.0005)
    _full_compare(bases, off, 31, 400)
    bases, off = common.synthetic(300, 3000, 40000, seed=42, err=0.002)
    _full_compare(bases, off, 27, 100, window=128)


def test_high_error_and_low_complexity():
    # many unanchorable reads and long bifurcation lists; homopolymer / repeat genome stresses the dictionary
    bases, off = common.synthetic(3000, 150, 20000, seed=43, err=0.15)
    _full_compare(bases, off, 31, 1000)
    rep = (b"ACGTACGTTTGACCA" * 40)[:500]
    reads = [rep[i % 300:i % 300 + 120] for i in range(800)] + [b"A" * 100] * 50 + [b"AC" * 60] * 50
    b2, off2 = O.reads_to_arrays(reads)
    _full_compare(b2, off2, 21, 250, window=100)


def test_sharded_contexts_reproduce_the_single_stream():
    """leon_dna_set_shard: N contexts fed the same batches resolve the same file-order dictionary and split the
    blocks; their union must be byte-identical to the one-context stream (= the oracle's)."""
    k, rpb = 31, 300
    bases, off = common.synthetic(4000, 150, 15000, seed=51, n_rate=0.001)     # 14 blocks, the last partial
    bl, solid, tai = common.make_bloom(bases, off, k)
    ref = O.encode(bases, off, k, rpb, bl, trace=False)
    for world in (2, 3, 4, 5, 8):          # 8 = BASELINE configurations #4 / #5; the first batch's 6 blocks leave two of 8 ranks empty-handed
        got, dicts = [], []
        for rank in range(world):
            ctx = _ctx(k, rpb, tai, resolve_window=1000)
            ctx.set_shard(rank, world)
            ctx.bloom_upload(bl.bits)
            # two batches (whole blocks in the first one) to exercise global block ids across calls
            cut = 6 * rpb
            blocks = ctx.encode_batch(bases, off[:cut + 1]) + ctx.encode_batch(bases, off[cut:])
            d, na = ctx.finish()
            assert na == ref.n_anchors
            dicts.append(d)
            got += blocks
            ctx.close()
        got.sort()
        assert [g[0] for g in got] == list(range(len(ref.blocks)))
        assert [g[1] for g in got] == ref.blocks and [g[2] for g in got] == ref.block_nreads
        assert dicts[0] == ref.anchor_dict and all(len(d) == 0 for d in dicts[1:])


def test_walk_divided_by_anchor_reproduces_the_single_stream():
    """leon_dna_set_exchange: the walk of an N-rank job divided by ANCHOR (rank r walks the r-th slice of the batch's reads sorted
    by anchor address, whatever block they belong to; what it finds travels to the rank that codes the read's block) instead of by
    block range.  LEON_XCH_EMULATE: one context plays every rank's slice and keeps the words meant for its own rank -- the bytes of a
    real run.  The union of the ranks' blocks must be the one-context stream (= the oracle's), for worlds that leave ranks without a
    block, with N / errors / ragged reads / reads without an anchor, two batches, one- and two-word k-mers."""
    from leon_amd import capi
    for k in (31, 41):
        rpb = 300
        bases, off = common.synthetic(4000, 150, 15000, seed=51 + k, n_rate=0.002, ragged=True, junk_reads=40)
        bl, solid, tai = common.make_bloom(bases, off, k)
        ref = O.encode(bases, off, k, rpb, bl, trace=False)
        for world in (2, 3, 8):
            got = []
            for rank in range(world):
                ctx = _ctx(k, rpb, tai, resolve_window=1000)
                ctx.set_shard(rank, world)
                ctx.set_exchange(capi.XCH_EMULATE)
                ctx.bloom_upload(bl.bits)
                cut = 6 * rpb
                b1 = ctx.encode_batch(bases, off[:cut + 1])
                st1 = ctx.stats()
                b2 = ctx.encode_batch(bases, off[cut:])
                d, na = ctx.finish()
                assert na == ref.n_anchors and (d == ref.anchor_dict if rank == 0 else len(d) == 0)
                # a rank walks about a world-th of the batch's anchored reads, and receives exactly what its blocks hold
                if st1["n_blocks"]:                       # (a rank without a block of the batch has nothing to receive: it walks nothing here)
                    assert st1["walk_launches"] == 1 and st1["walk_reads"] <= cut and st1["xch_words_received"] > 0
                got += b1 + b2
                ctx.close()
            got.sort()
            assert [g[0] for g in got] == list(range(len(ref.blocks)))
            assert [g[1] for g in got] == ref.blocks and [g[2] for g in got] == ref.block_nreads, (k, world)


def test_walk_divided_by_anchor_with_a_real_exchange_between_contexts():
    """the same with LEON_XCH_BY_ANCHOR and a real exchange: `world` contexts on the one device, one thread each, their callbacks
    meeting at a barrier and handing one another the words (device to device, leon_device_copy) -- what an all-to-all does between
    GPUs.  Every rank calls the exchange for every batch, also a rank that codes no block of it (world 8: the first batch's 6 blocks)."""
    import threading
    from leon_amd import capi
    k, rpb = 31, 300
    bases, off = common.synthetic(4000, 150, 15000, seed=77, n_rate=0.001, junk_reads=25)
    bl, solid, tai = common.make_bloom(bases, off, k)
    ref = O.encode(bases, off, k, rpb, bl, trace=False)
    for world, share_lookups in ((2, False), (2, True), (8, True)):
        barrier = threading.Barrier(world)
        posted = [None] * world                      # rank -> (d_send, counts)
        posted_g = [None] * world                    # rank -> its gather buffer
        gathers = [0] * world
        recv_bufs = [None] * world                   # rank -> device pointer it owns (freed at the next call)
        import leon_amd
        lib = leon_amd.load_library()
        results, errors = [None] * world, []

        def exchange(rank):
            def fn(d_send, counts):
                posted[rank] = (d_send, counts)
                barrier.wait()
                total = sum(posted[src][1][rank] for src in range(world))
                if recv_bufs[rank]:
                    lib.leon_device_free(C.c_void_p(recv_bufs[rank]))
                    recv_bufs[rank] = None
                ptr = C.c_void_p()
                assert lib.leon_device_alloc(0, max(total, 1) * 8, C.byref(ptr)) == 0
                recv_bufs[rank] = ptr.value
                at = 0
                for src in range(world):
                    d, cnt = posted[src]
                    skip = sum(cnt[:rank])
                    if cnt[rank]:
                        assert lib.leon_device_copy(0, C.c_void_p(ptr.value + 8 * at), C.c_void_p(d + 8 * skip), cnt[rank] * 8) == 0
                    at += cnt[rank]
                barrier.wait()                       # nobody's send buffer is reused before everybody has copied from it
                return ptr.value, total
            return fn

        def gather(rank):
            # leon_dna_set_gather: the window's look-ups divided among the ranks, every rank's part copied into everybody's buffer
            def fn(d_buf, part_bytes, w):
                assert w == world
                posted_g[rank] = (d_buf, part_bytes)
                barrier.wait()
                assert all(pg[1] == part_bytes for pg in posted_g)
                for src in range(world):
                    if src != rank and part_bytes:
                        assert lib.leon_device_copy(0, C.c_void_p(d_buf + src * part_bytes), C.c_void_p(posted_g[src][0] + src * part_bytes), part_bytes) == 0
                gathers[rank] += 1
                barrier.wait()
            return fn

        def run(rank):
            try:
                ctx = _ctx(k, rpb, tai, resolve_window=1000)
                ctx.set_shard(rank, world)
                ctx.set_exchange(capi.XCH_BY_ANCHOR, exchange(rank))
                if share_lookups:
                    ctx.set_gather(gather(rank))
                ctx.bloom_upload(bl.bits)
                cut = 6 * rpb
                blocks = ctx.encode_batch(bases, off[:cut + 1]) + ctx.encode_batch(bases, off[cut:])
                d, na = ctx.finish()
                st = ctx.stats()
                ctx.close()
                results[rank] = (blocks, d, na, st)
            except Exception as e:                   # noqa: BLE001
                errors.append((rank, e))
                barrier.abort()
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        for p_ in recv_bufs:
            if p_:
                lib.leon_device_free(C.c_void_p(p_))
        got = sorted(b for r in results for b in r[0])
        assert [g[0] for g in got] == list(range(len(ref.blocks)))
        assert [g[1] for g in got] == ref.blocks and [g[2] for g in got] == ref.block_nreads, world
        assert results[0][1] == ref.anchor_dict and all(len(r[1]) == 0 for r in results[1:])
        assert sum(r[3]["xch_words_sent"] for r in results) == sum(r[3]["xch_words_received"] for r in results)
        lookups_on = share_lookups and os.environ.get("LEON_XCH_LOOKUPS") != "0"        # (the measurement override keeps every look-up on every rank)
        assert all(g == (results[0][3]["resolve_windows"] + 2 if lookups_on else 0) for g in gathers), gathers   # one per window (first batch: 2 windows)


def test_dictionary_stream_on_device_equals_host_thread():
    """the file-wide anchor-dictionary stream: default = host worker thread; LEON_F_DICT_ON_DEVICE = the device range
    coder on one workgroup.  Same bytes (and == oracle)."""
    k, rpb = 31, 1000
    bases, off = common.synthetic(5000, 150, 30000, seed=61)
    bl, solid, tai = common.make_bloom(bases, off, k)
    ref = O.encode(bases, off, k, rpb, bl, trace=False)
    outs = []
    for on_dev in (False, True):
        ctx = _ctx(k, rpb, tai, dict_on_device=on_dev)
        ctx.bloom_upload(bl.bits)
        blocks = ctx.encode_batch(bases, off)
        d, na = ctx.finish()
        assert [b[1] for b in blocks] == ref.blocks and na == ref.n_anchors
        outs.append(d)
        ctx.close()
    assert outs[0] == outs[1] == ref.anchor_dict


@pytest.mark.parametrize("k", [32, 47, 63])
def test_two_word_kmers_bit_exact(k):
    """32 <= k <= 63 (BASELINE.json config 5 uses k = 63, 250 bp reads): the unsigned __int128 instances of the kernels"""
    bases, off = common.synthetic(3000, 250, 15000, seed=80 + k, n_rate=0.001)
    _full_compare(bases, off, k, 700, window=1000)
    bases, off = common.synthetic(1200, 100, 5000, seed=90 + k, ragged=True)
    _full_compare(bases, off, k, 500, window=64, batches=2)


def test_c_abi_error_behaviour():
    """status codes and messages instead of exceptions across the boundary (include/leon_dna.h)"""
    import leon_amd
    from leon_amd import capi
    k, rpb = 31, 100
    bases, off = common.synthetic(450, 120, 5000, seed=71)
    bl, solid, tai = common.make_bloom(bases, off, k)
    ctx = _ctx(k, rpb, tai)
    ctx.bloom_upload(bl.bits)
    with pytest.raises(leon_amd.LeonDnaError) as e:                      # wrong bloom size
        ctx.bloom_upload(bl.bits[:-1])
    assert e.value.code == -1
    # a sink that refuses the second block aborts the call with LEON_E_SINK
    seen = []

    def bad_sink(user, block_id, p, size, n_reads):
        seen.append(block_id)
        return 1 if block_id == 1 else 0
    with pytest.raises(leon_amd.LeonDnaError) as e:
        ctx.encode_batch(bases, off[:301], sink=capi.SINK(bad_sink))
    assert e.value.code == -6 and seen == [0, 1]
    ctx.close()
    ctx = _ctx(k, rpb, tai)
    ctx.bloom_upload(bl.bits)
    ctx.next_read = 7                                                    # stream must be continued in order
    with pytest.raises(leon_amd.LeonDnaError) as e:
        ctx.encode_batch(bases, off[:101])
    assert e.value.code == -4
    ctx.next_read = 0
    assert len(ctx.encode_batch(bases, off[:251])) == 3                  # 250 reads: the last block is partial ...
    with pytest.raises(leon_amd.LeonDnaError) as e:                      # ... so no batch may follow
        ctx.encode_batch(bases, off[250:351])
    assert e.value.code == -4
    with pytest.raises(leon_amd.LeonDnaError):                           # shard must be chosen before the first batch
        ctx.set_shard(0, 2)
    d1, n1 = ctx.finish()
    d2, n2 = ctx.finish()                                                # idempotent
    assert d1 == d2 and n1 == n2
    with pytest.raises(leon_amd.LeonDnaError) as e:                      # no batch after finish
        ctx.encode_batch(bases, off[:101])
    assert e.value.code == -4
    ctx.reset_stream()                                                   # a new file on the same context
    ref = O.encode(bases, off, k, rpb, bl, trace=False)
    assert [b[1] for b in ctx.encode_batch(bases, off)] == ref.blocks
    d, na = ctx.finish()
    assert d == ref.anchor_dict
    # the decode entry point: a block table that promises fewer bases than the payload holds, a dictionary stream that is
    # not one, wrong anchors -- status codes and messages, no crash
    blocks = [(i, b, nr) for i, (b, nr) in enumerate(zip(ref.blocks, ref.block_nreads))]
    anchors = capi.anchor_dict_decode(d, na, k)
    nb = [int(off[min(450, (b + 1) * rpb)] - off[b * rpb]) for b in range(len(blocks))]
    assert len(ctx.decode_blocks(anchors, blocks, nb)) == 450
    with pytest.raises(leon_amd.LeonDnaError) as e:
        ctx.decode_blocks(anchors, blocks, [x - 5 for x in nb])
    assert e.value.code == -1 and "does not decode" in str(e.value)
    with pytest.raises(leon_amd.LeonDnaError) as e:
        ctx.decode_blocks(anchors[:10], blocks, nb)
    assert e.value.code == -1
    with pytest.raises(leon_amd.LeonDnaError):
        capi.anchor_dict_decode(bytes(255 - x for x in d), na, k)
    ctx.close()
    # offsets that are not an offsets array are refused before any kernel indexes with them
    ctx = _ctx(k, rpb, tai)
    ctx.bloom_upload(bl.bits)
    bad = np.array(off[:101], dtype=np.uint64)
    bad[50] = bad[52] + 7
    with pytest.raises(leon_amd.LeonDnaError) as e:
        ctx.encode_batch(bases, bad)
    assert e.value.code == -1 and "monotonic" in str(e.value)
    assert len(ctx.encode_batch(bases, off[:101])) == 1               # the context is still usable
    ctx.close()


@pytest.mark.parametrize("k,min_ab,maxkeys", [(31, 3, 0), (31, 1, 50000), (21, 2, 0), (47, 3, 0), (63, 2, 40000)])
def test_device_solid_kmer_counting(k, min_ab, maxkeys):
    """leon_kmer_solid (sort-based counting on the device, hash partitions) == the oracle's exact counter, as a set;
    then the bloom built from it and the stream encoded with it are the oracle's"""
    from leon_amd import capi
    bases, off = common.synthetic(2500, 150, 9000, seed=200 + k, n_rate=0.002, ragged=(k == 21))
    exp = O.count_solid(bases, off, k, min_ab)
    got, hist = capi.kmer_solid(bases, off, k, min_ab, with_histogram=True, max_keys_per_pass=maxkeys)
    w = O.kwords(k)
    assert len(got) == len(exp)
    assert sorted(O.kmers_to_ints(got, k)) == sorted(O.kmers_to_ints(exp, k))
    all1 = O.count_solid(bases, off, k, 1)
    assert int(hist.sum()) == len(all1) // w                      # every distinct k-mer lands in one histogram bin
    assert int(hist[min_ab:].sum()) == len(exp) // w
    tai = max(len(got) // w * common.NB_BITS_PER_KMER, 1000)
    bl = O.Bloom(tai, k)
    bl.insert(exp)
    ctx = _ctx(k, 800, tai)
    ctx.bloom_insert(got)
    assert np.array_equal(ctx.bloom_download(), bl.bits)
    ref = O.encode(bases, off, k, 800, bl, trace=False)
    assert [b[1] for b in ctx.encode_batch(bases, off)] == ref.blocks
    ctx.close()


@pytest.mark.parametrize("k", [5, 31, 32, 63])
def test_solid_kmer_counting_homopolymers_and_padding(k):
    """the counter pads its partition buffers with the all-ones key (all G), which must never be a canonical k-mer nor be
    counted: reads of all G / all C / all A / all T beside ordinary ones, several partitions, inputs smaller than one chunk"""
    from leon_amd import capi
    bases, off = common.synthetic(300, 100, 2000, seed=500 + k)
    reads = [bytes(bases[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    reads += [b"G" * 90] * 5 + [b"C" * 90] * 4 + [b"A" * 70] * 3 + [b"T" * 70] * 3 + [b"G" * (k - 1)] + [b"ACGT" * 30] * 2
    b2, o2 = O.reads_to_arrays(reads)
    for maxkeys in (0, 3000):
        for min_ab in (1, 3):
            exp = O.count_solid(b2, o2, k, min_ab)
            got, hist = capi.kmer_solid(b2, o2, k, min_ab, with_histogram=True, max_keys_per_pass=maxkeys)
            assert sorted(O.kmers_to_ints(got, k)) == sorted(O.kmers_to_ints(exp, k)), (k, maxkeys, min_ab)
            assert int(hist.sum()) == len(O.count_solid(b2, o2, k, 1)) // O.kwords(k)
    one = capi.kmer_solid(*O.reads_to_arrays([b"G" * k]), k, 1)                      # a single k-mer, all G: counted as all C
    assert O.kmers_to_ints(one, k) == [int("01" * k, 2)]


def test_very_long_read_three_byte_numerics():
    # one read longer than 65535 bases: read size / anchor-relative positions need three-byte numerics (numeric models [0..3])
    import synth
    g = synth.make_genome(90000, seed=5)
    b1, off1 = synth.make_reads(g, 1, 70000, seed=6, err=0.001)
    b2, off2 = synth.make_reads(g, 400, 300, seed=7, err=0.01, n_rate=0.001)
    reads = [b2[int(off2[i]):int(off2[i + 1])].tobytes() for i in range(200)] + [b1.tobytes()] + \
            [b2[int(off2[i]):int(off2[i + 1])].tobytes() for i in range(200, 400)]
    bases, off = O.reads_to_arrays(reads)
    _full_compare(bases, off, 31, 150, window=128)


@pytest.mark.parametrize("k,kw", [(31, dict(n_rate=0.003, err=0.02)), (21, dict(ragged=True, n_rate=0.001)), (47, dict(err=0.03)),
                                  (63, dict(ragged=True, n_rate=0.002)), (31, dict(err=0.2))])
def test_device_decoder_round_trip(k, kw):
    """DnaDecoder on the device (one wave per block): decode(encode(x)) == x, the reference's own acceptance test, for reads
    with N, sequencing errors, ragged lengths, reads shorter than k and reads without an anchor; also == the oracle's decoder"""
    from leon_amd import capi
    rpb = 300
    bases, off = common.synthetic(2500, 160 if k < 32 else 220, 9000, seed=90 + k, **kw)
    bl, solid, tai = common.make_bloom(bases, off, k)
    ctx = _ctx(k, rpb, tai)
    ctx.bloom_upload(bl.bits)
    blocks = ctx.encode_batch(bases, off)
    dict_payload, n_anchors = ctx.finish()
    n = len(off) - 1
    reads = [bases[int(off[i]):int(off[i + 1])] for i in range(n)]
    nbases = [sum(len(r) for r in reads[b * rpb:(b + 1) * rpb]) for b in range(len(blocks))]
    anchors = capi.anchor_dict_decode(dict_payload, n_anchors, k)
    assert np.array_equal(anchors, ctx.anchor_kmers(n_anchors))
    got = ctx.decode_blocks(anchors, blocks, nbases)
    norm = lambda r: bytes(c if c in b"ACGT" else ord("N") for c in r)      # any non-ACGT byte is an N in the format
    assert len(got) == n
    for i in range(n):
        assert got[i] == norm(reads[i]), "read %d does not round-trip" % i
    # the oracle's decoder agrees block by block
    oa = O.decode_anchor_dict(dict_payload, n_anchors, k)
    for b in (0, len(blocks) - 1):
        dec = O.decode_block(k, bl, oa, blocks[b][1], blocks[b][2], nbases[b] + 16)
        assert dec == got[b * rpb:b * rpb + blocks[b][2]]
    # a corrupted payload is reported, not crashed on
    bad = list(blocks)
    bad[1] = (bad[1][0], bytes(255 - x for x in bad[1][1]), bad[1][2])
    try:
        out = ctx.decode_blocks(anchors, bad, nbases)
        assert out[rpb:2 * rpb] != [norm(r) for r in reads[rpb:2 * rpb]]
    except capi.LeonDnaError as e:
        assert "does not decode" in str(e)
    ctx.close()


@pytest.mark.parametrize("k", [31, 63])
def test_device_decoder_path_cache_changes_nothing_but_the_time(k, monkeypatch):
    """the decoder's path cache (what the waves learnt from the bloom, shared in HBM) is an accelerator, never a source of
    bases: off, at its default size, and so small that most insertions are dropped (1 MiB: buckets overflow), the decoded reads
    are the same, call after call on the same context; and a context whose bloom changes between two calls forgets the
    entries it learnt from the old one"""
    from leon_amd import capi
    rpb = 250
    bases, off = common.synthetic(6000, 150 if k < 32 else 230, 7000, seed=300 + k, err=0.02, n_rate=0.002)     # ~130x coverage: most walks meet known paths
    bl, solid, tai = common.make_bloom(bases, off, k)
    n = len(off) - 1
    reads = [bases[int(off[i]):int(off[i + 1])] for i in range(n)]
    norm = lambda r: bytes(c if c in b"ACGT" else ord("N") for c in r)
    want = [norm(r) for r in reads]
    ctx = _ctx(k, rpb, tai)
    ctx.bloom_upload(bl.bits)
    blocks = ctx.encode_batch(bases, off)
    d, na = ctx.finish()
    anchors = capi.anchor_dict_decode(d, na, k)
    nbases = [sum(len(r) for r in reads[b * rpb:(b + 1) * rpb]) for b in range(len(blocks))]
    for mb in ("0", None, None, "1", "1", "0"):
        if mb is None:
            monkeypatch.delenv("LEON_DC_CACHE_MB", raising=False)
        else:
            monkeypatch.setenv("LEON_DC_CACHE_MB", mb)
        assert ctx.decode_blocks(anchors, blocks, nbases) == want, "cache setting %r" % mb
    monkeypatch.delenv("LEON_DC_CACHE_MB", raising=False)
    # one block alone, then all of them: entries learnt by an earlier call serve the later one
    assert ctx.decode_blocks(anchors, blocks[:1], nbases[:1]) == want[:blocks[0][2]]
    assert ctx.decode_blocks(anchors, blocks, nbases) == want
    # another read set through the same context: new bloom bits, the old entries must not be used
    bases2, off2 = common.synthetic(3000, 150 if k < 32 else 230, 7000, seed=900 + k, err=0.02)
    bl2 = O.Bloom(tai, k)
    bl2.insert(O.count_solid(bases2, off2, k, 2))
    ctx.reset_stream()
    ctx.bloom_upload(bl2.bits)
    blocks2 = ctx.encode_batch(bases2, off2)
    d2, na2 = ctx.finish()
    reads2 = [bases2[int(off2[i]):int(off2[i + 1])] for i in range(len(off2) - 1)]
    nb2 = [sum(len(r) for r in reads2[b * rpb:(b + 1) * rpb]) for b in range(len(blocks2))]
    assert ctx.decode_blocks(capi.anchor_dict_decode(d2, na2, k), blocks2, nb2) == [norm(r) for r in reads2]
    ctx.close()


def _decode_round_trip(reads, k, rpb, bloom=None):
    from leon_amd import capi
    bases, off = O.reads_to_arrays(reads)
    bl, solid, tai = bloom if bloom is not None else common.make_bloom(bases, off, k)
    ctx = _ctx(k, rpb, tai)
    ctx.bloom_upload(bl.bits)
    blocks = ctx.encode_batch(bases, off)
    d, na = ctx.finish()
    nbases = [sum(len(r) for r in reads[b * rpb:(b + 1) * rpb]) for b in range(len(blocks))]
    got = ctx.decode_blocks(capi.anchor_dict_decode(d, na, k), blocks, nbases)
    ctx.close()
    norm = lambda r: bytes(c if c in b"ACGT" else ord("N") for c in r)
    assert got == [norm(r) for r in reads]


def test_device_decoder_edge_cases():
    """the decoder on the inputs the encoder's edge-case test uses (reads shorter than k, all-N, empty, unanchorable, an empty
    bloom) and on a 70 kb read with a fifth of its positions in error: more N / error positions than a block's own scratch
    holds (8192), so its lists come from the shared pool"""
    import synth
    k = 31
    bases, off = common.synthetic(2000, 150, 10000, seed=21)
    bl = common.make_bloom(bases, off, k)
    extra = [b"ACGT", b"A" * 31, b"N" * 50, b"ACGTTGCA" * 20, b"", b"ACGTNNNNACGT" * 10]
    reads = [bases[int(off[i]):int(off[i + 1])] for i in range(300)]
    mixed = []
    for i, r in enumerate(reads):
        mixed.append(r)
        if i % 50 == 0:
            mixed.append(extra[(i // 50) % len(extra)])
    _decode_round_trip(mixed, k, 100, bloom=bl)
    _decode_round_trip([reads[0]], k, 100, bloom=bl)
    _decode_round_trip(mixed, k, 100, bloom=(O.Bloom(1000, k), None, 1000))
    g = synth.make_genome(90000, seed=5)
    b1, off1 = synth.make_reads(g, 1, 70000, seed=6, err=0.2, n_rate=0.15)
    b2, off2 = synth.make_reads(g, 300, 300, seed=7, err=0.01)
    rr = [b2[int(off2[i]):int(off2[i + 1])].tobytes() for i in range(150)] + [b1.tobytes()] + \
         [b2[int(off2[i]):int(off2[i + 1])].tobytes() for i in range(150, 300)]
    _decode_round_trip(rr, k, 120)


def test_degenerate_batches():
    """empty and near-empty inputs through both directions: only empty reads, a single one-base read, blocks made of reads
    shorter than k, no reads at all"""
    import leon_amd
    k = 31
    bl = (O.Bloom(1000, k), None, 1000)
    _decode_round_trip([b""] * 7, k, 3, bloom=bl)
    _decode_round_trip([b"A"], k, 5, bloom=bl)
    _decode_round_trip([b"ACGTN" * 3, b"", b"T", b"NNNN"] * 5, k, 4, bloom=bl)
    ctx = _ctx(k, 10, 1000)
    assert ctx.encode_batch(b"", np.zeros(1, dtype=np.uint64)) == []
    d, na = ctx.finish()
    assert na == 0 and d == O.encode(b"", np.zeros(1, dtype=np.uint64), k, 10, bl[0], trace=False).anchor_dict
    assert ctx.decode_blocks(np.zeros(0, dtype=np.uint64), [], []) == []
    ctx.close()


def test_crafted_payloads_cannot_make_the_decoder_write_out_of_bounds():
    """values from a payload wrapped the decoder's bound checks (`w + len > wcap`, `apos + k > len`).  Payloads built
    symbol by symbol with the oracle's raw range coder: a read size of 'previous - 5' with previous = 0 (2^64 - 5), an anchor
    position beyond the read, a block that promises more bases than it holds -- each must come back as LEON_E_INVALID."""
    import leon_amd
    from leon_amd import capi
    k, rpb = 31, 4
    bases, off = common.synthetic(64, 100, 3000, seed=3)
    bl, solid, tai = common.make_bloom(bases, off, k)
    ctx = _ctx(k, rpb, tai)
    ctx.bloom_upload(bl.bits)
    blocks = ctx.encode_batch(bases, off)
    d, na = ctx.finish()
    anchors = capi.anchor_dict_decode(d, na, k)
    # model ids of leon_device.h: 0 read type, 4/5/6 delta types, 7 revcomp, numeric group g -> 8 + 9 g (+ byte index)
    sizes = [2, 5, 5, 2, 3, 3, 3, 2] + [256] * 72
    G_ADDR, G_POS, G_SIZE = 8 + 9 * 0, 8 + 9 * 1, 8 + 9 * 3

    def numeric(group, v):
        bs = []
        while True:
            bs.append(v & 255); v >>= 8
            if not v:
                break
        return [(group, len(bs))] + [(group + 1 + i, b) for i, b in enumerate(bs)]

    def payload(syms):
        models = np.array([m for m, _ in syms], dtype=np.uint8)
        vals = np.array([v for _, v in syms], dtype=np.uint8)
        return O.rc_encode_stream(models, vals, sizes)
    crafted = {
        "length wraps": [(0, 0), (4, 2)] + numeric(G_SIZE, 5),                                  # anchored read, size = prev(0) - 5
        "anchor beyond the read": [(0, 0), (4, 0)] + numeric(G_SIZE, 100) + [(5, 0)] + numeric(G_POS, 90) + [(6, 0)] + numeric(G_ADDR, 0) + [(7, 0)],
        "anchor position wraps": [(0, 0), (4, 0)] + numeric(G_SIZE, 100) + [(5, 2)] + numeric(G_POS, 7) + [(6, 0)] + numeric(G_ADDR, 0) + [(7, 0)],
        "no-anchor read longer than the block": [(0, 1)] + numeric(8 + 9 * 2, 1 << 40),
    }
    for what, syms in crafted.items():
        bad = [(0, payload(syms), 1)]
        with pytest.raises(leon_amd.LeonDnaError) as e:
            ctx.decode_blocks(anchors, bad, [100])
        assert e.value.code == -1 and "does not decode" in str(e.value), what
    # a block that holds fewer bases than its table entry says is refused too (w must end exactly at the block's end)
    nb = [int(off[min(64, (b + 1) * rpb)] - off[b * rpb]) for b in range(len(blocks))]
    with pytest.raises(leon_amd.LeonDnaError):
        ctx.decode_blocks(anchors, blocks, [x + 3 for x in nb])
    assert len(ctx.decode_blocks(anchors, blocks, nb)) == 64                                    # and the context still works
    ctx.close()


def test_what_the_gather_delivers_is_checked():
    """leon_dna_set_gather: the parts of the other ranks are INPUT.  A part that does not fit this rank's reads (a status that does not
    exist, a position past the read) or its dictionary (a 'found' k-mer that is no final key here) fails the batch with LEON_E_STATE
    instead of anchoring reads somewhere else; so does a callback that fails; the stream is poisoned until reset and then codes the
    file as if nothing had happened (emulated ranks)."""
    import leon_amd
    from leon_amd import capi
    if os.environ.get("LEON_XCH_LOOKUPS") == "0":
        pytest.skip("LEON_XCH_LOOKUPS=0: the measurement override keeps every look-up on every rank, nothing is gathered")
    k, rpb = 31, 200
    bases, off = common.synthetic(2000, 120, 8000, seed=12)
    bl, solid, tai = common.make_bloom(bases, off, k)
    ref = O.encode(bases, off, k, rpb, bl, trace=False)
    lib = leon_amd.load_library()

    def poke(word):
        def fn(d_buf, part_bytes, world):                         # rank 0 of 2: rank 1's part arrives as `word` for every read
            n = part_bytes // 8
            host = np.full(n, word, dtype=np.uint64)
            assert lib.leon_device_upload(0, C.c_void_p(d_buf + part_bytes), host.ctypes.data_as(C.c_void_p), n * 8) == 0
        return fn

    def boom(d_buf, part_bytes, world):
        raise RuntimeError("the collective failed")
    ctx = _ctx(k, rpb, tai, resolve_window=500)
    ctx.bloom_upload(bl.bits)
    for fn, what in ((poke(7), "do not fit"), (poke((5000 << 8) | 1), "do not fit"), (poke((3 << 8) | 1), "do not fit"), (boom, "gather callback")):
        ctx.set_shard(0, 2)
        ctx.set_exchange(capi.XCH_BY_ANCHOR, lambda d_send, counts: (0, 0))
        ctx.set_gather(fn)
        with pytest.raises(leon_amd.LeonDnaError) as e:
            ctx.encode_batch(bases, off)
        assert e.value.code == -4 and what in str(e.value), str(e.value)
        with pytest.raises(leon_amd.LeonDnaError) as e:
            ctx.encode_batch(bases, off)
        assert e.value.code == -4 and "reset_stream" in str(e.value)
        ctx.reset_stream()
    got = []
    for rank in range(2):
        ctx.reset_stream()
        ctx.set_shard(rank, 2)
        ctx.set_exchange(capi.XCH_EMULATE)
        ctx.set_gather(None)
        got += ctx.encode_batch(bases, off)
        d, na = ctx.finish()
        assert na == ref.n_anchors
    got.sort()
    assert [g[1] for g in got] == ref.blocks
    ctx.close()


def test_failed_batch_poisons_the_stream_until_reset():
    """include/leon_dna.h LEON_E_STATE: a batch refused for its arguments leaves the context untouched; one that fails after it
    began to change the stream (here: the sink) poisons it until leon_dna_reset_stream"""
    import leon_amd
    from leon_amd import capi
    k, rpb = 31, 100
    bases, off = common.synthetic(1000, 100, 5000, seed=4)
    bl, solid, tai = common.make_bloom(bases, off, k)
    ref = O.encode(bases, off, k, rpb, bl, trace=False)
    ctx = _ctx(k, rpb, tai)
    ctx.bloom_upload(bl.bits)
    bad = np.array(off, dtype=np.uint64)
    bad[50] = bad[52] + 7
    with pytest.raises(leon_amd.LeonDnaError) as e:                  # refused before anything changed ...
        ctx.encode_batch(bases, bad)
    assert e.value.code == -1
    assert [b[1] for b in ctx.encode_batch(bases, off)] == ref.blocks   # ... so the stream goes on as if nothing had happened
    ctx.reset_stream()
    seen = []
    refuse = capi.SINK(lambda user, bid, ptr, size, nreads: (seen.append(bid), 1 if bid == 3 else 0)[1])
    with pytest.raises(leon_amd.LeonDnaError) as e:
        ctx.encode_batch(bases, off, sink=refuse)
    assert e.value.code == -6 and seen == [0, 1, 2, 3]
    for call in (lambda: ctx.encode_batch(bases, off), lambda: ctx.finish(), lambda: ctx.header_encode_batch([b"x"])):
        with pytest.raises(leon_amd.LeonDnaError) as e:
            call()
        assert e.value.code == -4 and "reset_stream" in str(e.value)
    ctx.reset_stream()
    assert [b[1] for b in ctx.encode_batch(bases, off)] == ref.blocks
    dd, na = ctx.finish()
    assert dd == ref.anchor_dict and na == ref.n_anchors
    ctx.close()


def test_offsets_are_checked_before_the_uploader_indexes_with_them():
    """leon_dna_encode_batch's uploader thread copies the bases group by group (first 2^17 reads, then 8 windows at a
    time) using intermediate offsets; entries that are not an offsets array -- here a group boundary pointing far outside the
    buffer, in both directions -- must be refused (LEON_E_INVALID), never read through, and leave the stream usable"""
    import leon_amd
    k, rpb, n, L = 21, 5000, 300000, 40
    rng = np.random.default_rng(5)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n * L)].tobytes()
    off = np.arange(n + 1, dtype=np.uint64) * L
    ctx = _ctx(k, rpb, 100000, resolve_window=1 << 14)               # windows of 16 k reads: group boundaries at 16 k, 147 k, 278 k
    for at, val in ((1 << 14, 1 << 60), (1 << 14, 0), ((1 << 14) + 8 * (1 << 14), (n + 5) * L), (123, 7)):
        bad = off.copy()
        bad[at] = val
        with pytest.raises(leon_amd.LeonDnaError) as e:
            ctx.encode_batch(bases, bad)
        assert e.value.code == -1 and "monotonic" in str(e.value), (at, val, str(e.value))
    blocks = ctx.encode_batch(bases, off)                              # refused batches changed nothing
    assert len(blocks) == n // rpb and sum(b[2] for b in blocks) == n
    ctx.close()


def test_reserve_is_only_a_hint():
    """leon_dna_reserve sizes the buffers ahead of the first batch; a batch larger than the reservation (or with more symbols per
    read than it assumed) still encodes, bit-exact"""
    k, rpb = 31, 200
    bases, off = common.synthetic(3000, 150, 9000, seed=33, err=0.08, n_rate=0.01)      # error-dense: many symbols per read
    bl, solid, tai = common.make_bloom(bases, off, k)
    ref = O.encode(bases, off, k, rpb, bl, trace=False)
    for reserve in ((3000, len(bases)), (10, 1000), (1 << 20, 1 << 27)):
        ctx = _ctx(k, rpb, tai)
        ctx.bloom_upload(bl.bits)
        ctx.reserve(*reserve)
        assert [b[1] for b in ctx.encode_batch(bases, off)] == ref.blocks
        d, na = ctx.finish()
        assert d == ref.anchor_dict and na == ref.n_anchors
        ctx.close()


@pytest.mark.parametrize("k", [33, 47, 63])
def test_two_word_keys_under_insert_contention(k):
    """thousands of reads proposing the SAME few two-word keys in one resolution window (a low-complexity read set: every wave's
    inserting lane meets slots another wave is just writing): the lock-free two-step insert must neither duplicate a key nor
    lose one -- the stream stays bit-exact with the oracle"""
    rng = np.random.default_rng(k)
    unit = b"ACGTTGCATGCAAGCTTAGCTAGGATCCAGTCAGTCGATCGATTTAGCGCGATATCGGCTA"
    genome = (unit * 8)[:400]                                     # a 400 bp "genome": a handful of distinct k-mers
    reads = []
    for _ in range(60000):
        s = int(rng.integers(0, len(genome) - 150))
        reads.append(genome[s:s + 150])
    bases, off = O.reads_to_arrays(reads)
    bl, solid, tai = common.make_bloom(bases, off, k, 2)
    ref = O.encode(bases, off, k, 5000, bl, trace=False)
    for window in (0, 1 << 12):
        ctx = _ctx(k, 5000, tai, resolve_window=window)
        ctx.bloom_upload(bl.bits)
        blocks = ctx.encode_batch(bases, off)
        d, na = ctx.finish()
        assert na == ref.n_anchors and d == ref.anchor_dict
        assert [b[1] for b in blocks] == ref.blocks
        ctx.close()

"""-m gpu: BASELINE.json's single-GPU configurations #2 and #3 at FULL size (10 M and 100 M x 150 bp, k = 31) and
configuration #5's workload (250 bp reads, k = 63: two-word k-mers) at ONE GPU's share of it -- the configuration is
500 M reads over 8 GPUs, i.e. 62.5 M reads = 15.6 G bases per GPU --
checked through size-independent properties: decode(encode(x)) == x on sampled blocks (oracle decoder), run-to-run
determinism, shard-union == single stream via a checksum of block checksums, and the device decoder on every base."""
import hashlib
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RPB = 50000
# (reads, k, read length); LEON_FULLSIZE_CASES="n:k:L,..." overrides
CASES = [tuple(int(v) for v in x.split(":")) for x in
         os.environ.get("LEON_FULLSIZE_CASES", "10000000:31:150,100000000:31:150,62500000:63:250").split(",")]


def _checksum(blocks):
    h = hashlib.sha256()
    for bid, payload, nr in sorted(blocks):
        h.update(hashlib.sha256(payload).digest())
        h.update(int(bid).to_bytes(8, "little") + int(nr).to_bytes(4, "little"))
    return h.hexdigest()


@pytest.mark.parametrize("N_READS,K,L", CASES, ids=["%dM_k%d_L%d" % (c[0] // 1000000, c[1], c[2]) for c in CASES])
def test_full_size_properties(N_READS, K, L):
    import torch
    import bench
    import leon_amd
    from leon_amd import capi
    dev = torch.device("cuda", 0)
    G = N_READS * L // 30
    genome = bench.gen_genome(G, dev)
    reads = torch.empty((N_READS, L), dtype=torch.uint8, device=dev)
    for c0 in range((N_READS + bench.CHUNK - 1) // bench.CHUNK):
        lo, hi = c0 * bench.CHUNK, min(N_READS, (c0 + 1) * bench.CHUNK)
        reads[lo:hi] = bench.gen_reads_chunk(genome, c0, bench.CHUNK, 0.01, dev, L=L)[:hi - lo]
    # a few N and a block of garbage reads, so every branch of the path runs at size
    reads[123456, 40] = ord("N"); reads[5_000_001, 0:5] = ord("N")
    g = torch.Generator(device=dev); g.manual_seed(9)
    reads[7_000_000:7_000_200] = torch.tensor([65, 67, 84, 71], dtype=torch.uint8, device=dev)[
        torch.randint(0, 4, (200, L), device=dev, generator=g)]
    offsets = (torch.arange(N_READS + 1, dtype=torch.int64, device=dev) * L).contiguous()
    del genome
    torch.cuda.synchronize()
    d_solid, n_solid = capi.kmer_solid_device(reads.data_ptr(), offsets.data_ptr(), N_READS, K, 3)
    tai = n_solid * 12

    def run(rank=0, world=1, by_anchor=False):
        ctx = leon_amd.DnaEncodeContext(kmer_size=K, reads_per_block=RPB, bloom_tai=tai)
        ctx.set_shard(rank, world)
        if by_anchor:                                          # the walk divided by anchor, this context playing every rank's slice (leon_dna_set_exchange)
            ctx.set_exchange(capi.XCH_EMULATE)
        ctx.bloom_insert_device(d_solid, n_solid)
        blocks = ctx.encode_batch_device(reads.data_ptr(), offsets.data_ptr(), N_READS)
        d, na = ctx.finish()
        bits = ctx.bloom_download() if rank == 0 else None
        st = ctx.stats()
        ctx.close()
        return blocks, d, na, bits, st

    blocks, d, na, bits, st = run()
    n_blocks = N_READS // RPB
    assert [b[0] for b in blocks] == list(range(n_blocks)) and all(b[2] == RPB for b in blocks)
    assert st["n_reads"] == N_READS and st["n_bases"] == N_READS * L and na > 0
    # (1) determinism: a second run gives the same bytes
    blocks2, d2, na2, _, _ = run()
    assert _checksum(blocks2) == _checksum(blocks) and d2 == d and na2 == na
    # (2) shard union == single stream.  BASELINE configuration #4 is this file (100 M x 150 bp, k = 31) over EIGHT ranks: its 8-way
    # split is byte-checked here at its size (the other shapes take 3 ranks), and what every rank reports is what block_range says
    from leon_amd.shard import block_range
    world = 8 if (N_READS, K, L) == (100_000_000, 31, 150) else 3
    u = []
    for r in range(world):
        bl_r, d_r, na_r, _, st_r = run(r, world)
        lo, hi = block_range(r, world, n_blocks)
        assert [b[0] for b in bl_r] == list(range(lo, hi)) and all(b[2] == RPB for b in bl_r)
        assert st_r["n_blocks"] == hi - lo and st_r["n_reads"] == (hi - lo) * RPB and st_r["n_bases"] == (hi - lo) * RPB * L
        assert st_r["n_anchors"] == na and na_r == na and (d_r == d if r == 0 else len(d_r) == 0)
        print("rank %d of %d: %d blocks, device %.0f ms (resolve %.0f, walk %.0f, range coder %.0f)" % (r, world, hi - lo, st_r["ms_total"], st_r["ms_resolve"], st_r["ms_walk"], st_r["ms_rangecoder"]))
        u += bl_r
    assert _checksum(u) == _checksum(blocks)
    if world == 8:
        # ... and the same 8-way job with its walk divided by ANCHOR instead of by block range (what `bench.py --gpus 8` does): every seat's
        # blocks again, each seat walking an eighth of the anchor-sorted reads and receiving the rest of its blocks' events
        u = []
        for r in range(world):
            bl_r, d_r, na_r, _, st_r = run(r, world, by_anchor=True)
            lo, hi = block_range(r, world, n_blocks)
            assert [b[0] for b in bl_r] == list(range(lo, hi)) and na_r == na and (d_r == d if r == 0 else len(d_r) == 0)
            assert 0 < st_r["walk_reads"] < N_READS // 4 and st_r["xch_words_received"] > 0
            print("rank %d of %d, walk by anchor: slice of %d reads walked in %.0f ms, %d words sent, %d received, forming + exchange %.0f ms"
                  % (r, world, st_r["walk_reads"], st_r["ms_walk"], st_r["xch_words_sent"], st_r["xch_words_received"], st_r["ms_exchange"]))
            u += bl_r
        assert _checksum(u) == _checksum(blocks)
    # (3) the reference's own acceptance test (decompress(compress(x)) == x), on sampled blocks, through the oracle's decoder
    bl = O.Bloom(tai, K)
    bl.set_bits(bits)
    anchors = O.decode_anchor_dict(d, na, K)
    assert len(anchors) == na * O.kwords(K)
    host = reads.cpu().numpy()
    for b in (0, 1, 100, 140, n_blocks - 1):               # 100: holds the N reads; 140: the garbage reads
        dec = O.decode_block(K, bl, anchors, blocks[b][1], RPB, RPB * L + 16)
        assert b"".join(dec) == host[b * RPB:(b + 1) * RPB].tobytes(), "block %d does not round-trip" % b
    # (4) the device decoder (DnaDecoder, one wave per block) gives back every read of the file
    import time
    ctx = leon_amd.DnaEncodeContext(kmer_size=K, reads_per_block=RPB, bloom_tai=tai)
    ctx.bloom_upload(bits)
    t0 = time.perf_counter()
    anchors_dev = capi.anchor_dict_decode(d, na, K)
    t1 = time.perf_counter()
    out, lens = ctx.decode_blocks_raw(anchors_dev, blocks, [RPB * L] * n_blocks)
    t2 = time.perf_counter()
    ctx.close()
    print("decode: dictionary %.2f s (host), %d blocks %.2f s (device) = %.1f MB/s" % (t1 - t0, n_blocks, t2 - t1, N_READS * L / 1e6 / (t2 - t1)))
    assert np.array_equal(anchors_dev, np.asarray(anchors, dtype=np.uint64).reshape(-1))      # == the oracle's dictionary decoder
    assert np.all(lens == L)
    hn = host.reshape(-1).copy()
    hn[~np.isin(hn, np.frombuffer(b"ACGT", dtype=np.uint8))] = ord("N")
    assert np.array_equal(out, hn)
    capi.device_free(d_solid)


def test_configuration_5_as_rank_0_of_8_sees_it():
    """BASELINE configuration #5 -- 500 M x 250 bp, k = 63, 8 GPUs -- from the seat of ONE of its ranks, which is all a
    one-GPU box can hold of it and exactly what a rank does under leon_dna_set_shard: ALL 500 M reads go through the
    anchor resolution (replicated, file-order dictionary), rank 0's eighth of every batch's read blocks is walked and
    coded, and rank 0 also carries the file-wide dictionary stream.  The reads are generated on the device batch by
    batch (100 M reads = 25 GB of bases per leon_dna_encode_batch_device call) -- the file never exists as a whole.
    Checked: block ids / sizes of the share; the first batch's share == the same blocks of a one-GPU run over the first
    batch (the bytes do not depend on the sharding); decompress(compress(x)) == x on sampled blocks through the oracle's
    decoder; the 7.2 G-symbol dictionary stream decodes to exactly the anchors the device inserted; 32-bit addresses,
    dictionary capacity and the memory high-water mark.  LEON_CFG5="reads:batch" shrinks it for rehearsals."""
    import time
    import torch
    import bench
    import leon_amd
    from leon_amd import capi
    from leon_amd.shard import block_range
    N, B = (int(v) for v in os.environ.get("LEON_CFG5", "500000000:100000000").split(":"))
    K, L, WORLD, RANK = 63, 250, 8, 0
    assert N % B == 0 and B % bench.CHUNK == 0 and B % RPB == 0
    dev = torch.device("cuda", 0)
    t_start = time.perf_counter()
    G = N * L // 30
    genome = bench.gen_genome(G, dev)
    tai = (G - K + 1) * 12
    ctx = leon_amd.DnaEncodeContext(kmer_size=K, reads_per_block=RPB, bloom_tai=tai)
    # the bloom: the genome's own k-mers (the step before the path; the device counter's hash partitions over 94 G k-mers are
    # minutes of work and not what this test is about)
    for lo in range(0, G - K + 1, 1 << 26):
        km = bench.genome_kmers_chunk(genome, lo, min(G - K + 1, lo + (1 << 26)), K)
        torch.cuda.synchronize()
        ctx.bloom_insert_device(km.data_ptr(), km.shape[0])
        del km
    t_bloom = time.perf_counter() - t_start
    ctx.set_shard(RANK, WORLD)
    ctx.reserve(B, B * L)
    buf = torch.empty((B, L), dtype=torch.uint8, device=dev)
    off = (torch.arange(B + 1, dtype=torch.int64, device=dev) * L).contiguous()
    bpb = B // RPB                                           # blocks per batch
    lb0, lb1 = block_range(RANK, WORLD, bpb)                 # this rank's blocks of every batch
    sample = {0: (0, 1, lb1 - 1), 1: (lb0 + 17,), N // B - 1: (lb1 - 1,)}     # batch -> blocks of the share kept for the round trip
    kept_reads = {}

    def fill(batch):
        for c in range(B // bench.CHUNK):
            buf[c * bench.CHUNK:(c + 1) * bench.CHUNK] = bench.gen_reads_chunk(genome, batch * (B // bench.CHUNK) + c, bench.CHUNK, 0.01, dev, L=L)
        if batch == 0:                                       # a few N and a run of garbage reads inside the share
            buf[123456, 40] = ord("N"); buf[1, 0:5] = ord("N")
            g = torch.Generator(device=dev); g.manual_seed(9)
            buf[60_000:60_200] = torch.tensor([65, 67, 84, 71], dtype=torch.uint8, device=dev)[torch.randint(0, 4, (200, L), device=dev, generator=g)]
        torch.cuda.synchronize()

    blocks, stage, high_water = [], {}, 0
    t_gen = t_enc = 0.0
    for batch in range(N // B):
        t0 = time.perf_counter()
        fill(batch)
        for b in sample.get(batch, ()):
            kept_reads[batch * bpb + b] = buf[b * RPB:(b + 1) * RPB].cpu().numpy()
        t1 = time.perf_counter()
        got = ctx.encode_batch_device(buf.data_ptr(), off.data_ptr(), B)
        t2 = time.perf_counter()
        t_gen += t1 - t0; t_enc += t2 - t1
        assert [g[0] for g in got] == list(range(batch * bpb + lb0, batch * bpb + lb1)) and all(g[2] == RPB for g in got)
        blocks += got
        st = ctx.stats()
        assert st["n_reads"] == (lb1 - lb0) * RPB and st["n_bases"] == (lb1 - lb0) * RPB * L
        for key, v in st.items():
            if key.startswith("ms_") or key in ("resolve_rounds", "resolve_windows"):
                stage[key] = stage.get(key, 0) + v
        free_b, total_b = torch.cuda.mem_get_info()
        high_water = max(high_water, total_b - free_b)
    t0 = time.perf_counter()
    d, na = ctx.finish()
    t_finish = time.perf_counter() - t0
    st = ctx.stats()
    print("configuration #5, rank 0 of 8: %d reads in %d batches; bloom %.1f s, generation %.1f s, encode calls %.1f s + finish (dictionary chain) %.1f s; "
          "device ms: pack %.0f resolve %.0f (replicated: all reads) sort %.0f walk %.0f symbols %.0f range coder %.0f d2h %.0f = %.0f; chain busy %.0f ms; "
          "%d anchors, %d resolve rounds in %d windows; %d blocks, %.1f MB of payload + %.1f MB of dictionary stream; HBM high-water %.1f GB"
          % (N, N // B, t_bloom, t_gen, t_enc, t_finish, stage["ms_pack"], stage["ms_resolve"], stage["ms_sort"], stage["ms_walk"], stage["ms_symbols"],
             stage["ms_rangecoder"], stage["ms_d2h"], stage["ms_total"], st["ms_chain_busy"], na, stage["resolve_rounds"], stage["resolve_windows"],
             len(blocks), sum(len(b[1]) for b in blocks) / 1e6, len(d) / 1e6, high_water / 1e9))
    assert len(blocks) == (N // B) * (lb1 - lb0)
    assert 0 < na < 2 ** 32 and 4 * na <= 2 ** 32          # 32-bit anchor addresses; the dictionary (load <= 1/4) stays within its 2^32 slots
    assert high_water < 240e9, "a rank of configuration #5 must fit one MI355X (288 GB) with room to spare"
    # the dictionary stream (rank 0 writes it) decodes to exactly what the device inserted, in insertion order
    t0 = time.perf_counter()
    anchors = capi.anchor_dict_decode(d, na, K)
    t_dict = time.perf_counter() - t0
    assert np.array_equal(anchors, ctx.anchor_kmers(na))
    # decompress(compress(x)) == x on the sampled blocks, through the oracle's decoder
    bl = O.Bloom(tai, K)
    bl.set_bits(ctx.bloom_download())
    by_id = {b[0]: b for b in blocks}
    for bid, want in sorted(kept_reads.items()):
        dec = O.decode_block(K, bl, anchors, by_id[bid][1], RPB, RPB * L + 16)
        w = want.copy()
        w[~np.isin(w, np.frombuffer(b"ACGT", dtype=np.uint8))] = ord("N")
        assert b"".join(dec) == w.tobytes(), "block %d does not round-trip" % bid
    del bl
    # the share's bytes do not depend on the sharding: the first batch alone on one GPU gives the same first blocks
    ctx.reset_stream()
    ctx.set_shard(0, 1)
    fill(0)
    one = ctx.encode_batch_device(buf.data_ptr(), off.data_ptr(), B)
    assert [(b[0], b[1]) for b in one[lb0:lb1]] == [(b[0], b[1]) for b in blocks[:lb1 - lb0]]
    ctx.close()
    print("dictionary decode %.1f s; whole test %.1f s" % (t_dict, time.perf_counter() - t_start))


@pytest.mark.parametrize("order,k,L", [("sorted", 31, 150), ("pairs", 63, 250)])
def test_full_size_structured_file(order, k, L):
    """10 M reads with the structure real files have -- a genome with dispersed and tandem repeats, 10 % PCR duplicates, a third of the reads
    piled on a tenth of the genome, ragged lengths, reads in genome-position order (what `samtools sort | samtools fastq` writes) or mates
    interleaved -- through the whole device path (bench.structured_case): run-to-run determinism by a checksum of block checksums, the
    union of 3 shards with the walk divided by anchor == the single stream, the device decoder gives every base back, sampled blocks
    through the ORACLE's decoder.  (Stand-in for the reference's acceptance input, a real SRA file: /root/reference/scripts/simple_test.sh:11-34.)"""
    import torch
    import bench
    import leon_amd
    from leon_amd import capi
    dev = torch.device("cuda", 0)
    N = 10_000_000
    genome = bench.gen_structured_genome(N * L * 3 // 4 // 30, dev)
    flat, offsets = bench.gen_structured_reads(genome, N, L, dev, order=order)
    del genome
    torch.cuda.synchronize()
    d_solid, n_solid = capi.kmer_solid_device(flat.data_ptr(), offsets.data_ptr(), N, k, 3)
    tai = n_solid * 12

    def run(rank=0, world=1):
        ctx = leon_amd.DnaEncodeContext(kmer_size=k, reads_per_block=RPB, bloom_tai=tai)
        ctx.set_shard(rank, world)
        if world > 1:
            ctx.set_exchange(capi.XCH_EMULATE)
        ctx.bloom_insert_device(d_solid, n_solid)
        blocks = ctx.encode_batch_device(flat.data_ptr(), offsets.data_ptr(), N)
        d, na = ctx.finish()
        st = ctx.stats()
        return ctx, blocks, d, na, st

    ctx, blocks, d, na, st = run()
    assert [b[0] for b in blocks] == list(range(N // RPB)) and st["n_reads"] == N and na > 0
    if order == "sorted":
        assert st["resolve_chain_reads"] > N // 10          # position order leaves the rounds a long chain: the sequential pass took it
    # the device decoder on every base
    off_h = offsets.cpu().numpy()
    nb = [int(off_h[(b[0] + 1) * RPB] - off_h[b[0] * RPB]) for b in blocks]
    anchors = capi.anchor_dict_decode(d, na, k)
    out_bases, out_lens = ctx.decode_blocks_raw(anchors, blocks, nb)
    flat_h = flat.cpu().numpy()
    assert np.array_equal(out_lens.astype(np.int64), np.diff(off_h)) and np.array_equal(out_bases, flat_h)
    bits = ctx.bloom_download()
    ctx.close()
    # sampled blocks through the oracle's decoder (an independent implementation of the format)
    bl = O.Bloom(tai, k)
    bl.set_bits(bits)
    oa = O.decode_anchor_dict(d, na, k)
    for bid in (0, len(blocks) // 2, len(blocks) - 1):
        dec = O.decode_block(k, bl, oa, blocks[bid][1], RPB, 10 ** 8)
        r0 = bid * RPB
        for j in (0, 1, RPB // 2, RPB - 1):
            assert dec[j] == flat_h[off_h[r0 + j]:off_h[r0 + j + 1]].tobytes()
    # determinism, and the 3-way split with the walk divided by anchor
    want = _checksum(blocks)
    del out_bases, blocks
    ctx2, blocks2, d2, na2, _ = run()
    ctx2.close()
    assert _checksum(blocks2) == want and d2 == d and na2 == na
    del blocks2
    union = []
    for rank in range(3):
        c3, b3, d3, na3, st3 = run(rank, 3)
        c3.close()
        union += b3
        assert na3 == na and (d3 == d if rank == 0 else len(d3) == 0)
    assert _checksum(union) == want
    capi.device_free(d_solid)

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

    def run(rank=0, world=1):
        ctx = leon_amd.DnaEncodeContext(kmer_size=K, reads_per_block=RPB, bloom_tai=tai)
        ctx.set_shard(rank, world)
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
    # (2) shard union == single stream
    u = []
    for r in range(3):
        bl_r, d_r, na_r, _, _ = run(r, 3)
        assert na_r == na and (d_r == d if r == 0 else len(d_r) == 0)
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

"""-m gpu: a seeded random sweep over the path's parameters -- k (one- and two-word k-mers), block size, bloom geometry
(number of hashes, block bits), read length / raggedness / error and N rates, resolution window.  For every draw:
the HIP encoder's bytes == the oracle's, and the device decoder gives the input back; every third draw also as an N-rank job
(walk and window look-ups divided among emulated ranks): the union of the ranks' blocks == the oracle's."""
import random

import pytest

import common
import oracle_lib as O
import synth

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("rc_chains")]


@pytest.mark.parametrize("chunk", range(4))
def test_random_parameter_sweep(chunk):
    import leon_amd
    from leon_amd import capi
    rnd = random.Random(2024 + chunk)
    for it in range(8):
        k = rnd.choice([5, 9, 15, 21, 27, 31, 32, 33, 40, 47, 55, 63])
        rpb = rnd.choice([1, 7, 50, 333, 1000])
        n_hash, nbits = rnd.choice([1, 3, 7, 10]), rnd.choice([6, 9, 12, 14])
        L = max(rnd.choice([k, k + 1, 40, 100, 151, 260]), 8)
        n = rnd.choice([1, 17, 400, 1500])
        kw = dict(err=rnd.choice([0, 0.01, 0.08]), n_rate=rnd.choice([0, 0.002, 0.05]), ragged=rnd.random() < 0.5)
        what = dict(k=k, rpb=rpb, n_hash=n_hash, nbits=nbits, L=L, n=n, **kw)
        # every draw picks a structure: the i.i.d. genome with reads at random places in random order (rounds 1-4's only input), or a genome
        # with dispersed and tandem repeats read in one of the orders real files come in, with PCR duplicates and coverage skew
        shape = rnd.choice(["iid", "sorted", "pairs", "random", "sorted-strands"])
        G = rnd.choice([300, 3000, 20000])
        if shape == "iid":
            bases, off = common.synthetic(n, L, G, seed=1000 * chunk + it, **kw)
        else:
            g = synth.make_structured_genome(G, seed=1000 * chunk + it, dispersed=rnd.choice([0, 3, 10]), tandem=rnd.choice([0, 4, 30]))
            b_, off = synth.make_structured_reads(g, n, L, seed=1000 * chunk + it + 1, order=shape, dup_rate=rnd.choice([0, 0.1, 0.5]),
                                                  skew=rnd.choice([0, 0.5]), stride=rnd.choice([None, None, 1, 7]), **kw)
            bases = b_.tobytes()
        what["shape"] = shape
        bl, solid, tai = common.make_bloom(bases, off, k, rnd.choice([1, 2, 3]), n_hash, nbits)
        ref = O.encode(bases, off, k, rpb, bl, trace=False)
        ctx = leon_amd.DnaEncodeContext(kmer_size=k, reads_per_block=rpb, bloom_tai=tai, bloom_n_hash=n_hash,
                                        bloom_block_nbits=nbits, resolve_window=rnd.choice([0, 16, 300]))
        ctx.bloom_upload(bl.bits)
        blocks = ctx.encode_batch(bases, off)
        d, na = ctx.finish()
        assert [b[1] for b in blocks] == ref.blocks and d == ref.anchor_dict and na == ref.n_anchors, what
        reads = [bases[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
        nb = [sum(len(r) for r in reads[b * rpb:(b + 1) * rpb]) for b in range(len(blocks))]
        got = ctx.decode_blocks(capi.anchor_dict_decode(d, na, k), blocks, nb)
        assert got == [bytes(c if c in b"ACGT" else ord("N") for c in r) for r in reads], what
        ctx.close()
        if it % 3 == 0:
            # the same file as an N-rank job (leon_dna_set_shard + LEON_XCH_EMULATE: the walk divided by anchor and the resolution's
            # window look-ups divided among the ranks, every context playing the other ranks' parts): the union of the ranks' blocks
            world = rnd.choice([2, 3, 5])
            union = []
            for rank in range(world):
                ctx = leon_amd.DnaEncodeContext(kmer_size=k, reads_per_block=rpb, bloom_tai=tai, bloom_n_hash=n_hash,
                                                bloom_block_nbits=nbits, resolve_window=rnd.choice([0, 16, 300]))
                ctx.set_shard(rank, world)
                ctx.set_exchange(capi.XCH_EMULATE)
                ctx.bloom_upload(bl.bits)
                union += ctx.encode_batch(bases, off)
                d2, na2 = ctx.finish()
                assert na2 == ref.n_anchors and (d2 == ref.anchor_dict if rank == 0 else len(d2) == 0), (what, world, rank)
                ctx.close()
            union.sort()
            assert [b[0] for b in union] == list(range(len(ref.blocks))) and [b[1] for b in union] == ref.blocks, (what, world)

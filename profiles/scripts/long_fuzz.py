"""One-off, longer version of tests/test_gpu_fuzz.py (more draws, larger read sets, more windows / batches): the HIP encoder's
bytes == the oracle's and decode == input for every draw.  Run on the GPU box: python profiles/scripts/long_fuzz.py [draws]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import common  # noqa: E402
import synth  # noqa: E402
import oracle_lib as O  # noqa: E402
import leon_amd  # noqa: E402
from leon_amd import capi  # noqa: E402

draws = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rnd = random.Random(int(os.environ.get("LEON_FUZZ_SEED", 777)))
t0 = time.time()
for it in range(draws):
    k = rnd.choice([5, 9, 15, 21, 27, 31, 32, 33, 40, 47, 55, 63]) if it % 2 else rnd.randrange(5, 64)      # (every k: every geometry of the final keys' filter)
    rpb = rnd.choice([1, 7, 50, 333, 1000, 5000])
    n_hash, nbits = rnd.choice([1, 3, 7, 10]), rnd.choice([6, 9, 12, 14])
    L = max(rnd.choice([k, k + 1, 40, 100, 151, 260, 600]), 8)
    n = rnd.choice([1, 17, 400, 1500, 6000, 20000])
    kw = dict(err=rnd.choice([0, 0.01, 0.08]), n_rate=rnd.choice([0, 0.002, 0.05]), ragged=rnd.random() < 0.5)
    window = rnd.choice([0, 16, 64, 300, 4096])
    G = rnd.choice([300, 3000, 20000, 200000])
    what = dict(it=it, k=k, rpb=rpb, n_hash=n_hash, nbits=nbits, L=L, n=n, window=window, G=G, **kw)
    # (round 5) every draw picks a structure, as tests/test_gpu_fuzz.py does, a chunk size for the sequential resolution pass and a form of the block coder
    shape = rnd.choice(["iid", "iid", "sorted", "pairs", "random", "sorted-strands"])
    if shape == "iid":
        bases, off = common.synthetic(n, L, G, seed=5000 + it, **kw)
    else:
        g = synth.make_structured_genome(G, seed=5000 + it, dispersed=rnd.choice([0, 3, 10]), tandem=rnd.choice([0, 4, 30]))
        b_, off = synth.make_structured_reads(g, n, L, seed=5001 + it, order=shape, dup_rate=rnd.choice([0, 0.1, 0.5]),
                                              skew=rnd.choice([0, 0.5]), stride=rnd.choice([None, None, 1, 7]), **kw)
        bases = b_.tobytes()
    env = {"LEON_CHAIN_CHUNK": rnd.choice([None, None, "1", "64", "1000"]), "LEON_RC_GROUP": rnd.choice([None, "1", "2", "4", "8"]),
           "LEON_RC_HOST_BLOCKS": rnd.choice([None, "0", "400"]), "LEON_RC_CMP": rnd.choice([None, "0"]),
           "LEON_RESOLVE_ROUNDS": rnd.choice([None, None, "1", "9"])}
    for kk, vv in env.items():
        if vv is None: os.environ.pop(kk, None)
        else: os.environ[kk] = vv
    what.update(shape=shape, **{kk: vv for kk, vv in env.items() if vv is not None})
    bl, solid, tai = common.make_bloom(bases, off, k, rnd.choice([1, 2, 3]), n_hash, nbits)
    ref = O.encode(bases, off, k, rpb, bl, trace=False)
    ctx = leon_amd.DnaEncodeContext(kmer_size=k, reads_per_block=rpb, bloom_tai=tai, bloom_n_hash=n_hash, bloom_block_nbits=nbits, resolve_window=window)
    ctx.bloom_upload(bl.bits)
    # one batch, or several batches of whole blocks
    nreads = len(off) - 1
    if rnd.random() < 0.5 and nreads > 2 * rpb:
        cut = (rnd.randrange(1, nreads // rpb)) * rpb
        b1 = ctx.encode_batch(bases, np.asarray(off[:cut + 1]))
        sub_off = np.asarray(off[cut:])
        b2 = ctx.encode_batch(bases, sub_off)
        blocks = b1 + b2
    else:
        blocks = ctx.encode_batch(bases, off)
    d, na = ctx.finish()
    assert [b[1] for b in blocks] == ref.blocks and d == ref.anchor_dict and na == ref.n_anchors, what
    reads = [bases[int(off[i]):int(off[i + 1])] for i in range(nreads)]
    nb = [sum(len(r) for r in reads[b * rpb:(b + 1) * rpb]) for b in range(len(blocks))]
    got = ctx.decode_blocks(capi.anchor_dict_decode(d, na, k), blocks, nb)
    assert got == [bytes(c if c in b"ACGT" else ord("N") for c in r) for r in reads], what
    ctx.close()
    if it % 4 == 0:                                              # the same file as an N-rank job: walk and window look-ups divided among emulated ranks
        world = rnd.choice([2, 3, 5, 8])
        union = []
        for rank in range(world):
            ctx = leon_amd.DnaEncodeContext(kmer_size=k, reads_per_block=rpb, bloom_tai=tai, bloom_n_hash=n_hash, bloom_block_nbits=nbits, resolve_window=window)
            ctx.set_shard(rank, world)
            ctx.set_exchange(capi.XCH_EMULATE)
            ctx.bloom_upload(bl.bits)
            union += ctx.encode_batch(bases, off)
            d2, na2 = ctx.finish()
            assert na2 == na and (d2 == d if rank == 0 else len(d2) == 0), (what, world, rank)
            ctx.close()
        union.sort()
        assert [b[0] for b in union] == list(range(len(ref.blocks))) and [b[1] for b in union] == ref.blocks, (what, world)
    if it % 10 == 9:
        print("draw %d ok (%.0f s)" % (it + 1, time.time() - t0), flush=True)
print("long fuzz: %d draws, all bit-exact and round-tripping, %.0f s" % (draws, time.time() - t0))

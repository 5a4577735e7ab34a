"""VERDICT r2 item 5, measured: does walking the anchor groups in GENOME order (instead of insertion order) let neighbouring
waves share bloom windows in L2 / Infinity Cache?  The synthetic reads' true positions give the ideal adjacency order any
overlap-graph / list-ranking pass over the dictionary could at best reproduce: key = genome position of the read's anchor
k-mer (both strands of an anchor share it).  Variants: the product's order (anchor address), the genome order, and each with
the XCD-contiguous workgroup mapping (LEON_WALK_XCD=1).  Prints ms_walk per variant; run under rocprofv3 --pmc FETCH_SIZE
for the traffic (WALK_ORDER_ONLY=<variant index> runs one variant's timed step alone)."""
import ctypes
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import leon_amd  # noqa: E402
from leon_amd import capi  # noqa: E402

N = int(os.environ.get("SWEEP_READS", 100_000_000))
K, L, RPB = bench.K, bench.L, bench.RPB
dev = torch.device("cuda", 0)
G = N * L // 30
genome = bench.gen_genome(G, dev)
reads = torch.empty((N, L), dtype=torch.uint8, device=dev)
starts = torch.empty(N, dtype=torch.int64, device=dev)
rev = torch.empty(N, dtype=torch.bool, device=dev)
for c0 in range((N + bench.CHUNK - 1) // bench.CHUNK):
    lo, hi = c0 * bench.CHUNK, min(N, (c0 + 1) * bench.CHUNK)
    reads[lo:hi] = bench.gen_reads_chunk(genome, c0, bench.CHUNK, 0.01, dev)[:hi - lo]
    g = torch.Generator(device=dev)                      # the generator's first two draws again: read starts and strands
    g.manual_seed(43 + c0)
    st = torch.randint(0, G - L + 1, (bench.CHUNK,), device=dev, generator=g)
    rv = torch.rand(bench.CHUNK, device=dev, generator=g) < 0.5
    starts[lo:hi] = st[:hi - lo]; rev[lo:hi] = rv[:hi - lo]
offsets = (torch.arange(N + 1, dtype=torch.int64, device=dev) * L).contiguous()
del genome
torch.cuda.synchronize()
d_solid, n_solid = capi.kmer_solid_device(reads.data_ptr(), offsets.data_ptr(), N, K, 3)
tai = n_solid * 12
only = os.environ.get("WALK_ORDER_ONLY")


def run(keys, xcd, label):
    if xcd:
        os.environ["LEON_WALK_XCD"] = "1"
    else:
        os.environ.pop("LEON_WALK_XCD", None)
    ctx = leon_amd.DnaEncodeContext(kmer_size=K, reads_per_block=RPB, bloom_tai=tai)
    ctx.reserve(N, N * L)
    ctx.bloom_insert_device(d_solid, n_solid)
    best, digest, apos = None, None, None
    for step in range(1 if only is not None else 3):
        ctx.reset_stream()
        h = hashlib.sha256()

        def sink(user, bid, p, size, nr):
            h.update(hashlib.sha256(ctypes.string_at(p, size)).digest() + int(bid).to_bytes(8, "little"))
            return 0
        if keys is not None:
            ctx._chk(ctx.lib.leon_dna_debug_walk_order(ctx.h, ctypes.c_void_p(keys.data_ptr())))
        ctx.encode_batch_device(reads.data_ptr(), offsets.data_ptr(), N, sink=capi.SINK(sink))
        d, na = ctx.finish()
        st = ctx.stats()
        if best is None or st["ms_walk"] < best["ms_walk"]:
            best = st
        digest = h.hexdigest()[:16] + ":" + hashlib.sha256(d).hexdigest()[:16]
    if keys is None and not xcd:
        apos = torch.from_numpy(ctx.trace_anchors(N)[0]).to(dev)
    ctx.close()
    print(json.dumps({"order": label, "xcd_contiguous": bool(xcd), "ms_walk": round(best["ms_walk"], 1), "ms_sort": round(best["ms_sort"], 1),
                      "ms_total": round(best["ms_total"], 1), "bytes": digest}), flush=True)
    return apos


apos = run(None, False, "anchor address (product)") if only in (None, "0") else None
if only in (None, "1"):
    run(None, True, "anchor address (product)")
if only in (None, "2", "3"):
    if apos is None:                                      # anchors of every read: one untimed pass
        os.environ.pop("LEON_WALK_XCD", None)
        c0 = leon_amd.DnaEncodeContext(kmer_size=K, reads_per_block=RPB, bloom_tai=tai)
        c0.bloom_insert_device(d_solid, n_solid)
        c0.encode_batch_device(reads.data_ptr(), offsets.data_ptr(), N, sink=capi.SINK(lambda *a: 0))
        c0.finish()
        apos = torch.from_numpy(c0.trace_anchors(N)[0]).to(dev)
        c0.close()
    a = apos.to(torch.int64)
    gpos = torch.where(rev, starts + (L - K) - a, starts + a)
    keys = torch.where(a >= 0, gpos, torch.full_like(gpos, (1 << 40) - 1)).contiguous()
    del a, gpos
    if only in (None, "2"):
        run(keys, False, "true genome position of the anchor")
    if only in (None, "3"):
        run(keys, True, "true genome position of the anchor")
capi.device_free(d_solid)

"""Stage times of the encode path under measurement overrides of the anchor-resolution stage (filters, load hints, window
size): the read set and the bloom are built once, every variant gets a fresh context and two steps.  Also checks that
every variant produces the same bytes (checksum of block checksums + dictionary stream)."""
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
for c0 in range((N + bench.CHUNK - 1) // bench.CHUNK):
    lo, hi = c0 * bench.CHUNK, min(N, (c0 + 1) * bench.CHUNK)
    reads[lo:hi] = bench.gen_reads_chunk(genome, c0, bench.CHUNK, 0.01, dev)[:hi - lo]
offsets = (torch.arange(N + 1, dtype=torch.int64, device=dev) * L).contiguous()
del genome
torch.cuda.synchronize()
d_solid, n_solid = capi.kmer_solid_device(reads.data_ptr(), offsets.data_ptr(), N, K, 3)
tai = n_solid * 12
VARIANTS = json.loads(os.environ.get("SWEEP_VARIANTS", "[{}]"))
KEYS = ("LEON_LOOKUP_BLOCKS_PER_CU", "LEON_DICT_NT", "LEON_FBITS_LOG2", "LEON_FBITS2_LOG2", "LEON_RESOLVE_WINDOW", "LEON_WALK_ORDER")
ref = None
for var in VARIANTS:
    for k_ in KEYS:
        os.environ.pop(k_, None)
    for k_, v in var.items():
        os.environ[k_] = str(v)
    ctx = leon_amd.DnaEncodeContext(kmer_size=K, reads_per_block=RPB, bloom_tai=tai, resolve_window=int(var.get("LEON_RESOLVE_WINDOW", 0)))
    ctx.reserve(N, N * L)
    ctx.bloom_insert_device(d_solid, n_solid)
    h = hashlib.sha256()

    def sink(user, bid, p, size, nr):
        h.update(hashlib.sha256(ctypes.string_at(p, size)).digest() + int(bid).to_bytes(8, "little"))
        return 0
    cb = capi.SINK(sink)
    best = None
    for step in range(3):
        ctx.reset_stream()
        h = hashlib.sha256()
        ctx.encode_batch_device(reads.data_ptr(), offsets.data_ptr(), N, sink=cb)
        d, na = ctx.finish()
        st = ctx.stats()
        if step and (best is None or st["ms_total"] < best["ms_total"]):
            best = st
    digest = h.hexdigest()[:16] + ":" + hashlib.sha256(d).hexdigest()[:16]
    if ref is None:
        ref = digest
    print(json.dumps({"variant": var, "same_bytes": digest == ref, "ms": {k_: round(v, 1) for k_, v in best.items() if k_.startswith("ms_")},
                      "rounds": best["resolve_rounds"], "windows": best["resolve_windows"]}), flush=True)
    ctx.close()
capi.device_free(d_solid)

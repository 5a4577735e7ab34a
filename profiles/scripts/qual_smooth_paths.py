"""leon_qual_smooth_batch_device at 100 M reads three ways: every read for itself in file order (LEON_QUAL_ORDER=0), the reads of a
locus sharing their probes aligned on their minimizers, and -- right after an encode of the same reads -- aligned on their anchors."""
import ctypes, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, leon_amd
from leon_amd import capi
N = int(os.environ.get("SWEEP_READS", 100_000_000)); K, L, RPB = bench.K, bench.L, bench.RPB
dev = torch.device("cuda", 0)
genome = bench.gen_genome(N * L // 30, dev)
reads = torch.empty((N, L), dtype=torch.uint8, device=dev)
for c0 in range((N + bench.CHUNK - 1) // bench.CHUNK):
    lo, hi = c0 * bench.CHUNK, min(N, (c0 + 1) * bench.CHUNK)
    reads[lo:hi] = bench.gen_reads_chunk(genome, c0, bench.CHUNK, 0.01, dev)[:hi - lo]
offsets = (torch.arange(N + 1, dtype=torch.int64, device=dev) * L).contiguous()
del genome
d_solid, n_solid = capi.kmer_solid_device(reads.data_ptr(), offsets.data_ptr(), N, K, 3)
ctx = leon_amd.DnaEncodeContext(kmer_size=K, reads_per_block=RPB, bloom_tai=n_solid * 12)
ctx.bloom_insert_device(d_solid, n_solid)
ref = None
def smooth(label):
    global ref
    quals = torch.full((N * L,), 70, dtype=torch.uint8, device=dev)
    quals[::7] = 40                                            # some below '@': only the coverage can smooth those
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = ctx.lib.leon_qual_smooth_batch_device(ctx.h, ctypes.c_void_p(reads.data_ptr()), ctypes.c_void_p(offsets.data_ptr()), N, ctypes.c_void_p(quals.data_ptr()))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert rc == 0
    h = int(quals.to(torch.int64).sum().item())
    if ref is None: ref = h
    print(json.dumps({"path": label, "ms": round(dt * 1e3, 1), "GBps": round(N * L / 1e9 / dt, 1), "same_bytes_checksum": h == ref}), flush=True)
os.environ["LEON_QUAL_ORDER"] = "0"; smooth("file order, every read for itself")
del os.environ["LEON_QUAL_ORDER"]; smooth("minimizer-anchored lockstep"); smooth("minimizer-anchored lockstep (again)")
ctx.encode_batch_device(reads.data_ptr(), offsets.data_ptr(), N, sink=capi.SINK(lambda *a: 0)); ctx.finish()
smooth("anchor-aligned lockstep (after the encode of the same reads)")

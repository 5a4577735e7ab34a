#!/usr/bin/env python3
"""bench.py -- compressed-input MB/s of the DNA encode path on synthetic 150 bp reads (BASELINE.json metric).

One "step" = one pass of the hot path (pack -> anchor resolution -> walk -> symbols -> range coder -> blocks
handed to the sink, plus leon_dna_finish) over the whole synthetic read set, inputs resident in HBM, fed to the
library as ONE ordered stream in batches of at most --batch-reads reads (whole read blocks; 100 M by default, so
BASELINE's 100 M x 150 bp job is one batch and the 500 M x 250 bp job of configuration #5 is five).
N > 1: one process per GPU (torch.distributed / RCCL); rank 0 builds the bloom and broadcasts it over xGMI; every
rank holds the read set, resolves the anchors of all reads (replicated: file-order dictionary semantics without any
exchange, leon_dna_set_shard) and walks / codes its contiguous range of every batch's read blocks; rank 0 writes the
dictionary stream.  The job is one file of fixed size: strong scaling.  Prints ONE JSON line on rank 0.
`python bench.py --gpus N` without a launcher starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
itself, as a child process, before anything touches the GPU, and relays rank 0's line.

LEON_BENCH_AS_RANK=r:N (one process, one GPU): this process takes the seat of rank r of an N-rank job -- leon_dna_set_shard(r, N),
every collective of the N-rank code path on a process group of one (RCCL by default) -- so that what such a rank does, how long it
takes and how much memory it needs can be measured on a one-GPU box; the line says `"as_rank": "r:N"` and its `value` is what the
N-GPU job would report if this rank were its slowest.

`value` is the HBM-resident figure (the bench contract: inputs resident when the timed region starts).  SURVEY 8(d)
also wants the figure with the H2D copy inside: `pcie_inclusive` (one step through leon_dna_encode_batch, reads in
pageable host memory) is measured in every default N = 1 run, beside `verify` (checksum of block checksums), `decode`
(the whole file back through the device decoder, compared with the input) and `end_to_end` (`leon -c -lossless` and
`leon -d -test-file` on a FASTQ in /dev/shm) -- none of them is ever `value`; --quick skips them.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K = int(os.environ.get("LEON_BENCH_K", 31))          # 31 / 150: BASELINE's configurations 2-4; 63 / 250: configuration 5's read shape
L = int(os.environ.get("LEON_BENCH_L", 150))
N_HASH = 7
BITS_PER_KMER = 12
ABUNDANCE = 3
RPB = 50000
CHUNK = 1_000_000           # reads generated per torch call; also the unit of the rank partition
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("LEON_BENCH_READS", 100_000_000)),
                    help="total reads of the job (BASELINE.json metric: 100M x 150 bp)")
    ap.add_argument("--batch-reads", type=int, default=int(os.environ.get("LEON_BENCH_BATCH_READS", 100_000_000)),
                    help="reads handed to the library per leon_dna_encode_batch_device call (rounded down to whole read blocks)")
    ap.add_argument("--genome", type=int, default=0, help="genome length; default reads*L/30 (30x coverage)")
    ap.add_argument("--bloom-from", choices=("auto", "count", "genome"), default="auto",
                    help="where the bloom's solid k-mers come from, outside the timed region: `count` = the device k-mer counter over the "
                         "reads (abundance >= 3), `genome` = the genome's own k-mers (the counter's partitions would take minutes on "
                         "configuration #5's 94 G k-mers); auto = count up to 200 M reads")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("LEON_BENCH_CPU_SAMPLE", 250_000)),
                    help="reads PER WORKER timed through the CPU restatement on rank 0 at N=1, one worker per allowed CPU (0 = skip)")
    ap.add_argument("--err", type=float, default=0.01)
    ap.add_argument("--kmer-max-keys", type=int, default=0, help="k-mers sorted per pass by the solid k-mer counter (0 = sized by the library)")
    ap.add_argument("--walk-by", choices=("auto", "block", "anchor"), default=os.environ.get("LEON_BENCH_WALK_BY", "auto"),
                    help="N > 1: how a batch's walk is divided among the ranks -- `block`: every rank walks the reads of its own block range (no exchange); "
                         "`anchor`: rank r walks the r-th slice of the batch's reads sorted by anchor address and the walk events reach the rank that codes the "
                         "read's block through one all-to-all per batch (leon_dna_set_exchange); auto = anchor")
    ap.add_argument("--quick", action="store_true",
                    help="only the timed steps, roofline and cpu_baseline: skip pcie_inclusive / verify / decode / streams / end_to_end")
    ap.add_argument("--decode", action="store_true", help="(always on at N = 1 unless --quick) decode the whole file on the device and compare it with the input")
    ap.add_argument("--verify", action="store_true", help="(always on unless --quick) one more UNTIMED step whose sink hashes every block")
    ap.add_argument("--host-input", action="store_true", help="(always on at N = 1 unless --quick) one step through leon_dna_encode_batch, PCIe included")
    ap.add_argument("--e2e-reads", type=int, default=int(os.environ.get("LEON_BENCH_E2E_READS", 10_000_000)),
                    help="reads of the FASTQ that `end_to_end` takes through the leon CLI (0 = skip)")
    ap.add_argument("--other-configs", default=os.environ.get("LEON_BENCH_OTHER_CONFIGS"),
                    help="reads:k:L,... -- BASELINE's other single-GPU configurations, 5 timed steps each after the headline steps, reported under `other_configs` "
                         "(default, for the headline workload only: 10000000:31:150 = configuration #2, 20000000:63:250 = configuration #5's read shape); '' = none")
    ap.add_argument("--structured-reads", type=int, default=int(os.environ.get("LEON_BENCH_STRUCTURED_READS", 10_000_000)),
                    help="reads of the `structured` entry: a file with repeats, duplicates, coverage skew and ragged lengths in genome-position order (0 = skip)")
    ap.add_argument("--streams", action="store_true",
                    help="(always on at N = 1 unless --quick) also time the kernels of the streams either side of the DNA stream on device-resident synthetic data: the header "
                         "stream (records + range coder, 10 M SRA-style headers) and the lossy quality smoothing (the workload's reads); "
                         "reported as `streams`, never as value")
    return ap.parse_args()


def gen_genome(G, device):
    g = torch.Generator(device=device)
    g.manual_seed(42)
    return torch.randint(0, 4, (G,), dtype=torch.uint8, device=device, generator=g)   # gatb codes A0 C1 T2 G3


def gen_reads_chunk(genome, chunk_id, n, err, device, L=None):
    """reads of chunk `chunk_id` (same bytes whatever the world size): uint8 [n, L] ASCII"""
    L = L or globals()["L"]
    g = torch.Generator(device=device)
    g.manual_seed(43 + chunk_id)
    G = genome.numel()
    starts = torch.randint(0, G - L + 1, (n,), device=device, generator=g)
    idx = starts[:, None] + torch.arange(L, device=device)[None, :]
    codes = genome[idx]
    rev = torch.rand(n, device=device, generator=g) < 0.5
    codes = torch.where(rev[:, None], codes.flip(1) ^ 2, codes)                     # reverse complement
    if err > 0:
        m = torch.rand(n, L, device=device, generator=g) < err
        sub = (codes + torch.randint(1, 4, (n, L), dtype=torch.uint8, device=device, generator=g)) & 3
        codes = torch.where(m, sub, codes)
    lut = torch.tensor([65, 67, 84, 71], dtype=torch.uint8, device=device)          # A C T G
    return lut[codes.long()]


def gen_structured_genome(G, device, families=0, microsats=0, seed=52):
    """a genome with the structure real ones have: dispersed repeats (`families` segments of 0.1-5 kbp copied to 2-6 other places, half of
    the copies reverse-complemented) and tandem repeats (`microsats` units of 1-6 bp repeated over 40-400 bp) written into an i.i.d. one.
    Defaults scale with G: one family per 250 kbp, one microsatellite per 25 kbp."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    genome = torch.randint(0, 4, (G,), dtype=torch.uint8, device=device, generator=g)
    rnd = np.random.default_rng(seed + 1)
    for _ in range(families or max(4, G // 250_000)):
        seg = int(min(max(100, G // 8), rnd.integers(100, 5001)))
        src = int(rnd.integers(0, G - seg))
        piece = genome[src:src + seg].clone()
        for _ in range(int(rnd.integers(2, 7))):
            dst = int(rnd.integers(0, G - seg))
            genome[dst:dst + seg] = (piece.flip(0) ^ 2) if rnd.random() < 0.5 else piece
    for _ in range(microsats or max(4, G // 25_000)):
        unit = torch.from_numpy(rnd.integers(0, 4, size=int(rnd.integers(1, 7))).astype(np.uint8)).to(device)
        span = int(min(max(8, G // 10), rnd.integers(40, 401)))
        dst = int(rnd.integers(0, max(1, G - span)))
        genome[dst:dst + span] = unit.repeat(span // unit.numel() + 1)[:span]
    return genome


def gen_structured_reads(genome, n, L, device, order="sorted", err=0.01, dup_rate=0.1, skew=0.3, ragged=True, seed=53):
    """reads with the structure real files have, flat ASCII + offsets (int64) on the device:
    order "sorted" = by start position (a position-sorted BAM turned back into FASTQ), "pairs" = mates interleaved (forward read, then a
    reverse read ~2.2 read lengths downstream), "random"; dup_rate = share of PCR duplicates (another read's placement, own errors);
    skew = share of the reads that fall into a tenth of the genome; ragged = lengths uniform in [L/2, L]."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    G = genome.numel()
    u = torch.rand(n, device=device, generator=g)
    if skew > 0:
        hot = torch.rand(n, device=device, generator=g) < skew
        u = torch.where(hot, 0.45 + 0.1 * u, u)
    starts = (u * (G - L + 1)).long()
    rev = torch.rand(n, device=device, generator=g) < 0.5
    if order == "pairs":
        half = n // 2
        starts[1:2 * half:2] = torch.clamp(starts[0:2 * half:2] + int(2.2 * L), max=G - L)
        rev[0:2 * half:2] = False
        rev[1:2 * half:2] = True
    if dup_rate > 0:
        dup = torch.rand(n, device=device, generator=g) < dup_rate
        src = torch.randint(0, n, (n,), device=device, generator=g)
        starts = torch.where(dup, starts[src], starts)
        rev = torch.where(dup, rev[src], rev)
    if order == "sorted":
        starts, o = torch.sort(starts, stable=True)
        rev = rev[o]
    lens = torch.randint(L // 2, L + 1, (n,), device=device, generator=g) if ragged else torch.full((n,), L, device=device, dtype=torch.int64)
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=device)
    offsets[1:] = torch.cumsum(lens, 0)
    total = int(offsets[-1].item())
    flat = torch.empty(total, dtype=torch.uint8, device=device)
    lut = torch.tensor([65, 67, 84, 71], dtype=torch.uint8, device=device)          # A C T G
    ar = torch.arange(L, device=device)[None, :]
    for lo in range(0, n, CHUNK):
        hi = min(n, lo + CHUNK)
        codes = genome[starts[lo:hi, None] + ar]
        codes = torch.where(rev[lo:hi, None], codes.flip(1) ^ 2, codes)              # reverse complement of the L-base window, then cut to length
        if err > 0:
            m = torch.rand(hi - lo, L, device=device, generator=g) < err
            sub = (codes + torch.randint(1, 4, (hi - lo, L), dtype=torch.uint8, device=device, generator=g)) & 3
            codes = torch.where(m, sub, codes)
        keep = ar < lens[lo:hi, None]
        flat[int(offsets[lo].item()):int(offsets[hi].item())] = lut[codes.long()][keep]
        del codes, keep
    return flat, offsets.contiguous()


def structured_case(n=10_000_000, order="sorted", k=None, L_=None, G=0, decode=True, device=None, steps=2, **kw):
    """One file with real-genome structure through the whole device path, never `value`: repeats, duplicates, coverage skew, ragged
    lengths, reads in `order`.  Returns the stage times of the last of `steps` passes, what the anchor resolution did (parallel rounds,
    reads left to the sequential pass), and -- decode=True -- whether the device decoder gives every base back."""
    import leon_amd
    from leon_amd import capi
    k = k or K
    L_ = L_ or L
    device = device or torch.device("cuda", torch.cuda.current_device())
    n = max(RPB, n // RPB * RPB)
    G = G or max(n * L_ * 3 // 4 // 30, 10 * L_)             # ~30x at the mean ragged length
    genome = gen_structured_genome(G, device)
    flat, offsets = gen_structured_reads(genome, n, L_, device, order=order, **kw)
    del genome
    n_bases = int(offsets[-1].item())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d_solid, n_solid = capi.kmer_solid_device(flat.data_ptr(), offsets.data_ptr(), n, k, ABUNDANCE, device_id=device.index or 0)
    ctx = leon_amd.DnaEncodeContext(kmer_size=k, reads_per_block=RPB, bloom_tai=max(n_solid, 100) * BITS_PER_KMER, bloom_n_hash=N_HASH,
                                    device_id=device.index or 0, resolve_window=int(os.environ.get("LEON_RESOLVE_WINDOW", 0)))
    ctx.bloom_insert_device(d_solid, n_solid)
    capi.device_free(d_solid)
    bloom_s = time.perf_counter() - t0
    kept = []
    keep = capi.SINK(lambda user, bid, ptr, size, nreads: (kept.append((int(bid), ctypes.string_at(ptr, size), int(nreads))), 0)[1])
    best = None
    for _ in range(steps):
        kept.clear()
        ctx.reset_stream()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.encode_batch_device(flat.data_ptr(), offsets.data_ptr(), n, sink=keep)
        st = ctx.stats()
        dstream, n_anch = ctx.finish()
        dt = time.perf_counter() - t0
        st2 = ctx.stats()
        best = dt if best is None else min(best, dt)
    out = {"reads": n, "bases": n_bases, "read_len": "%d..%d" % (L_ // 2, L_) if kw.get("ragged", True) else L_, "kmer_size": k, "order": order,
           "genome": G, "dup_rate": kw.get("dup_rate", 0.1), "skew": kw.get("skew", 0.3), "solid_kmers": int(n_solid),
           "step_ms": round(best * 1e3, 1), "MBps": round(n_bases / 1e6 / best, 1),
           "stages_ms": {s_: round(st[s_], 2) for s_ in ("ms_pack", "ms_resolve", "ms_resolve_chain", "ms_sort", "ms_walk", "ms_symbols", "ms_rangecoder", "ms_d2h", "ms_total")},
           "chain_busy_ms": round(st2["ms_chain_busy"], 1),
           "resolve": {"windows": int(st["resolve_windows"]), "parallel_rounds": int(st["resolve_rounds"]),
                       "reads_left_to_the_sequential_pass": int(st["resolve_chain_reads"]), "windows_with_a_sequential_pass": int(st["resolve_chain_windows"])},
           "anchors": int(n_anch), "bits_per_base": round(8.0 * (sum(len(b[1]) for b in kept) + len(dstream)) / n_bases, 4),
           "kmer_count_and_bloom_s": round(bloom_s, 2)}
    if decode:
        anchors = capi.anchor_dict_decode(dstream, n_anch, k)
        off_h = offsets.cpu().numpy()
        nb = [int(off_h[min(n, (b[0] + 1) * RPB)] - off_h[b[0] * RPB]) for b in sorted(kept)]
        t0 = time.perf_counter()
        out_bases, out_lens = ctx.decode_blocks_raw(anchors, kept, nb)
        out["decode_s"] = round(time.perf_counter() - t0, 2)
        out["decode_equals_input"] = bool(np.array_equal(out_bases, flat.cpu().numpy())) and bool(np.array_equal(out_lens.astype(np.int64), np.diff(off_h)))
    ctx.close()
    return out


def other_config(n, k, L_, device, steps=5, warmup=1):
    """another single-GPU configuration of BASELINE.json driven through the same timed loop as the headline one:
    its own reads (the headline generator at this shape), solid k-mers, bloom and context; `steps` timed steps back to back after `warmup`,
    bracketed once; reported under `other_configs`, never as `value`"""
    import leon_amd
    from leon_amd import capi
    n = max(RPB, n // RPB * RPB)
    G = max(n * L_ // 30, 10 * L_)
    genome = gen_genome(G, device)
    reads = torch.empty((n, L_), dtype=torch.uint8, device=device)
    for c0 in range((n + CHUNK - 1) // CHUNK):
        lo, hi = c0 * CHUNK, min(n, (c0 + 1) * CHUNK)
        reads[lo:hi] = gen_reads_chunk(genome, c0, CHUNK, 0.01, device, L=L_)[:hi - lo]
    del genome
    offsets = (torch.arange(n + 1, dtype=torch.int64, device=device) * L_).contiguous()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d_solid, n_solid = capi.kmer_solid_device(reads.data_ptr(), offsets.data_ptr(), n, k, ABUNDANCE, device_id=device.index or 0)
    ctx = leon_amd.DnaEncodeContext(kmer_size=k, reads_per_block=RPB, bloom_tai=n_solid * BITS_PER_KMER, bloom_n_hash=N_HASH, device_id=device.index or 0)
    ctx.reserve(n, n * L_)
    ctx.bloom_insert_device(d_solid, n_solid)
    capi.device_free(d_solid)
    prep_s = time.perf_counter() - t0
    got = [0, 0]

    def sink(user, block_id, p, size, n_reads):
        got[0] += size; got[1] += 1
        return 0
    cb = capi.SINK(sink)

    def one():
        got[0] = got[1] = 0
        ctx.reset_stream()
        ctx.encode_batch_device(reads.data_ptr(), offsets.data_ptr(), n, sink=cb)
        st = ctx.stats()
        dsz, na = ctx.finish(copy=False)
        return st, ctx.stats(), dsz, na
    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = t_prev = time.perf_counter()
    per, walk, dev, chain = [], [], [], []
    for _ in range(steps):
        st, st2, dsz, na = one()
        now = time.perf_counter()
        per.append((now - t_prev) * 1e3); t_prev = now
        walk.append(st["ms_walk"]); dev.append(st["ms_total"]); chain.append(st2["ms_chain_busy"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    wb = ((L_ + 3) // 4) * 2 + (L_ - k) * 64
    ach = n * wb / (float(np.mean(walk)) * 1e-3) / 1e9
    out = {"workload": "%d x %d bp synthetic reads, k=%d, genome %d bp (30x), 1%% substitutions" % (n, L_, k, G),
           "value": round(n * L_ / 1e6 / dt, 1), "unit": "MB/s", "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3, 2),
           "step_ms": [round(v, 1) for v in per], "device_ms": round(float(np.mean(dev)), 2), "host_chain_ms": round(float(np.mean(chain)), 2),
           "stages_ms": {k_: round(st[k_], 2) for k_ in ("ms_pack", "ms_resolve", "ms_resolve_chain", "ms_sort", "ms_walk", "ms_symbols", "ms_rangecoder", "ms_d2h", "ms_total")},
           "roofline": {"kernel": "k_walk", "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                        "traffic": None, "algorithmic_bytes_per_launch": n * wb, "avg_launch_ms": round(float(np.mean(walk)), 3)},
           "anchors": int(na), "blocks": got[1], "bits_per_base": round(8.0 * (got[0] + dsz) / (n * L_), 4), "solid_kmers": int(n_solid),
           "resolve_rounds": int(st["resolve_rounds"]), "kmer_count_and_bloom_s": round(prep_s, 2)}
    ctx.close()
    return out


def genome_kmers_chunk(genome, lo, hi, k):
    """canonical k-mers starting at genome positions [lo, hi) as the C-ABI wants them (one uint64 below k = 32, else
    (low word, high word) pairs), computed with torch on the device: the `--bloom-from genome` source of solid k-mers"""
    dev = genome.device
    n = hi - lo
    seg = genome[lo:hi + k - 1].to(torch.int64)
    W = 2 if k >= 32 else 1
    fw = [torch.zeros(n, dtype=torch.int64, device=dev) for _ in range(W)]      # [low, high]
    rc = [torch.zeros(n, dtype=torch.int64, device=dev) for _ in range(W)]
    for j in range(k):
        b = seg[j:j + n]                                  # base j of every k-mer: goes to bit 2*(k-1-j) of the forward k-mer
        sh = 2 * (k - 1 - j)
        fw[sh // 64] |= b << (sh % 64)
        sh = 2 * j                                        # and, complemented (code ^ 2), to bit 2*j of the reverse complement
        rc[sh // 64] |= (b ^ 2) << (sh % 64)
    if W == 1:
        return torch.minimum(fw[0], rc[0]).contiguous()   # k <= 31: both fit 62 bits, signed compare == unsigned compare
    sgn = torch.tensor(-2 ** 63, dtype=torch.int64, device=dev)
    hi_lt = fw[1] < rc[1]                                 # high words hold 2k - 64 <= 62 bits: non-negative
    hi_eq = fw[1] == rc[1]
    lo_lt = (fw[0] ^ sgn) < (rc[0] ^ sgn)                 # unsigned compare of the low words
    take_f = hi_lt | (hi_eq & lo_lt)
    out = torch.empty((n, 2), dtype=torch.int64, device=dev)
    out[:, 0] = torch.where(take_f, fw[0], rc[0])
    out[:, 1] = torch.where(take_f, fw[1], rc[1])
    return out.contiguous()


def walk_bytes_per_read(k):
    """algorithmic bytes of ONE read in the dominant kernel k_walk, SURVEY.md section 8(d)'s per-unit terms for it:
    the 2-bit read in, ONE 64-byte bloom line per extension step (L-k steps), the per-position event bytes out.
    (DESIGN.md 'Roofline accounting': the restated bloom spreads a step's 7 probes over a 514-byte window, so
    without reuse between reads a step could cost up to 7 sectors; the PMC traffic is reported next to it.)"""
    return (L + 3) // 4 + (L - k) * 64 + (L + 3) // 4


def walk_source_id():
    """identity of the k_walk the library was built from: profiles/walk_traffic.json names the one its PMC figure was
    measured on, and `roofline.traffic` is null for any other"""
    import hashlib
    h = hashlib.sha256()
    for f in ("dna_kernels.hip", "leon_device.h"):
        h.update(open(os.path.join(ROOT, "leon_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


class Watch:
    """Every collective of the N-rank path (and every call that holds one: the C-ABI's exchange / gather callbacks) runs inside
    `with WATCH("name"):`.  A thread looks at what is open once a second; one that has been open for more than LEON_BENCH_COLL_TIMEOUT
    seconds (180: the longest legitimate wait is a rank without the dictionary chain waiting for rank 0's, 8 s per step at configuration #5) is reported BY NAME on stderr, with the rank and the device memory in use, and the process exits non-zero at once --
    os._exit, never a re-exec -- so that the launcher tears the job down instead of the job sitting in a collective until the driver's
    limit.  An exception inside the block is reported the same way (a failed rank must not leave the others waiting for it)."""
    def __init__(self, rank, device):
        import threading
        self.rank, self.device = rank, device
        self.limit = float(os.environ.get("LEON_BENCH_COLL_TIMEOUT", 180))
        self.open = None
        self.log = {}                       # name -> [calls, seconds]
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _mem(self):
        try:
            f, t = torch.cuda.mem_get_info(self.device)
            return round((t - f) / 1e9, 1)
        except Exception:                   # noqa: BLE001
            return None

    def die(self, what, code=3):
        print(json.dumps({"bench_error": what, "rank": self.rank, "hbm_in_use_GB": self._mem()}), file=sys.stderr)
        sys.stderr.flush(); sys.stdout.flush()
        os._exit(code)

    def _run(self):
        while True:
            time.sleep(1.0)
            cur = self.open
            if cur and time.perf_counter() - cur[1] > self.limit:
                self.die("collective `%s` did not complete in %.0f s (LEON_BENCH_COLL_TIMEOUT)" % (cur[0], self.limit))

    def __call__(self, name):
        w = self

        class _Ctx:
            def __enter__(self_):
                w.open = (name, time.perf_counter())

            def __exit__(self_, et, ev, tb):
                t0 = w.open[1]
                w.open = None
                rec = w.log.setdefault(name.split(" [")[0], [0, 0.0])
                rec[0] += 1; rec[1] += time.perf_counter() - t0
                if et is not None and not issubclass(et, SystemExit):
                    w.die("collective `%s` failed: %s: %s" % (name, et.__name__, str(ev)[:300]))
                return False
        return _Ctx()

    def summary(self):
        return {k: {"calls": v[0], "ms": round(v[1] * 1e3, 2)} for k, v in self.log.items()}


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of a torch.distributed.run child
    (this process has not touched the GPU and never will) and relay rank 0's JSON line"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    for l in p.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1])
    sys.stdout.flush()
    raise SystemExit(p.returncode if p.returncode else (0 if lines else 1))


_LINE_FD = None


def claim_stdout():
    """The driver takes ONE JSON line from rank 0's stdout.  Libraries write there too -- RCCL prints a five-line version banner on stdout when its
    first communicator starts (seen on one box of the pool, not on others) -- so from here on descriptor 1 IS stderr for everything in this
    process, and the line goes out through a copy of the real stdout kept aside."""
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)


def emit_line(text):
    sys.stdout.flush()
    os.write(_LINE_FD if _LINE_FD is not None else 1, (text + "\n").encode())


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a)
    claim_stdout()
    pg_world = int(os.environ.get("WORLD_SIZE", "1"))        # the process group: one process per GPU
    pg_rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    as_rank = os.environ.get("LEON_BENCH_AS_RANK")            # "r:N": this ONE process is rank r of an N-rank job (rehearsal on a one-GPU box)
    if pg_world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, pg_world))
    # the job's shape as leon_dna_set_shard sees it: (rank, world) = the process group's, or the seat LEON_BENCH_AS_RANK names
    rank, world = pg_rank, pg_world
    if as_rank:
        if pg_world != 1:
            raise SystemExit("LEON_BENCH_AS_RANK is for a single process (--gpus 1)")
        try:
            rank, world = (int(v) for v in as_rank.split(":"))
        except ValueError:
            raise SystemExit("LEON_BENCH_AS_RANK=%r: expected r:N" % as_rank)
        if not 0 <= rank < world:
            raise SystemExit("LEON_BENCH_AS_RANK=%r: rank out of range" % as_rank)
        # A rank of a real job shares its node's CPUs with the other ranks, and the library gives it 1 / world of them for its host threads
        # (capi.hip rc_host_threads); the seat is alone on this box and stands for a rank on a node that grants EVERY rank this box's CPUs
        # (8 x 16 on the pool's 8-GPU nodes, if they are sized like its one-GPU boxes): the threads one process gets, named.
        os.environ.setdefault("LEON_RC_HOST_THREADS", str(max(2, min(32, host_info(0, 0)["cpus_usable"] - 1))))
    import torch.distributed as dist
    n_dev = torch.cuda.device_count()
    if local >= n_dev:                       # rehearsal of the N>1 path on a one-GPU box (LEON_BENCH_BACKEND=gloo)
        local = local % max(n_dev, 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # (LEON_BENCH_FORCE_DIST=1: a one-rank run under torch.distributed.run still goes through the process group, the RCCL broadcast
    # of the bloom and the reductions -- the N > 1 code on the one GPU a test box has; LEON_BENCH_AS_RANK does the same by itself)
    use_dist = pg_world > 1 or (os.environ.get("LEON_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ) or bool(as_rank)
    backend = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LEON_BENCH_BACKEND", "nccl")     # nccl = RCCL over xGMI; gloo only to rehearse
        kw = {}
        if "RANK" not in os.environ:                              # a process group of one, started here (LEON_BENCH_AS_RANK without a launcher)
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            kw = dict(init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=float(os.environ.get("LEON_BENCH_COLL_TIMEOUT", 180)))   # (the process group's own; WATCH below names the collective)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, **kw)
        else:
            dist.init_process_group(backend, **kw)
    WATCH = Watch(pg_rank, device)
    is_root = pg_rank == 0                   # the process that prints the line (and, in the job, the rank that codes the dictionary: rank 0)

    import leon_amd
    from leon_amd import capi
    from leon_amd.shard import block_range

    n_total = (a.reads // RPB) * RPB or RPB
    G = a.genome or max(n_total * L // 30, 10 * L)
    n_blocks = n_total // RPB
    B = max(RPB, min(a.batch_reads, n_total) // RPB * RPB)
    batches = [(lo, min(n_total, lo + B)) for lo in range(0, n_total, B)]
    # this rank's blocks: its contiguous share of EVERY batch's blocks (leon_dna_set_shard)
    n_local = sum((lambda r: r[1] - r[0])(block_range(rank, world, (hi - lo) // RPB)) for lo, hi in batches) * RPB
    bloom_from = a.bloom_from if a.bloom_from != "auto" else ("count" if n_total <= 200_000_000 else "genome")

    genome = gen_genome(G, device)
    # the whole read set on every rank (same seeds everywhere), generated chunk by chunk
    reads = torch.empty((n_total, L), dtype=torch.uint8, device=device)
    for c0 in range(0, (n_total + CHUNK - 1) // CHUNK):
        lo, hi = c0 * CHUNK, min(n_total, (c0 + 1) * CHUNK)
        chunk = gen_reads_chunk(genome, c0, CHUNK, a.err, device)
        reads[lo:hi] = chunk[:hi - lo]
        del chunk
    offsets = (torch.arange(n_total + 1, dtype=torch.int64, device=device) * L).contiguous()
    torch.cuda.synchronize()
    if use_dist:                            # the replicated anchor resolution needs the SAME reads on every rank
        chk = torch.stack([reads[::97].sum(dtype=torch.int64), reads[-1].sum(dtype=torch.int64)])
        lo_, hi_ = chk.clone(), chk.clone()
        with WATCH("all_reduce (the ranks' read sets compared)"):
            dist.all_reduce(lo_, op=dist.ReduceOp.MIN); dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
            torch.cuda.synchronize()
        if not torch.equal(lo_, hi_):
            raise SystemExit("rank %d: the synthetic read set differs between ranks" % rank)

    # bloom = the reads' solid k-mers (abundance >= 3, Leon's `-abundance 3`), counted on the device by rank 0
    # (leon_kmer_solid_device, the DSK stand-in: outside the timed region, it is the step before the path); for read sets
    # beyond the counter's comfortable size (--bloom-from genome) the genome's own k-mers, inserted slice by slice
    t_b = time.time()
    n_solid_t = torch.zeros(1, dtype=torch.int64, device=device)
    d_solid = 0
    if bloom_from == "count":
        if is_root:
            d_solid, n_solid = capi.kmer_solid_device(reads.data_ptr(), offsets.data_ptr(), n_total, K, ABUNDANCE, device_id=local,
                                                      max_keys_per_pass=a.kmer_max_keys)
            n_solid_t[0] = n_solid
    else:
        n_solid_t[0] = G - K + 1
    count_s = time.time() - t_b
    if use_dist:
        with WATCH("broadcast (solid k-mer count)"):
            dist.broadcast(n_solid_t, src=0)
            torch.cuda.synchronize()
    n_solid = int(n_solid_t.item())
    tai = n_solid * BITS_PER_KMER
    ctx = leon_amd.DnaEncodeContext(kmer_size=K, reads_per_block=RPB, bloom_tai=tai, bloom_n_hash=N_HASH, device_id=local,
                                    resolve_window=int(os.environ.get("LEON_RESOLVE_WINDOW", 0)))
    ctx.set_shard(rank, world)
    walk_by = "block" if world == 1 else ("anchor" if a.walk_by == "auto" else a.walk_by)
    if walk_by == "anchor":
        # a seat of an N-rank job on its own (LEON_BENCH_AS_RANK) plays the other ranks' slices itself; real ranks exchange
        if as_rank:
            ctx.set_exchange(capi.XCH_EMULATE)
        else:
            ctx.set_exchange(capi.XCH_BY_ANCHOR, make_exchange(dist, device, backend, pg_rank, pg_world, capi, WATCH))
            if os.environ.get("LEON_XCH_LOOKUPS", "1") != "0":     # the resolution's window look-ups divided among the ranks as well
                ctx.set_gather(make_gather(dist, device, backend, pg_rank, pg_world, capi, WATCH))
    ctx.reserve(B, B * L)                      # what a host does while it parses: the first step then allocates nothing large
    nbytes = ctx.bloom_nbytes
    bcast_ms = 0.0
    if is_root:
        if bloom_from == "count":
            ctx.bloom_insert_device(d_solid, n_solid)
            capi.device_free(d_solid)
        else:
            SL = 1 << 26
            for lo in range(0, G - K + 1, SL):
                km = genome_kmers_chunk(genome, lo, min(G - K + 1, lo + SL), K)
                torch.cuda.synchronize()
                ctx.bloom_insert_device(km.data_ptr(), km.shape[0])
                del km
    del genome
    if use_dist:                           # RCCL broadcast of the bloom over xGMI, device to device
        bits = torch.empty(nbytes, dtype=torch.uint8, device=device)
        if is_root:
            ctx.bloom_download_device(bits.data_ptr(), nbytes)
        if os.environ.get("LEON_BENCH_TEST_STALL_RANK") == str(pg_rank):     # test hook: this rank never reaches the broadcast (tests/test_gpu_multiprocess.py)
            time.sleep(10 ** 6)
        with WATCH("barrier (before the bloom broadcast)"):
            torch.cuda.synchronize(); dist.barrier()
        t0 = time.time()
        with WATCH("broadcast (bloom, %d bytes)" % nbytes):
            dist.broadcast(bits, src=0)
            torch.cuda.synchronize()
        bcast_ms = (time.time() - t0) * 1e3
        if rank != 0:                          # (a rehearsed seat r > 0 takes the receiving side too: the same bits back into its context)
            ctx.bloom_upload_device(bits.data_ptr(), nbytes)
        del bits
    bloom_s = time.time() - t_b

    payload = [0, 0]

    def sink(user, block_id, p, size, n_reads):
        payload[0] += size
        payload[1] += 1
        return 0
    cb = capi.SINK(sink)
    STAGES = ("ms_pack", "ms_resolve", "ms_resolve_chain", "ms_sort", "ms_walk", "ms_symbols", "ms_rangecoder", "ms_d2h", "ms_total", "ms_exchange", "ms_exchange_call", "ms_gather_call", "ms_emulated", "ms_emulated_lookups")
    COUNTS = ("n_symbols", "resolve_rounds", "resolve_windows", "walk_launches", "xch_words_sent", "xch_words_received", "walk_reads", "resolve_chain_reads")

    def encode_stream(the_sink):
        """one file: every batch in order through leon_dna_encode_batch_device; returns the stage times summed over the batches"""
        t0 = time.perf_counter()
        ctx.reset_stream()
        t1 = time.perf_counter()
        acc = {k: 0.0 for k in STAGES + COUNTS}
        for lo, hi in batches:
            ctx.encode_batch_device(reads.data_ptr(), offsets.data_ptr() + 8 * lo, hi - lo, sink=the_sink)
            st = ctx.stats()
            for k in STAGES + COUNTS:
                acc[k] += st[k]
        acc["host_reset_ms"], acc["host_encode_calls_ms"] = (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3
        return acc

    def step():
        payload[0] = payload[1] = 0
        acc = encode_stream(cb)
        t0 = time.perf_counter()
        d_size, na = ctx.finish(copy=False)                  # the stream stays in the context, as for a C caller
        acc["host_finish_ms"] = (time.perf_counter() - t0) * 1e3
        st = ctx.stats()
        acc["ms_anchor_wait"], acc["ms_chain_busy"] = st["ms_anchor_wait"], st["ms_chain_busy"]
        return d_size, na, acc

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            with WATCH("barrier (around the timed steps)"):
                dist.barrier()
                torch.cuda.synchronize()

    # the very first pass through the path in this process (whether it is a warm-up or a timed step): what `leon -c`,
    # which encodes a file exactly once, sees -- code-object loads and whatever leon_dna_reserve did not size
    cold_ms = [None]

    def timed_step():
        sync()
        t0 = time.perf_counter()
        r = step()
        sync()
        dt = time.perf_counter() - t0
        if cold_ms[0] is None:
            cold_ms[0] = dt * 1e3
        return r, dt

    for _ in range(a.warmup):
        timed_step()
    # the K timed steps, bracketed ONCE on both sides (barrier + synchronize), back to back in between: a step ends with
    # leon_dna_finish, which has every block delivered and the dictionary stream complete, so nothing of it is still in flight.
    # (Bracketing every step cost up to two scheduler ticks per step: torch.cuda.synchronize() on an idle device returns on a
    # 10 ms boundary here -- steps of exactly 820.0 / 830.0 ms around a chain of 800.)
    times, walk_ms, walk_n, dev_ms, chain_ms, emul_ms, stage = [], [], [], [], [], [], None
    sync()
    t_begin = t_prev = time.perf_counter()
    for _ in range(a.steps):
        dict_bytes, n_anchors, acc = step()
        now = time.perf_counter()
        times.append(now - t_prev)
        t_prev = now
        chain_ms.append(acc["ms_chain_busy"])
        walk_ms.append(acc["ms_walk"]); walk_n.append(max(acc["walk_launches"], 1))
        dev_ms.append(acc["ms_total"] - acc["ms_emulated"])      # (a rehearsed seat's own work: not the other ranks' slices it walked in their stead)
        emul_ms.append(acc["ms_emulated"])
        stage = acc
    sync()
    wall_s = time.perf_counter() - t_begin
    if as_rank:                                                  # a seat's wall time without what it walked and looked up in the other ranks' stead --
        # but never less than its dictionary chain, which ran beside that work and which the real rank 0 would wait for just the same
        wall_s = max(wall_s - sum(emul_ms) * 1e-3, sum(chain_ms) * 1e-3, 1e-9)
    if cold_ms[0] is None:                                       # (no warm-up: the first timed step was the process's first)
        cold_ms[0] = times[0] * 1e3
    # max over ranks: whole step, device stages alone (HIP events on each rank's stream), cold first step
    red = torch.tensor([wall_s, float(np.mean(dev_ms)), cold_ms[0]], dtype=torch.float64, device=device)
    if use_dist:
        with WATCH("all_reduce (max over ranks of the timings)"):
            dist.all_reduce(red, op=dist.ReduceOp.MAX)
            torch.cuda.synchronize()
    total_s, device_ms_max, cold_first_step_ms = float(red[0].item()), float(red[1].item()), float(red[2].item())
    ms_per_step = total_s / a.steps * 1e3
    value = n_total * L / 1e6 / (total_s / a.steps)

    extras = not a.quick
    verify = None
    if a.verify or extras:
        import hashlib
        mine = []

        def hsink(user, block_id, p, size, n_reads):
            mine.append((int(block_id), hashlib.sha256(ctypes.string_at(p, size)).hexdigest(), int(n_reads)))
            return 0
        hcb = capi.SINK(hsink)
        encode_stream(hcb)
        dstream, na_v = ctx.finish()
        tables = [mine]
        if use_dist:
            tables = [None] * pg_world
            with WATCH("all_gather_object (block checksums)"):
                dist.all_gather_object(tables, mine)
        if is_root and as_rank:                                      # one seat of an N-rank job: its own blocks only (the union needs the other seats)
            h = hashlib.sha256()
            for bid, digest, nr in sorted(mine):
                h.update(bytes.fromhex(digest) + bid.to_bytes(8, "little") + nr.to_bytes(4, "little"))
            lo_b, hi_b = (min(b[0] for b in mine), max(b[0] for b in mine)) if mine else (None, None)
            verify = {"blocks_sha256_of_this_seat": h.hexdigest(), "n_blocks": len(mine), "first_block": lo_b, "last_block": hi_b,
                      "block_digests": {str(b[0]): b[1][:16] for b in mine} if n_blocks <= 256 else None,     # (small files: block by block, for a union over seats)
                      "dict_sha256": hashlib.sha256(dstream).hexdigest(), "n_anchors": int(na_v),
                      "what": "this seat's blocks only (checksum of block checksums) + the dictionary stream's (empty unless the seat is rank 0)"}
        elif is_root:
            from leon_amd.shard import merge_block_tables
            h = hashlib.sha256()
            for bid, digest, nr in merge_block_tables(tables):       # raises on a gap or a duplicate block
                h.update(bytes.fromhex(digest) + bid.to_bytes(8, "little") + nr.to_bytes(4, "little"))
            verify = {"blocks_sha256": h.hexdigest(), "n_blocks": sum(len(t) for t in tables),
                      "block_digests": {str(b[0]): b[1][:16] for t in tables for b in t} if n_blocks <= 256 else None,
                      "dict_sha256": hashlib.sha256(dstream).hexdigest(), "n_anchors": int(na_v),
                      "blocks_per_rank": [len(t) for t in tables],
                      "what": "checksum of block checksums over the union of all ranks' blocks (gap-free table checked) + the dictionary "
                              "stream's: equal for every world size and batch size"}
        del dstream

    # dominant kernel: k_walk, HIP events recorded on the library's launch stream around each launch
    n_launch = float(np.mean(walk_n))
    walk_avg_ms = float(np.mean(walk_ms)) / n_launch
    alg_bytes = n_local * walk_bytes_per_read(K) / n_launch
    achieved = alg_bytes / (walk_avg_ms * 1e-3) / 1e9
    traffic, traffic_note = None, "no PMC figure for this workload in profiles/walk_traffic.json"
    try:                                                      # PMC-measured HBM bytes per launch (profiles/README.md)
        tj = json.load(open(os.path.join(ROOT, "profiles", "walk_traffic.json")))
        wl = tj["workload"]
        if (wl["reads"], wl["read_len"], wl["kmer_size"], wl["genome"], wl["n_gpus"]) == (n_total, L, K, G, world) and len(batches) == 1:
            if tj.get("k_walk_source") == walk_source_id():
                traffic, traffic_note = tj["traffic_bytes"], "PMC FETCH_SIZE + WRITE_SIZE of this k_walk (source id %s) on this workload" % tj["k_walk_source"]
            else:
                traffic_note = "stale: profiles/walk_traffic.json was measured on another k_walk (%s, this build %s)" % (tj.get("k_walk_source"), walk_source_id())
    except (OSError, KeyError, ValueError):
        pass
    roofline = {"kernel": "k_walk", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": int(traffic) if traffic else None, "traffic_note": traffic_note,
                "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(walk_avg_ms, 3), "launches_per_step": n_launch}

    # What follows is reported beside `value`, never as it, and must never cost the line: an extra that fails is reported as
    # {"error": ...} and the stream is reset for the next one.  (World 1 only: nothing here waits for another rank.)
    def guarded(fn):
        try:
            return fn()
        except Exception as e:                                   # noqa: BLE001 -- whatever it was, the headline numbers stand
            try:
                ctx.reset_stream()
            except Exception:                                    # noqa: BLE001
                pass
            return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    def do_pcie():
        h_reads = reads.cpu().numpy()
        h_off = offsets.cpu().numpy().astype(np.uint64)
        h_batches = [(h_reads[lo:hi].reshape(-1), np.ascontiguousarray(h_off[lo:hi + 1] - h_off[lo])) for lo, hi in batches]   # (views + offsets, made outside the timed region)
        h_times = []
        for _ in range(2):                                       # the first call also pins the library's staging buffers
            ctx.reset_stream()
            payload[0] = payload[1] = 0
            sync()
            t0 = time.perf_counter()
            for hb, ho in h_batches:
                ctx.encode_batch(hb, ho, sink=cb)
            ctx.finish(copy=False)
            sync()
            h_times.append(time.perf_counter() - t0)
        dt = h_times[-1]
        h_st = ctx.stats()
        return {"value": round(n_total * L / 1e6 / dt, 1), "unit": "MB/s", "ms": round(dt * 1e3, 1), "first_call_ms": round(h_times[0] * 1e3, 1),
                "last_batch_device_ms": round(h_st["ms_total"], 1), "last_batch_resolve_ms": round(h_st["ms_resolve"], 1), "chain_busy_ms": round(h_st["ms_chain_busy"], 1),
                "what": "SURVEY 8(d)'s form of the metric: one step through leon_dna_encode_batch, reads and offsets in pageable host "
                        "memory, H2D inside the timed region (three staging threads copy 16 MiB pieces through pinned buffers at PCIe's rate while "
                        "the device packs and resolves what has arrived); second of two calls"}

    def do_decode():
        kept = []
        keep = capi.SINK(lambda user, bid, ptr, size, nreads: (kept.append((int(bid), ctypes.string_at(ptr, size), int(nreads))), 0)[1])
        encode_stream(keep)
        dstream, n_anch = ctx.finish()
        t0 = time.perf_counter()
        anchors = capi.anchor_dict_decode(dstream, n_anch, K)
        t1 = time.perf_counter()
        out_bases, out_lens = ctx.decode_blocks_raw(anchors, kept, [b[2] * L for b in kept])
        t2 = time.perf_counter()
        ref = reads.cpu().numpy().reshape(-1)
        same = bool(np.array_equal(out_bases, ref)) and bool(np.all(out_lens == L))
        del out_bases, out_lens
        # a second call on the same context: the decoder's path cache already holds what the first call learnt from the bloom
        t2b = time.perf_counter()
        out_bases, out_lens = ctx.decode_blocks_raw(anchors, kept, [b[2] * L for b in kept])
        t3 = time.perf_counter()
        same = same and bool(np.array_equal(out_bases, ref)) and bool(np.all(out_lens == L))
        ctx.reset_stream()
        return {"value": round(n_total * L / 1e6 / (t2 - t0), 1), "unit": "MB/s", "dictionary_s": round(t1 - t0, 2),
                "blocks_s": round(t2 - t1, 2), "blocks_MBps": round(n_total * L / 1e6 / (t2 - t1), 1),
                "blocks_s_second_call": round(t3 - t2b, 2), "equals_input": same,
                "what": "leon_host_anchor_dict_decode (one host core) then leon_dna_decode_blocks (one wave per block, path cache "
                        "in HBM), payloads in host memory, bases back in host memory; every base compared with the input"}

    pcie = guarded(do_pcie) if (a.host_input or extras) and world == 1 else None
    decode = guarded(do_decode) if (a.decode or extras) and world == 1 else None
    streams = guarded(lambda: bench_streams(ctx, capi, reads, offsets, n_total, device)) if (a.streams or extras) and world == 1 else None
    e2e = guarded(lambda: end_to_end(min(a.e2e_reads, n_total), device)) if extras and world == 1 and rank == 0 and a.e2e_reads > 0 else None
    cpu = guarded(lambda: cpu_baseline(ctx, reads, a.cpu_sample)) if rank == 0 and world == 1 and a.cpu_sample > 0 else None
    # a file with real-genome structure in genome-position order, and BASELINE's other single-GPU configurations: each with a context of its own
    structured = None
    if extras and world == 1 and not as_rank and a.structured_reads > 0:
        structured = guarded(lambda: structured_case(min(a.structured_reads, n_total), order="sorted", device=device))
    oc = a.other_configs
    if oc is None:
        oc = "10000000:31:150,20000000:63:250" if (n_total, K, L) == (100_000_000, 31, 150) else ""
    other_configs = None
    if extras and world == 1 and not as_rank and oc:
        other_configs = {}
        for spec in oc.split(","):
            n_, k_, l_ = (int(v) for v in spec.split(":"))
            other_configs["%dM_x_%dbp_k%d" % (n_ // 1_000_000, l_, k_) if n_ >= 1_000_000 else "%d_x_%dbp_k%d" % (n_, l_, k_)] = guarded(lambda: other_config(n_, k_, l_, device))

    # what an N-rank curve needs, rank by rank: every rank's device stages and what it coded (gathered; one entry at N = 1)
    mine_rank = {"rank": rank, "blocks": payload[1], "payload_bytes": payload[0], "device_ms": round(float(np.mean(dev_ms)), 2),
                 "chain_ms": round(stage["ms_chain_busy"], 2),
                 # (device memory in use on this rank's GPU after the timed steps: the read set, the context's per-batch buffers, the dictionary, the bloom)
                 "hbm_in_use_GB": round((lambda f, t: (t - f) / 1e9)(*torch.cuda.mem_get_info(device)), 1),
                 "stages_ms": {k: round(v, 2) for k, v in stage.items() if k.startswith("ms_")},
                 "collectives": WATCH.summary() if use_dist else None}
    per_rank = [mine_rank]
    if use_dist:
        per_rank = [None] * pg_world
        with WATCH("all_gather_object (per-rank stages)"):
            dist.all_gather_object(per_rank, mine_rank)

    if is_root:
        out = {
            "metric": "compressed input MB/s (DNA encode path)", "value": round(value, 1), "unit": "MB/s",
            "n_gpus": pg_world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 2),
            **({"as_rank": as_rank, "as_rank_note": "ONE process in the seat of rank %d of a %d-rank job (leon_dna_set_shard): `value` and `ms_per_step` are the wall time "
                                                     "MINUS what this process did in the other ranks' stead (ms_emulated: their slices of the walk, their window look-ups), and at least the seat's dictionary chain (rank 0's) -- what that job would "
                                                     "report if this rank were its slowest, not a measured %d-GPU figure; host threads for the blocks' chains as on a node that grants every rank this box's CPUs (LEON_RC_HOST_THREADS=%s)" % (rank, world, world, os.environ.get("LEON_RC_HOST_THREADS"))} if as_rank else {}),
            "value_hbm_resident": round(value, 1),
            "value_h2d_inclusive": pcie.get("value") if pcie else None,
            "cold_first_step_ms": round(cold_first_step_ms, 2),
            # the multi-GPU truth: the job also waits for the file-wide dictionary stream, one serial
            # chain on a host core of rank 0 whatever N is; the device stages are what shards
            "device_ms_max_over_ranks": round(device_ms_max, 2), "host_chain_ms": round(stage["ms_chain_busy"], 2),
            "value_device_only": round(n_total * L / 1e6 / (device_ms_max * 1e-3), 1),
            "bloom_bcast_ms": round(bcast_ms, 2),
            # the collectives' backend and how many ranks the process group really has (RCCL when "nccl"; null without a process group)
            "collective_backend": backend, "rccl_ranks": dist.get_world_size() if use_dist and backend == "nccl" else None,
            "per_rank": per_rank,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "parity": "bit-identical to oracle/leon_oracle.c in the -m gpu tests; the oracle is a restatement: parity with reference Leon is UNPINNED (gatb-core absent)",
            "config": {"workload": "%d x %d bp synthetic reads, k=%d, genome %d bp (30x), 1%% substitutions, "
                                   "bloom %d bits/k-mer x %d hashes over %d solid k-mers (%s)"
                                   % (n_total, L, K, G, BITS_PER_KMER, N_HASH, n_solid,
                                      "the reads' k-mers of abundance >= %d, device counter" % ABUNDANCE if bloom_from == "count" else "the genome's k-mers"),
                       "reads": n_total, "read_len": L, "kmer_size": K, "reads_per_block": RPB,
                       "batches": len(batches), "batch_reads": B,
                       "sharding": ("bloom broadcast over RCCL; anchor resolution replicated on every rank (file-order dictionary, no exchange); "
                                    + ("walk divided by anchor (rank r walks the r-th slice of the reads sorted by anchor address), its events sent to the "
                                       "rank that codes the read's block in one all-to-all per batch; " if walk_by == "anchor" else "walk on the rank's own block range; ")
                                    + "range coder on contiguous block ranges of every batch; dictionary stream on rank 0") if world > 1 else "single GPU",
                       "walk_by": walk_by,
                       "bloom_bytes": nbytes, "bloom_bcast_ms": round(bcast_ms, 2), "bloom_build_s": round(bloom_s, 2),
                       "kmer_count_s": round(count_s, 2), "solid_kmers": n_solid},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "pcie_inclusive": pcie,
            "decode": decode,
            "verify": verify,
            "end_to_end": e2e,
            "streams": streams,
            "structured": structured,
            "other_configs": other_configs,
            "collectives": WATCH.summary() if use_dist else None,
            "stages_ms_rank0": {k: round(v, 2) for k, v in stage.items() if k.startswith("ms_")},
            # every timed step on rank 0 (ms_per_step is their mean, max over ranks): the whole step, and the dictionary chain inside it
            "step_ms_rank0": [round(t * 1e3, 1) for t in times], "chain_ms_rank0": [round(c, 1) for c in chain_ms],
            # the last step's host view: leon_dna_reset_stream, the encode calls (they return when the blocks are delivered), leon_dna_finish (waits for the chain)
            "step_parts_ms_rank0": {k: round(stage[k], 1) for k in ("host_reset_ms", "host_encode_calls_ms", "host_finish_ms")},
            # `value` is the host chain's (one core of rank 0): which CPU that was, and what a dictionary symbol cost on it
            "host": host_info(stage["ms_chain_busy"], n_anchors * K),
            "rank0": {"anchors": n_anchors, "payload_bytes": payload[0] + dict_bytes, "blocks": payload[1],
                      "symbols": int(stage["n_symbols"]), "resolve_rounds": int(stage["resolve_rounds"]), "resolve_chain_reads": int(stage["resolve_chain_reads"]),
                      "bits_per_base": round(8.0 * (payload[0] + dict_bytes) / (max(n_local, 1) * L), 4)},
        }
        emit_line(json.dumps(out))
    if use_dist:
        with WATCH("barrier (end of the run)"):
            dist.barrier()
        dist.destroy_process_group()


def make_exchange(dist, device, backend, pg_rank, pg_world, capi, WATCH):
    """the all-to-all of leon_dna_set_exchange: 64-bit words in device memory, send_counts[d] of them for rank d, over the process
    group -- RCCL's all_to_all_single between GPUs (the words never touch the host); over gloo (rehearsals on one device) every rank
    gathers everybody's words through host memory and keeps its own pieces"""
    keep = {}

    def fn(d_send, counts):
        n_send = sum(counts)
        send = torch.empty(max(n_send, 1), dtype=torch.int64, device=device)
        if n_send:
            capi.device_copy(send.data_ptr(), d_send, n_send * 8, device_id=device.index or 0)
        mine = torch.tensor(counts, dtype=torch.int64)
        if backend == "nccl":
            mat = torch.empty(pg_world * pg_world, dtype=torch.int64, device=device)
            with WATCH("all_gather_into_tensor (walk exchange: the counts)"):
                dist.all_gather_into_tensor(mat, mine.to(device))
                mat = mat.cpu().view(pg_world, pg_world)             # mat[src][dst]
            recv_counts = [int(mat[src][pg_rank]) for src in range(pg_world)]
            recv = torch.empty(max(sum(recv_counts), 1), dtype=torch.int64, device=device)
            with WATCH("all_to_all_single (walk exchange: the event words) [%d sent, %d received]" % (n_send, sum(recv_counts))):
                dist.all_to_all_single(recv[:sum(recv_counts)], send[:n_send], output_split_sizes=recv_counts, input_split_sizes=list(counts))
                torch.cuda.synchronize()
        else:
            rows = [torch.empty(pg_world, dtype=torch.int64) for _ in range(pg_world)]
            with WATCH("all_gather (walk exchange: the counts)"):
                dist.all_gather(rows, mine)
            width = max(int(r.sum()) for r in rows)
            padded = torch.zeros(max(width, 1), dtype=torch.int64)
            padded[:n_send] = send[:n_send].cpu()
            bufs = [torch.empty_like(padded) for _ in range(pg_world)]
            with WATCH("all_gather (walk exchange: the event words, through host memory)"):
                dist.all_gather(bufs, padded)
            pieces = []
            for src in range(pg_world):
                at = int(rows[src][:pg_rank].sum())
                pieces.append(bufs[src][at:at + int(rows[src][pg_rank])])
            got = torch.cat(pieces) if pieces else torch.empty(0, dtype=torch.int64)
            recv = got.to(device) if got.numel() else torch.empty(1, dtype=torch.int64, device=device)
            recv_counts = [int(rows[src][pg_rank]) for src in range(pg_world)]
            torch.cuda.synchronize()
        keep["recv"] = recv                                          # stays alive until the next exchange
        return recv.data_ptr(), sum(recv_counts)
    return fn


def make_gather(dist, device, backend, pg_rank, pg_world, capi, WATCH):
    """the all-gather of leon_dna_set_gather: every rank's part of a device buffer (part r at d_buf + r * part_bytes) to every rank --
    RCCL's all_gather_into_tensor between GPUs; over gloo (rehearsals on one device) through host memory"""
    def fn(d_buf, part_bytes, world):
        assert world == pg_world
        words = part_bytes // 8
        mine = torch.empty(max(words, 1), dtype=torch.int64, device=device)
        if words:
            capi.device_copy(mine.data_ptr(), d_buf + pg_rank * part_bytes, part_bytes, device_id=device.index or 0)
        if backend == "nccl":
            everybody = torch.empty(max(words, 1) * pg_world, dtype=torch.int64, device=device)
            with WATCH("all_gather_into_tensor (a window's look-ups)"):
                dist.all_gather_into_tensor(everybody, mine)
                torch.cuda.synchronize()
        else:
            parts = [torch.empty(max(words, 1), dtype=torch.int64) for _ in range(pg_world)]
            with WATCH("all_gather (a window's look-ups, through host memory)"):
                dist.all_gather(parts, mine.cpu())
            everybody = torch.cat(parts).to(device)
            torch.cuda.synchronize()
        if words:
            capi.device_copy(d_buf, everybody.data_ptr(), part_bytes * pg_world, device_id=device.index or 0)    # (complete on return)
    return fn


def host_info(chain_ms, chain_symbols):
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    q = _cpu_quota()
    return {"cpu": model, "cpus_allowed": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": round(q, 2) if q else None,
            "cpus_usable": len(os.sched_getaffinity(0)) if not q else max(1, min(len(os.sched_getaffinity(0)), int(q + 0.5))),
            "chain_ns_per_symbol": round(chain_ms * 1e6 / chain_symbols, 3) if chain_symbols else None}


def bench_streams(ctx, capi, reads, offsets, n_total, device):
    """--streams: the kernels of the streams either side of the DNA stream on device-resident synthetic data"""
    streams = {}
    # lossy qualities: DnaEncoder::smoothQuals over the workload's reads against the file's bloom (qualities: a fixed ramp)
    quals = torch.full((n_total * L,), 70, dtype=torch.uint8, device=device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = ctx.lib.leon_qual_smooth_batch_device(ctx.h, ctypes.c_void_p(reads.data_ptr()), ctypes.c_void_p(offsets.data_ptr()), n_total,
                                               ctypes.c_void_p(quals.data_ptr()))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert rc == 0
    smoothed = float((quals == 64).float().mean().item())
    streams["qual_smooth"] = {"ms": round(dt * 1e3, 1), "MBps": round(n_total * L / 1e6 / dt, 1), "fraction_smoothed": round(smoothed, 4),
                              "what": "leon_qual_smooth_batch_device over the workload's reads: pack + one wave per read, L-k+1 bloom look-ups per read"}
    del quals
    # lossless qualities: the read blocks' zlib streams written by the device (runs + dynamic Huffman codes per 32 KB), qualities of
    # the end-to-end FASTQ's kind resident in HBM, the streams back in host memory; beside it zlib's default strategy on a sample
    import zlib
    nq = min(10_000_000, n_total)
    g = torch.Generator(device=device); g.manual_seed(11)
    qalpha = torch.tensor(list(b"#5:?ABCDEFGHIJ"), dtype=torch.uint8, device=device)
    qd = qalpha[torch.minimum(torch.randint(0, 14, (nq, L), device=device, generator=g), torch.randint(4, 14, (nq, 1), device=device, generator=g))].contiguous()
    qoff = (np.arange(nq + 1, dtype=np.uint64) * L)
    torch.cuda.synchronize()
    got = {}
    qsink = capi.SINK(lambda user, bid, ptr, size, nreads: (got.__setitem__(int(bid), ctypes.string_at(ptr, size) if int(bid) in (0, 7) else size), 0)[1])
    best = None
    for _ in range(2):
        got.clear()
        t0 = time.perf_counter()
        rc = ctx.lib.leon_qual_deflate_blocks_device(device.index or 0, ctypes.c_void_p(qd.data_ptr()), capi._ptr(qoff, capi._u64p), nq, RPB, qsink, None, 0)
        dt = time.perf_counter() - t0
        assert rc == 0, rc
        best = dt if best is None else min(best, dt)
    out_bytes = sum(len(v) if isinstance(v, bytes) else v for v in got.values())
    h = qd[:8 * RPB].cpu().numpy()
    text = lambda b: b"".join(r.tobytes() + b"\n" for r in h[b * RPB:(b + 1) * RPB])
    inflates = all(zlib.decompress(got[b]) == text(b) for b in (0, 7) if b in got and (b + 1) * RPB <= nq)
    t0 = time.perf_counter()
    zl = len(zlib.compress(text(0)))
    z_dt = time.perf_counter() - t0
    streams["qual_deflate"] = {"reads": nq, "bytes_in": nq * (L + 1), "bytes_out": out_bytes, "ms": round(best * 1e3, 1), "MBps": round(nq * (L + 1) / 1e6 / best, 1),
                               "ratio": round(out_bytes / (nq * (L + 1)), 4), "inflates_to_input": bool(inflates),
                               "zlib_default_one_core": {"MBps": round(len(text(0)) / 1e6 / z_dt, 1), "ratio": round(zl / len(text(0)), 4)},
                               "what": "leon_qual_deflate_blocks_device: qualities resident in HBM -> one zlib stream per read block in host memory (text with "
                                       "newlines, one workgroup per 32 KB deflate block, gather, D2H), best of 2; sampled blocks through Python's zlib.decompress"}
    del qd
    # header stream: 10 M SRA-style headers resident in HBM -> records (one lane per header) -> k_rc_encode
    nh = min(10_000_000, n_total)
    blob, hoff = sra_headers(nh, seed=7)
    d_blob = torch.from_numpy(blob.copy()).to(device); d_hoff = torch.from_numpy(hoff).to(device)
    first = blob[:int(hoff[1])].tobytes()
    hbytes = [0]
    hsink = capi.SINK(lambda user, bid, ptr, size, nreads: (hbytes.__setitem__(0, hbytes[0] + size), 0)[1])
    best = None
    for _ in range(3):
        ctx.reset_stream()
        hbytes[0] = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = ctx.lib.leon_header_encode_batch_device(ctx.h, ctypes.c_void_p(d_blob.data_ptr()), ctypes.c_void_p(d_hoff.data_ptr()), nh, 0,
                                                     first, len(first), hsink, None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert rc == 0
        best = dt if best is None else min(best, dt)
    streams["header"] = {"headers": nh, "bytes_in": int(hoff[-1]), "bytes_out": hbytes[0], "ms": round(best * 1e3, 1),
                         "MBps": round(int(hoff[-1]) / 1e6 / best, 1),
                         "what": "leon_header_encode_batch_device, headers resident in HBM, best of 3: k_hdr_symbols x2 + scan + k_rc_encode + D2H"}
    # ... and back: the blocks of the last run through both decoders (C calls only: payloads in, text + offsets out)
    hblocks = []
    keep_h = capi.SINK(lambda user, bid, ptr, size, nreads: (hblocks.append((int(bid), ctypes.string_at(ptr, size), int(nreads))), 0)[1])
    ctx.reset_stream()
    rc = ctx.lib.leon_header_encode_batch_device(ctx.h, ctypes.c_void_p(d_blob.data_ptr()), ctypes.c_void_p(d_hoff.data_ptr()), nh, 0, first, len(first), keep_h, None)
    assert rc == 0
    pay, poff, pnr = capi._join_blocks(hblocks)
    out_off = np.zeros(nh + 1, dtype=np.uint64); need = ctypes.c_uint64(); cap = int(hoff[-1]) + 64
    texts = {}
    for name, call in (("host_threads", lambda o: ctx.lib.leon_host_header_decode_blocks(capi._ptr(pay, capi._u8p), capi._ptr(poff, capi._u64p), capi._ptr(pnr, capi._u32p),
                            len(hblocks), first, len(first), capi._ptr(o, capi._u8p), cap, capi._ptr(out_off, capi._u64p), ctypes.byref(need), 0)),
                       ("device_symbols", lambda o: ctx.lib.leon_header_decode_blocks(ctx.h, capi._ptr(pay, capi._u8p), capi._ptr(poff, capi._u64p), capi._ptr(pnr, capi._u32p),
                            len(hblocks), first, len(first), capi._ptr(o, capi._u8p), cap, capi._ptr(out_off, capi._u64p), ctypes.byref(need), 0))):
        best = None
        for _ in range(2):
            o = np.empty(cap, dtype=np.uint8)
            t0 = time.perf_counter(); rc = call(o); dt = time.perf_counter() - t0
            assert rc == 0, rc
            best = dt if best is None else min(best, dt)
        texts[name] = (round(best * 1e3, 1), bool(np.array_equal(o[:int(hoff[-1])], blob)))
    streams["header_decode"] = {"headers": nh, "blocks": len(hblocks), "host_threads_ms": texts["host_threads"][0], "device_symbols_ms": texts["device_symbols"][0],
                                "equal_input": texts["host_threads"][1] and texts["device_symbols"][1],
                                "what": "leon_host_header_decode_blocks (all the CPUs of the quota) against leon_header_decode_blocks (symbols on the device, one wave per "
                                        "block; text on the host threads), payloads in and text out in host memory, best of 2"}
    del d_blob, d_hoff, hblocks, pay
    ctx.reset_stream()

    return streams


def _digits(v, d):
    """the d decimal digits of every entry of v as ASCII, most significant first (division by the constant 10, column by column:
    numpy's broadcast int64 division by an array of powers is two orders of magnitude slower)"""
    out = np.empty((len(v), d), dtype=np.uint8)
    v = np.asarray(v, dtype=np.int64).copy()
    for j in range(d - 1, -1, -1):
        q = v // 10
        out[:, j] = (v - q * 10 + 48).astype(np.uint8)
        v = q
    return out


def sra_headers(n, seed=7, L_field=150):
    """n SRA-style header texts (`SRR387476.<i> HWI-ST1234:3:1101:<x>:<y> length=150`) back to back + their offsets, built as
    fixed-width matrices per digit count of the index (numpy's char routines take a minute for 10 M of them)"""
    rng = np.random.default_rng(seed)
    x, y = rng.integers(10000, 100000, n), rng.integers(100000, 1000000, n)
    idx = np.arange(1, n + 1, dtype=np.int64)

    def digits(v, d):
        return _digits(v, d)
    parts, lens = [], np.empty(n, dtype=np.int64)
    lo = 0
    while lo < n:
        d = len(str(int(idx[lo])))
        hi = min(n, lo + max(1, 10 ** d - int(idx[lo])))
        cols = [b"SRR387476.", digits(idx[lo:hi], d), b" HWI-ST1234:3:1101:", digits(x[lo:hi], 5), b":", digits(y[lo:hi], 6), b" length=%d" % L_field]
        w = sum(len(c) if isinstance(c, bytes) else c.shape[1] for c in cols)
        rec = np.empty((hi - lo, w), dtype=np.uint8)
        c0 = 0
        for c in cols:
            if isinstance(c, bytes):
                rec[:, c0:c0 + len(c)] = np.frombuffer(c, dtype=np.uint8)[None, :]; c0 += len(c)
            else:
                rec[:, c0:c0 + c.shape[1]] = c; c0 += c.shape[1]
        parts.append(rec.reshape(-1)); lens[lo:hi] = w
        lo = hi
    off = np.zeros(n + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    return np.concatenate(parts), off


def write_fastq(path, n, Lr, device):
    """a synthetic FASTQ of n reads of the bench's generator (genome n*Lr/30, 1 % substitutions) with SRA-style headers and
    structured qualities, built as fixed-width record matrices (reads grouped by the digit count of their index)"""
    genome = gen_genome(max(n * Lr // 30, 10 * Lr), device)
    qalpha = np.frombuffer(b"#5:?ABCDEFGHIJ", dtype=np.uint8)

    def digits(v, d):
        return _digits(v, d)
    with open(path, "wb") as f:
        for c0 in range((n + CHUNK - 1) // CHUNK):
            m = min(CHUNK, n - c0 * CHUNK)
            rd = gen_reads_chunk(genome, c0, CHUNK, 0.01, device, L=Lr)[:m].cpu().numpy()
            rng = np.random.default_rng(c0)
            x, y = rng.integers(10000, 100000, m), rng.integers(100000, 1000000, m)
            q = qalpha[np.minimum(rng.integers(0, 14, (m, Lr)), rng.integers(4, 14, (m, 1)))]
            idx = np.arange(c0 * CHUNK + 1, c0 * CHUNK + m + 1, dtype=np.int64)
            lo = 0
            while lo < m:
                d = len(str(int(idx[lo])))
                hi = min(m, lo + max(1, 10 ** d - int(idx[lo])))
                parts = [b"@SRR387476.", digits(idx[lo:hi], d), b" HWI-ST1234:3:1101:", digits(x[lo:hi], 5), b":", digits(y[lo:hi], 6),
                         b" length=%d\n" % Lr, rd[lo:hi], b"\n+\n", q[lo:hi], b"\n"]
                w = sum(len(p) if isinstance(p, bytes) else p.shape[1] for p in parts)
                rec = np.empty((hi - lo, w), dtype=np.uint8)
                c = 0
                for p in parts:
                    if isinstance(p, bytes):
                        rec[:, c:c + len(p)] = np.frombuffer(p, dtype=np.uint8)[None, :]
                        c += len(p)
                    else:
                        rec[:, c:c + p.shape[1]] = p
                        c += p.shape[1]
                f.write(rec.tobytes())
                lo = hi
    del genome
    torch.cuda.empty_cache()


def end_to_end(n, device):
    """the reference's own acceptance test through the C++ host mirror (/root/reference/scripts/simple_test.sh:51,54,62):
    `leon -c -lossless` then `leon -d -test-file` on a synthetic FASTQ in /dev/shm; wall time of each command"""
    import shutil
    import subprocess
    import tempfile
    leon = os.path.join(ROOT, "leon_amd", "lib", "leon")
    if not os.path.exists(leon):
        return {"error": "leon_amd/lib/leon is not built"}
    need = 3 * n * (2 * L + 70)                                # the FASTQ, its .leon and the restored copy
    base = next((d for d in ("/dev/shm", tempfile.gettempdir()) if os.path.isdir(d) and os.access(d, os.W_OK) and shutil.disk_usage(d).free > need), None)
    if base is None:
        return {"error": "no directory with %d MB free for the FASTQ" % (need >> 20)}
    work = tempfile.mkdtemp(prefix="leon_e2e_", dir=base)
    fq = os.path.join(work, "reads.fastq")
    out = {"reads": n, "read_len": L}
    try:
        t0 = time.time()
        write_fastq(fq, n, L, device)
        out["generate_s"] = round(time.time() - t0, 1)
        out["fastq_bytes"] = os.path.getsize(fq)

        def run(name, *args):
            t = time.time()
            r = subprocess.run([leon] + list(args), capture_output=True, text=True)
            out[name + "_s"] = round(time.time() - t, 2)
            out[name + "_rc"] = r.returncode
            if r.returncode:
                out[name + "_stderr"] = r.stderr[-400:]
            return r
        # the same file with the quality blocks written by the device's deflate (`-qual-deflate device`: inflates to the same text, not
        # zlib's bytes), then -- what is kept and decoded back -- by zlib itself, the default and the reference's bytes
        run("compress_lossless_qual_deflate_device", "-file", fq, "-c", "-lossless", "-qual-deflate", "device")
        out["leon_bytes_qual_deflate_device"] = os.path.getsize(fq + ".leon") if os.path.exists(fq + ".leon") else None
        run("compress_lossless", "-file", fq, "-c", "-lossless")
        out["leon_bytes"] = os.path.getsize(fq + ".leon") if os.path.exists(fq + ".leon") else None
        r = run("decompress_test_file", "-file", fq + ".leon", "-d", "-test-file")
        out["identical"] = r.returncode == 0 and "is identical to" in r.stdout
        if out.get("compress_lossless_rc") == 0:
            out["compress_MBps_of_file"] = round(out["fastq_bytes"] / 1e6 / out["compress_lossless_s"], 1)
        if out.get("compress_lossless_qual_deflate_device_rc") == 0:
            out["compress_MBps_of_file_qual_deflate_device"] = round(out["fastq_bytes"] / 1e6 / out["compress_lossless_qual_deflate_device_s"], 1)
        if out["identical"]:
            out["decompress_MBps_of_file"] = round(out["fastq_bytes"] / 1e6 / out["decompress_test_file_s"], 1)
        out["what"] = ("leon -file X.fastq -c -lossless (quality blocks: zlib's own bytes, the default; and once with -qual-deflate device), then "
                       "leon -file X.fastq.leon -d -test-file (byte comparison with the original), whole commands timed from outside: parse, k-mer "
                       "counting, bloom, the three streams, HDF5 container")
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return out


def _cpu_topology():
    """(CPUs this process may run on, physical cores among them): siblings of one core share `thread_siblings_list`"""
    cpus = sorted(os.sched_getaffinity(0))
    cores = set()
    for c in cpus:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                cores.add(f.read().strip())
        except OSError:
            cores.add(str(c))
    return cpus, len(cores)


def _cpu_quota():
    """the cgroup's CPU quota in CPUs (None = none): a box may ALLOW 256 CPUs (affinity) and grant 16 CPUs' worth of time"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / per if q > 0 and per > 0 else None
    except (OSError, ValueError):
        return None


def cpu_baseline(ctx, reads, per_worker):
    """The CPU restatement (oracle/leon_oracle.c, kind = "port") on EVERY CPU this process may run on, two figures:
    `value`: one worker thread per allowed CPU (the C code runs outside the GIL), each coding its own slice of the workload's first reads as
             an independent stream against the file's bloom -- how Leon's threads would fare on as many files: small dictionaries, nothing
             shared but the read-only bloom; an upper bound on the port, not a measurement of Leon.  value = all bases / slowest worker;
    `one_stream`: ONE thread, one shared stream (`-nb-cores 1`, the only order byte parity is defined against)."""
    import concurrent.futures as cf
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cpus, n_phys = _cpu_topology()
    quota = _cpu_quota()
    # one worker per CPU this process can actually USE: the allowed CPUs, or the cgroup's quota when that is smaller (measured on a box that
    # allows 256 hardware threads and grants 16 CPUs: 256 workers finish 64 M reads in 205 s -- 47 MB/s, the quota's rate -- and the run
    # takes five minutes; 16 workers say the same in 12 s)
    workers = len(cpus) if not quota else max(1, min(len(cpus), int(quota + 0.5)))
    n_avail = reads.shape[0]
    per = max(1000, min(per_worker, n_avail // workers))
    per = per // RPB * RPB if per >= RPB else per
    bl = O.Bloom(ctx.bloom_tai, K, N_HASH, 12)
    bl.set_bits(ctx.bloom_download())
    h_reads = reads[:per * workers].contiguous().cpu().numpy()
    off = np.arange(per + 1, dtype=np.uint64) * L

    def work(i):
        bases = h_reads[i * per:(i + 1) * per].tobytes()
        t0 = time.perf_counter()
        O.encode(bases, off, K, RPB, bl, trace=False)
        return time.perf_counter() - t0

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(workers) as pool:
        secs = list(pool.map(work, range(workers)))
    wall = time.perf_counter() - t0
    slowest = max(secs)
    # one shared stream on one core: as many reads as ~15 s of one worker's rate take
    rate1 = per * L / (sum(secs) / len(secs))
    one_n = int(min(n_avail, max(per, 15.0 * rate1 / L)))
    one_n = max(RPB, one_n // RPB * RPB) if one_n >= RPB else one_n
    one_bases = reads[:one_n].contiguous().cpu().numpy().tobytes()
    t0 = time.perf_counter()
    O.encode(one_bases, np.arange(one_n + 1, dtype=np.uint64) * L, K, RPB, bl, trace=False)
    one_s = time.perf_counter() - t0
    return {"value": round(per * workers * L / 1e6 / slowest, 2), "unit": "MB/s", "cores": workers,
            "cpus_allowed": len(cpus), "physical_cores_allowed": n_phys, "cgroup_cpu_quota": round(quota, 2) if quota else None,
            "smt": "%d hardware threads allowed on %d physical cores" % (len(cpus), n_phys) if n_phys != len(cpus) else "one thread per physical core",
            "cores_note": "cores = the CPUs this process can use at once: min(allowed CPUs, cgroup CPU quota)",
            "kind": "port",
            "sample": "first %d reads of the workload, %d reads per worker thread (an independent stream each, the file's bloom shared read-only), "
                      "slowest worker %.1f s, mean %.1f s (%.1f s with start-up); oracle/leon_oracle.c; reference Leon itself cannot be "
                      "built here (gatb-core absent)" % (per * workers, per, slowest, sum(secs) / len(secs), wall),
            "one_stream": {"value": round(one_n * L / 1e6 / one_s, 2), "unit": "MB/s", "cores": 1, "kind": "port",
                           "sample": "first %d reads of the workload as ONE stream on one core (-nb-cores 1: file order, one dictionary), %.1f s" % (one_n, one_s)}}


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:              # noqa: BLE001 -- a rank that fails must END, non-zero and at once: the launcher then tears the job down
        import traceback                    # instead of the other ranks waiting in their next collective for this one
        traceback.print_exc()
        mem = None
        try:
            f_, t_ = torch.cuda.mem_get_info()
            mem = round((t_ - f_) / 1e9, 1)
        except Exception:                   # noqa: BLE001
            pass
        print(json.dumps({"bench_error": "%s: %s" % (type(e).__name__, str(e)[:400]), "rank": int(os.environ.get("RANK", "0")), "hbm_in_use_GB": mem}), file=sys.stderr)
        sys.stderr.flush(); sys.stdout.flush()
        os._exit(1)

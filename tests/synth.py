"""Seeded synthetic read sets modelled on data/toy.fasta's generator fields (er0.01, indel0, rev0/1)."""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b


def make_genome(length, seed=42):
    rng = np.random.default_rng(seed)
    return _ACGT[rng.integers(0, 4, size=length)]


def make_reads(genome, n_reads, read_len, seed=43, err=0.01, n_rate=0.0, ragged=False):
    """returns (bases uint8[total], offsets uint64[n+1]); reads from both strands with substitutions."""
    rng = np.random.default_rng(seed)
    G = len(genome)
    if ragged:
        lens = rng.integers(max(1, read_len // 4), read_len + 1, size=n_reads)
    else:
        lens = np.full(n_reads, read_len, dtype=np.int64)
    lens = np.minimum(lens, G)
    off = np.zeros(n_reads + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    total = int(off[-1])
    starts = (rng.random(n_reads) * (G - lens + 1)).astype(np.int64)
    rid = np.repeat(np.arange(n_reads), lens)
    within = np.arange(total) - np.repeat(off[:-1].astype(np.int64), lens)
    strand = rng.integers(0, 2, size=n_reads).astype(bool)
    fwd_idx = starts[rid] + within
    rev_idx = starts[rid] + (lens[rid] - 1 - within)
    idx = np.where(strand[rid], rev_idx, fwd_idx)
    b = genome[idx]
    b = np.where(strand[rid], _COMP[b], b)
    if err > 0:
        m = rng.random(total) < err
        sub = _ACGT[rng.integers(0, 4, size=total)]
        sub = np.where(sub == b, _ACGT[(np.searchsorted(_ACGT, sub) + 1) % 4], sub)
        b = np.where(m, sub, b)
    if n_rate > 0:
        m = rng.random(total) < n_rate
        b = np.where(m, np.uint8(ord("N")), b)
    return np.ascontiguousarray(b, dtype=np.uint8), off


def read_fasta(path):
    reads = []
    with open(path) as f:
        for line in f:
            if not line.startswith(">"):
                reads.append(line.strip())
    return reads


# ---- inputs with the structure of real genomes and real files ----------------------------------------
# The sets above place reads uniformly at random over an i.i.d. genome, in random order: file-order dependency chains of the
# anchor dictionary are 3-6 reads long there.  Real files are not like that: reads sorted by position (samtools sort | fastq,
# amplicon tilings, many simulators) make EVERY read depend on the one before it; repeats pile many reads on one dictionary key;
# PCR duplicates propose the same anchor again and again.

def make_structured_genome(length, seed=42, dispersed=4, tandem=4):
    """an i.i.d. genome with `dispersed` families of copied segments (0.1-5 kbp, 2-6 copies each, some reverse-complemented)
    and `tandem` microsatellites (a 1-6 bp unit repeated over 40-400 bp) written into it"""
    rng = np.random.default_rng(seed)
    g = _ACGT[rng.integers(0, 4, size=length)].copy()
    for _ in range(dispersed):
        seg = int(min(max(100, length // 8), rng.integers(100, 5001)))
        if seg >= length:
            continue
        src = int(rng.integers(0, length - seg))
        piece = g[src:src + seg].copy()
        for _ in range(int(rng.integers(2, 7))):
            dst = int(rng.integers(0, length - seg))
            g[dst:dst + seg] = _COMP[piece[::-1]] if rng.random() < 0.5 else piece
    for _ in range(tandem):
        unit = _ACGT[rng.integers(0, 4, size=int(rng.integers(1, 7)))]
        span = int(min(max(8, length // 10), rng.integers(40, 401)))
        dst = int(rng.integers(0, max(1, length - span)))
        g[dst:dst + span] = np.tile(unit, span // len(unit) + 1)[:span]
    return g


ORDERS = ("random", "sorted", "pairs", "sorted-strands")


def make_structured_reads(genome, n_reads, read_len, seed=43, order="sorted", err=0.01, n_rate=0.0, ragged=False,
                          dup_rate=0.0, skew=0.0, stride=None):
    """returns (bases uint8[total], offsets uint64[n+1]).
    order: "random" | "sorted" (by start position: what a position-sorted BAM gives back) | "sorted-strands" (sorted, all reads
           forward: tiled amplicons) | "pairs" (mates interleaved: forward read at p, reverse read ~2.2 read lengths downstream).
    dup_rate: that share of the reads are copies of another read's placement (PCR duplicates; errors drawn independently).
    skew: 0 = uniform coverage; > 0 concentrates the starts (a share `skew` of the reads falls into a tenth of the genome).
    stride: reads start every `stride` bases instead of at random places (with order="sorted": a tiling)."""
    assert order in ORDERS
    rng = np.random.default_rng(seed)
    G = len(genome)
    lens = rng.integers(max(1, read_len // 4), read_len + 1, size=n_reads) if ragged else np.full(n_reads, read_len, dtype=np.int64)
    lens = np.minimum(lens, G).astype(np.int64)
    span = np.maximum(G - lens + 1, 1)
    if stride is not None:
        starts = (np.arange(n_reads, dtype=np.int64) * stride) % span
    else:
        u = rng.random(n_reads)
        if skew > 0:
            hot = rng.random(n_reads) < skew
            lo = rng.random() * 0.9
            u = np.where(hot, lo + 0.1 * u, u)
        starts = (u * span).astype(np.int64)
    strand = rng.integers(0, 2, size=n_reads).astype(bool)
    if order == "sorted-strands":
        strand[:] = False
    if order == "pairs":
        half = n_reads // 2
        gap = int(2.2 * read_len)
        starts[1:2 * half:2] = np.minimum(starts[0:2 * half:2] + gap, span[1:2 * half:2] - 1)
        strand[0:2 * half:2] = False
        strand[1:2 * half:2] = True
    if dup_rate > 0:
        dup = np.nonzero(rng.random(n_reads) < dup_rate)[0]
        src = rng.integers(0, n_reads, size=len(dup))
        starts[dup] = np.minimum(starts[src], span[dup] - 1)
        strand[dup] = strand[src]
    if order in ("sorted", "sorted-strands"):
        o = np.argsort(starts, kind="stable")
        starts, strand, lens = starts[o], strand[o], lens[o]
    off = np.zeros(n_reads + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    total = int(off[-1])
    rid = np.repeat(np.arange(n_reads), lens)
    within = np.arange(total) - np.repeat(off[:-1].astype(np.int64), lens)
    idx = np.where(strand[rid], starts[rid] + (lens[rid] - 1 - within), starts[rid] + within)
    b = genome[idx]
    b = np.where(strand[rid], _COMP[b], b)
    if err > 0:
        m = rng.random(total) < err
        sub = _ACGT[rng.integers(0, 4, size=total)]
        sub = np.where(sub == b, _ACGT[(np.searchsorted(_ACGT, sub) + 1) % 4], sub)
        b = np.where(m, sub, b)
    if n_rate > 0:
        b = np.where(rng.random(total) < n_rate, np.uint8(ord("N")), b)
    return np.ascontiguousarray(b, dtype=np.uint8), off

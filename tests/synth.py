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

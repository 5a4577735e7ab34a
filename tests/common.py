"""Shared helpers for the parity tests: build a read set + solid k-mers + oracle bloom."""
import os

import numpy as np

import oracle_lib as O
import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NB_BITS_PER_KMER = 12


def toy_reads():
    reads = synth.read_fasta(os.path.join(GOLDEN, "toy.fasta"))
    bases, off = O.reads_to_arrays(reads)
    return bases, off


def synthetic(n_reads, read_len, genome_len, seed=1, junk_reads=0, **kw):
    """junk_reads: that many reads, spread over the set, are random bases (no solid k-mer: reads without an anchor)"""
    g = synth.make_genome(genome_len, seed=seed)
    b, off = synth.make_reads(g, n_reads, read_len, seed=seed + 1, **kw)
    if junk_reads:
        rng = np.random.default_rng(seed + 2)
        b = b.copy()
        for r in rng.choice(n_reads, size=junk_reads, replace=False):
            s, e = int(off[r]), int(off[r + 1])
            b[s:e] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=e - s)]
    return b.tobytes(), off


def make_bloom(bases, off, k, min_abundance=3, n_hash=7, block_nbits=12):
    solid = O.count_solid(bases, off, k, min_abundance)
    tai = max(len(solid) // O.kwords(k) * NB_BITS_PER_KMER, 1000)
    bl = O.Bloom(tai, k, n_hash, block_nbits)
    bl.insert(solid)
    return bl, solid, tai

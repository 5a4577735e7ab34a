"""-m gpu: inputs with the structure of real files -- reads in genome-position order, tilings, repeats, PCR duplicates, coverage skew,
interleaved mates -- through the HIP path against the oracle, stage by stage (anchors -> events -> block bytes -> dictionary stream).

Why these have a file of their own: the anchor dictionary has file-order semantics (Leon::findAndInsertAnchor under its mutex,
DnaEncoder::findExistingAnchor [RECALLED]; SURVEY 7.3 hard part 1), and the device resolves a window of reads by a fixpoint whose
round count is the longest chain of reads each waiting for the one before it.  Uniformly placed reads in random order (every other
test's input) give chains of 3-6; position-sorted reads give ONE chain through the whole window.  Until round 5 the fixpoint failed
such a batch after 63 rounds (LEON_E_STATE); now the rounds hand what they leave to an exact sequential pass (k_chain_*).
The reference takes reads in any order (/root/reference/README.md:36-47)."""
import numpy as np
import pytest

import common
import oracle_lib as O
import synth
from test_gpu_parity import _full_compare

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("rc_chains")]


def _reads(n, L, G, order, seed=1, genome=None, **kw):
    g = synth.make_structured_genome(G, seed=seed) if genome is None else genome
    b, off = synth.make_structured_reads(g, n, L, seed=seed + 1, order=order, **kw)
    return b.tobytes(), off


@pytest.mark.parametrize("window", [0, 64])
def test_position_sorted_reads(window):
    # 20 000 x 150 bp over 100 kbp (30 x), sorted by start: every read contains the k-mer the read before it proposes
    bases, off = _reads(20000, 150, 100000, "sorted", seed=7)
    ref, st = _full_compare(bases, off, 31, 1000, window=window)
    assert st["resolve_chain_reads"] > 0                     # the rounds alone do not settle this input


def test_tiled_reads_one_strand():
    # an amplicon-style tiling: a read every 5 bases, all forward, no errors -- the longest chain there is
    bases, off = _reads(50000, 150, 250150, "sorted-strands", seed=8, stride=5, err=0.0)
    _full_compare(bases, off, 31, 5000)
    _full_compare(bases, off, 31, 5000, window=4096, batches=3)


def test_tandem_repeat_genome_and_duplicates():
    g = synth.make_structured_genome(60000, seed=9, dispersed=12, tandem=40)
    for order in ("sorted", "random", "pairs"):
        bases, off = _reads(12000, 150, 0, order, seed=10, genome=g, dup_rate=0.5, skew=0.5, n_rate=0.001)
        _full_compare(bases, off, 31, 1000)
    # two-word k-mers, ragged reads, a genome that is mostly microsatellites
    g = synth.make_structured_genome(20000, seed=11, dispersed=2, tandem=120)
    bases, off = _reads(6000, 250, 0, "sorted", seed=12, genome=g, ragged=True, dup_rate=0.2)
    _full_compare(bases, off, 47, 700, window=512)


def test_deep_coverage_and_long_reads_in_order():
    # amplicon depth: 9 000 reads over 900 bases, sorted -- a read contains the proposals of more than a hundred reads before it (the
    # sequential pass's entry lists outgrow the rows it stages in LDS) -- and sorted long reads (k-mers by the thousand per read)
    bases, off = _reads(9000, 150, 900, "sorted", seed=14, err=0.02)
    _full_compare(bases, off, 31, 1000)
    bases, off = _reads(9000, 150, 900, "sorted", seed=15, err=0.03, ragged=True, dup_rate=0.3)
    _full_compare(bases, off, 21, 1000, window=3000)
    bases, off = _reads(1500, 2500, 60000, "sorted", seed=16, err=0.01)
    _full_compare(bases, off, 31, 300)


@pytest.mark.parametrize("chunk", ["1000", "64", "1"])
def test_the_sequential_pass_chunk_by_chunk(chunk, monkeypatch):
    # A window that leaves more than 2^19 reads unsettled goes through the sequential pass in chunks (a bit per key name in LDS), tent
    # re-proposed by what is left between two of them.  At full size that path is only checked by round trips (which any valid anchor
    # choice passes); LEON_CHAIN_CHUNK makes the chunks small -- 1000 reads (not a multiple of a step), one step, ONE read -- so that the
    # oracle checks its every stage: anchors, events, bytes.
    monkeypatch.setenv("LEON_CHAIN_CHUNK", chunk)
    n = 20000 if chunk != "1" else 3000
    bases, off = _reads(n, 150, 5 * n, "sorted", seed=21, dup_rate=0.2)
    ref, st = _full_compare(bases, off, 31, 1000)
    assert st["resolve_chain_reads"] > n // 4
    g = synth.make_structured_genome(20000, seed=22, dispersed=6, tandem=60)
    bases, off = _reads(n // 2, 200, 0, "sorted", seed=23, genome=g, ragged=True, dup_rate=0.3)
    _full_compare(bases, off, 41, 500, window=2500)


def test_every_read_the_same():
    # 3 000 copies of one read, then of its reverse complement: one anchor, every later read finds it
    g = synth.make_genome(400, seed=13)
    one = g[100:250].tobytes()
    rc = bytes(synth._COMP[np.frombuffer(one, dtype=np.uint8)][::-1])
    b, off = O.reads_to_arrays([one] * 3000 + [rc] * 3000)
    _full_compare(b, off, 31, 1000)
    _full_compare(b, off, 31, 1000, window=100)

"""-m "not gpu": the host-side helpers of bench.py that decide what the GPU runs measure -- the genome's k-mers that fill the
bloom of the large configurations (`--bloom-from genome`), the synthetic FASTQ of `end_to_end`, the header set of `streams`,
the batch plan -- checked on the CPU (torch tensors on the CPU device)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import oracle_lib as O  # noqa: E402


@pytest.mark.parametrize("k", [21, 31, 32, 33, 63])
def test_genome_kmers_are_the_canonical_kmers_the_oracle_counts(k):
    dev = torch.device("cpu")
    g = bench.gen_genome(4000, dev)
    km = bench.genome_kmers_chunk(g, 7, 1207, k).numpy().view(np.uint64).reshape(-1, O.kwords(k))
    seq = np.frombuffer(b"ACTG", dtype=np.uint8)[g.numpy()].tobytes()[7:1207 + k - 1]
    solid = np.asarray(O.count_solid(seq, np.array([0, len(seq)], dtype=np.uint64), k, 1), dtype=np.uint64).reshape(-1, O.kwords(k))
    assert set(map(tuple, km.tolist())) == set(map(tuple, solid.tolist()))
    assert len(km) == 1200


def test_synthetic_fastq_and_headers(tmp_path):
    dev = torch.device("cpu")
    old = bench.CHUNK
    bench.CHUNK = 700
    try:
        fq = str(tmp_path / "t.fastq")
        bench.write_fastq(fq, 1500, 150, dev)
    finally:
        bench.CHUNK = old
    lines = open(fq, "rb").read().split(b"\n")
    assert len(lines) == 4 * 1500 + 1 and lines[-1] == b""
    assert lines[0].startswith(b"@SRR387476.1 ") and lines[4 * 1499].startswith(b"@SRR387476.1500 ")
    assert all(len(l) == 150 and set(l) <= set(b"ACGT") for l in lines[1::4]) and all(l == b"+" for l in lines[2::4])
    assert all(len(l) == 150 for l in lines[3::4])
    leon = os.path.join(ROOT, "leon_amd", "lib", "leon")
    if os.path.exists(leon):                                   # the product's reader agrees
        r = subprocess.run([leon, "-selftest-bank", fq], capture_output=True, text=True)
        assert r.returncode == 0 and json.loads(r.stdout)["reads"] == 1500 and json.loads(r.stdout)["bases"] == 1500 * 150
    blob, off = bench.sra_headers(12345, seed=3)
    assert off[0] == 0 and off[-1] == len(blob) and np.all(np.diff(off) > 0)
    heads = [blob[int(off[i]):int(off[i + 1])].tobytes() for i in (0, 8, 9, 99, 12344)]
    assert [h.split(b" ")[0] for h in heads] == [b"SRR387476.1", b"SRR387476.9", b"SRR387476.10", b"SRR387476.100", b"SRR387476.12345"]
    assert all(h.endswith(b" length=150") and h.count(b":") == 4 for h in heads)


def test_walk_traffic_file_names_its_kernel():
    tj = json.load(open(os.path.join(ROOT, "profiles", "walk_traffic.json")))
    assert len(tj["k_walk_source"]) == 16 and tj["traffic_bytes"] == int((tj["fetch_size_kb"] + tj["write_size_kb"]) * 1024)
    assert len(bench.walk_source_id()) == 16


def test_structured_generators_on_the_cpu_device():
    """bench.gen_structured_genome / gen_structured_reads (the `structured` entry, tests/test_gpu_fullsize.py): shapes, alphabet,
    determinism, ragged lengths; in "sorted" order every read maps back to a start that never decreases (error-free reads here)"""
    dev = torch.device("cpu")
    old = bench.CHUNK
    bench.CHUNK = 1000
    try:
        g = bench.gen_structured_genome(20000, dev)
        flat, off = bench.gen_structured_reads(g, 2500, 100, dev, order="sorted", err=0.0, dup_rate=0.2, skew=0.3, ragged=True)
        flat2, off2 = bench.gen_structured_reads(g, 2500, 100, dev, order="sorted", err=0.0, dup_rate=0.2, skew=0.3, ragged=True)
    finally:
        bench.CHUNK = old
    assert torch.equal(flat, flat2) and torch.equal(off, off2)
    lens = np.diff(off.numpy())
    assert off[0] == 0 and off[-1] == flat.numel() and lens.min() >= 50 and lens.max() <= 100 and len(set(lens.tolist())) > 10
    text = flat.numpy().tobytes()
    assert set(text) <= set(b"ACGT")
    # over a genome WITHOUT repeats every read has one place: in "sorted" order the places never go backwards, and a fifth are duplicates
    g = bench.gen_genome(20000, dev)
    flat, off = bench.gen_structured_reads(g, 2500, 100, dev, order="sorted", err=0.0, dup_rate=0.2, skew=0.0, ragged=False)
    text = flat.numpy().tobytes()
    gs = np.frombuffer(b"ACTG", dtype=np.uint8)[g.numpy()].tobytes()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    places = []
    for i in range(2500):
        r = text[100 * i:100 * i + 100]
        p = gs.find(r)
        places.append(p if p >= 0 else gs.find(r.translate(comp)[::-1]))
    assert min(places) >= 0 and places == sorted(places)
    assert 300 < sum(a == b_ for a, b_ in zip(places, places[1:])) < 900
    # "pairs": a forward read, then its mate ~2.2 read lengths downstream on the other strand
    flat, off = bench.gen_structured_reads(g, 2000, 100, dev, order="pairs", err=0.0, dup_rate=0.0, skew=0.0, ragged=False)
    text = flat.numpy().tobytes()
    for i in range(0, 2000, 2):
        a_, b_ = text[100 * i:100 * i + 100], text[100 * i + 100:100 * i + 200]
        pa, pb = gs.find(a_), gs.find(b_.translate(comp)[::-1])
        assert pa >= 0 and pb >= 0 and (pb - pa == 220 or pb == 20000 - 100)


def test_watch_names_the_collective_and_exits(tmp_path):
    """bench.Watch: a collective that does not complete is reported by name with the rank, and the process exits non-zero at once"""
    script = "import os, sys, time\nsys.path.insert(0, %r)\nimport bench\nw = bench.Watch(5, None)\nwith w('all_to_all_single (test) [3 sent]'):\n    time.sleep(30)\nprint('survived')\n" % ROOT
    p = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, LEON_BENCH_COLL_TIMEOUT="1"), capture_output=True, text=True, timeout=60)
    assert p.returncode == 3 and "survived" not in p.stdout
    err = json.loads([l for l in p.stderr.splitlines() if l.startswith("{")][-1])
    assert err["rank"] == 5 and "all_to_all_single (test)" in err["bench_error"] and "did not complete" in err["bench_error"]
    script = "import sys\nsys.path.insert(0, %r)\nimport bench\nw = bench.Watch(0, None)\nwith w('broadcast (x)'):\n    raise RuntimeError('boom')\n" % ROOT
    p = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3 and "broadcast (x)" in p.stderr and "boom" in p.stderr

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

"""Generates tests/golden/*.json: SELF-golden vectors of the CPU restatement (oracle/leon_oracle.c).
They are NOT reference Leon output -- the reference cannot be built here (gatb-core absent) and ships no
golden vectors; these fixtures freeze the restatement so that a change to it (or to the HIP path checked
against it) is visible.  Re-run: python tests/make_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import common  # noqa: E402
import oracle_lib as O  # noqa: E402


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def case(name, bases, off, k, rpb, min_abundance=3):
    bl, solid, tai = common.make_bloom(bases, off, k, min_abundance)
    res = O.encode(bases, off, k, rpb, bl)
    return {
        "name": name, "k": k, "reads_per_block": rpb, "min_abundance": min_abundance, "n_reads": len(off) - 1,
        "n_bases": len(bases), "input_sha256": sha(bases), "n_solid": len(solid), "bloom_tai": tai,
        "bloom_sha256": sha(bl.bits.tobytes()), "n_anchors": res.n_anchors, "n_symbols": res.n_symbols,
        "anchor_dict_sha256": sha(res.anchor_dict), "anchor_dict_bytes": len(res.anchor_dict),
        "block_sizes": [len(b) for b in res.blocks], "block_nreads": res.block_nreads,
        "block_sha256": [sha(b) for b in res.blocks],
        "first_block_head_hex": res.blocks[0][:32].hex() if res.blocks else "",
        "anchor_pos_sha256": sha(res.anchor_pos.tobytes()), "events_sha256": sha(res.events.tobytes()),
    }


def known_answers():
    """small known-answer vectors of the primitives"""
    rc_syms = [(0, 1), (1, 4), (2, 3), (8, 1), (9, 200), (3, 0), (3, 1), (8, 2), (9, 7), (10, 1)]
    sizes = [2, 5, 5, 2, 3, 3, 3, 2] + [256] * 72
    payload = O.rc_encode_stream([m for m, _ in rc_syms], [v for _, v in rc_syms], sizes)
    bl = O.Bloom(5000, 31)
    kmers = np.array([0x0123456789ABCDE, 0x3FFFFFFFFFFFFFFF, 0, 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1)], dtype=np.uint64)
    bl.insert(kmers)
    return {
        "hash64": [[hex(k), hex(s), hex(O.lib.lo_hash64(k, s))] for k, s in
                   [(0, 0), (1, 0), (0x0123456789ABCDEF, O.lib.lo_hash_seed(0)), (2 ** 62 - 1, O.lib.lo_hash_seed(0))]],
        "hash_seed0": hex(O.lib.lo_hash_seed(0)),
        "random_values_head": [hex(O.lib.lo_random_value(i)) for i in range(4)],
        "revcomp_k31": [[hex(int(x)), hex(O.lib.lo_revcomp(int(x), 31))] for x in kmers],
        "rc_stream": {"symbols": rc_syms, "payload_hex": payload.hex()},
        "bloom_5000_k31": {"kmers": [hex(int(x)) for x in kmers], "nbytes": len(bl.bits),
                           "set_bits": [int(i) for i in np.flatnonzero(np.unpackbits(bl.bits, bitorder="little"))],
                           "contains4_right": [bl.contains4(int(x), 1) for x in kmers],
                           "contains4_left": [bl.contains4(int(x), 0) for x in kmers]},
    }


def stream_cases():
    """self-golden vectors of the header stream (oracle's HeaderEncoder) and of the lossy quality rule (smoothQuals)"""
    import hdr_samples as H
    import synth
    out = {"header": [], "qual_smooth": []}
    toy_heads = [l[1:].rstrip("\n").encode() for l in open(os.path.join(common.GOLDEN, "toy.fasta")) if l.startswith(">")]
    for name, heads, rpb in (("toy.fasta headers rpb50000", toy_heads, 50000), ("toy.fasta headers rpb64", toy_heads, 64),
                             ("sra 3000 rpb1000", H.sra(3000, seed=1), 1000), ("nasty 600 rpb100", H.nasty(600, seed=3), 100)):
        blocks = [O.header_encode_block(heads[i:i + rpb], heads[0]) for i in range(0, len(heads), rpb)]
        out["header"].append({"name": name, "n": len(heads), "reads_per_block": rpb, "input_sha256": sha(b"\n".join(heads)),
                              "block_sizes": [len(b) for b in blocks], "block_sha256": [sha(b) for b in blocks],
                              "first_block_head_hex": blocks[0][:32].hex()})
    bases, off = common.synthetic(800, 120, 5000, seed=103, n_rate=0.003, err=0.02)
    reads = [bases[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
    quals = [q[:len(r)].ljust(len(r), b"J") for q, r in zip(H.fastq_quals(len(reads), 130, seed=9), reads)]
    for k in (21, 31):
        bl, solid, tai = common.make_bloom(bases, off, k)
        sm = b"".join(O.qual_smooth(bl, k, r, q) for r, q in zip(reads, quals))
        out["qual_smooth"].append({"name": "synthetic 800x120 k%d" % k, "k": k, "bloom_tai": tai, "bloom_sha256": sha(bl.bits.tobytes()),
                                   "quals_sha256": sha(b"".join(quals)), "smoothed_sha256": sha(sm),
                                   "changed": sum(a != b for a, b in zip(sm, b"".join(quals)))})
    return out


def main():
    out = {"_note": "SELF-golden vectors of oracle/leon_oracle.c (parity with reference Leon is UNPINNED)", "cases": []}
    bases, off = common.toy_reads()
    out["cases"].append(case("toy.fasta k31 rpb50", bases, off, 31, 50))
    out["cases"].append(case("toy.fasta k31 rpb50000", bases, off, 31, 50000))
    bases, off = common.synthetic(3000, 150, 12000, seed=101, n_rate=0.002)
    out["cases"].append(case("synthetic 3000x150 N0.002", bases, off, 31, 1000))
    bases, off = common.synthetic(2000, 100, 6000, seed=102, ragged=True)
    out["cases"].append(case("synthetic ragged 2000x<=100 k21", bases, off, 21, 700))
    out["known_answers"] = known_answers()
    out["streams"] = stream_cases()
    with open(os.path.join(common.GOLDEN, "self_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.join(common.GOLDEN, "self_golden.json"))


if __name__ == "__main__":
    main()

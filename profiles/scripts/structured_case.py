"""python profiles/scripts/structured_case.py [reads] [order] [k] [L]: one file with real-genome structure (bench.structured_case) through
the device path; prints its JSON line.  LEON_TRACE_CHAIN=1 adds the sequential pass's per-window counters on stderr."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
order = sys.argv[2] if len(sys.argv) > 2 else "sorted"
k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
L = int(sys.argv[4]) if len(sys.argv) > 4 else 150
kw = {}
if os.environ.get("PLAIN") == "1":            # the i.i.d. shape in another order: no duplicates, no skew, fixed length
    kw = dict(dup_rate=0.0, skew=0.0, ragged=False)
for name, key in (("DUP_RATE", "dup_rate"), ("SKEW", "skew")):      # e.g. DUP_RATE=0.5 SKEW=0.9: half the reads duplicates, nine tenths of them on a tenth of the genome
    if name in os.environ:
        kw[key] = float(os.environ[name])
G = int(os.environ.get("GENOME", 0))               # GENOME=30000: amplicon depth (reads * L / GENOME)
print(json.dumps(bench.structured_case(n, order=order, k=k, L_=L, G=G, decode=os.environ.get("DECODE", "1") == "1", **kw)))

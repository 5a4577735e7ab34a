"""`leon -c` / `-d` through the C++ host mirror on a synthetic FASTQ at scale (default 10 M x 150 bp = BASELINE config #2's
read count as FASTQ): wall time of each command and the sizes of the three streams.  Prints one JSON line."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

N = int(os.environ.get("LEON_CLI_READS", 10_000_000))
L = 150
work = os.environ.get("LEON_CLI_DIR", "/tmp/leon_cli")
os.makedirs(work, exist_ok=True)
fq = os.path.join(work, "reads.fastq")
dev = torch.device("cuda", 0)
genome = bench.gen_genome(N * L // 30, dev)
t0 = time.time()
with open(fq, "wb") as f:
    for c0 in range((N + bench.CHUNK - 1) // bench.CHUNK):
        n = min(bench.CHUNK, N - c0 * bench.CHUNK)
        reads = bench.gen_reads_chunk(genome, c0, bench.CHUNK, 0.01, dev, L=L)[:n].cpu().numpy()
        rng = np.random.default_rng(c0)
        idx = np.arange(c0 * bench.CHUNK, c0 * bench.CHUNK + n)
        head = np.char.add(np.char.add(b"@SRR387476.", (idx + 1).astype("S")), np.char.add(b" HWI-ST1234:3:1101:", np.char.add(rng.integers(1000, 20000, n).astype("S"), b" length=150\n")))
        quals = np.frombuffer(b"#5:?ABCDEFGHIJ", dtype=np.uint8)[np.minimum(rng.integers(0, 14, (n, L)), rng.integers(4, 14, (n, 1)))]
        rec = [h + r.tobytes() + b"\n+\n" + q.tobytes() + b"\n" for h, r, q in zip(head.tolist(), reads, quals)]
        f.write(b"".join(rec))
        if c0 % 10 == 9:
            print("generated %d M reads, %.0f s" % (c0 + 1, time.time() - t0), file=sys.stderr, flush=True)
gen_s = time.time() - t0
del genome
torch.cuda.empty_cache()
leon = os.path.join(ROOT, "leon_amd", "lib", "leon")
out = {"reads": N, "read_len": L, "fastq_bytes": os.path.getsize(fq), "generate_s": round(gen_s, 1)}


def timed(name, *args):
    t = time.time()
    r = subprocess.run([leon] + list(args), capture_output=True, text=True)
    out[name + "_s"] = round(time.time() - t, 2)
    out[name + "_rc"] = r.returncode
    out[name + "_stdout"] = r.stdout.strip().splitlines()
    trace = [l for l in r.stderr.splitlines() if l.startswith("[leon ")]
    if trace:
        out[name + "_trace"] = trace[:12] + (["... %d more" % (len(trace) - 12)] if len(trace) > 12 else [])
    if r.returncode:
        out[name + "_stderr"] = r.stderr[-500:]


only = os.environ.get("LEON_CLI_ONLY", "")                    # e.g. "lossless": just that command (measurement runs)
timed("compress_lossless", "-file", fq, "-c", "-lossless", "-verbose", "1")
out["leon_bytes_lossless"] = os.path.getsize(fq + ".leon") if os.path.exists(fq + ".leon") else None
if only != "lossless":
    timed("decompress", "-file", fq + ".leon", "-d", "-test-file", "-verbose", "1")
    if os.environ.get("LEON_CLI_SKIP_LOSSY") != "1":
        timed("compress_lossy", "-file", fq, "-c", "-verbose", "1")
        out["leon_bytes_lossy"] = os.path.getsize(fq + ".leon") if os.path.exists(fq + ".leon") else None
for f in (fq, fq + ".leon", fq + ".d"):
    if os.path.exists(f):
        os.remove(f)
print(json.dumps(out))

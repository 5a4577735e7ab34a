"""`leon -d` at configuration #3's size under round / call shapes: ONE 100 M-read FASTQ generated and compressed once (quality blocks by the
device's deflate, to keep the run short), then decompressed under each setting.  Prints one JSON line."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("LEON_CLI_READS", "100000000")
N = int(os.environ["LEON_CLI_READS"])
work = os.environ.get("LEON_CLI_DIR", "/dev/shm/leon_cli")
os.makedirs(work, exist_ok=True)
fq = os.path.join(work, "reads.fastq")
import torch  # noqa: E402
import bench  # noqa: E402
t0 = time.time()
bench.write_fastq(fq, N, 150, torch.device("cuda", 0))
out = {"reads": N, "generate_s": round(time.time() - t0, 1), "fastq_bytes": os.path.getsize(fq)}
leon = os.path.join(ROOT, "leon_amd", "lib", "leon")
t = time.time()
r = subprocess.run([leon, "-file", fq, "-c", "-lossless", "-qual-deflate", "device"], capture_output=True, text=True)
out["compress_s"] = round(time.time() - t, 2)
assert r.returncode == 0, r.stderr[-500:]
runs = []
for env in ({}, {"LEON_DECODE_BLOCKS": "250"}, {"LEON_DECODE_BLOCKS": "250", "LEON_DECODE_DNA_ROUNDS": "4"}, {"LEON_DECODE_BLOCKS": "334", "LEON_DECODE_DNA_ROUNDS": "3"},
            {"LEON_DECODE_BLOCKS": "125", "LEON_DECODE_DNA_ROUNDS": "4"}):
    t = time.time()
    r = subprocess.run([leon, "-file", fq + ".leon", "-d", "-verbose", "1"], capture_output=True, text=True, env=dict(os.environ, **env))
    dt = time.time() - t
    line = next((l for l in r.stdout.splitlines() if l.startswith("time:")), "")
    runs.append({"env": env, "wall_s": round(dt, 2), "rc": r.returncode, "time_line": line})
    print(env, round(dt, 2), line, file=sys.stderr, flush=True)
out["runs"] = runs
for f in (fq, fq + ".leon", fq + ".d"):
    if os.path.exists(f):
        os.remove(f)
print(json.dumps(out))

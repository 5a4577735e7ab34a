"""`leon -c -lossless -verbose 1` / `-c` (lossy) / `-d -test-file -verbose 1` on the bench's 10 M-read FASTQ: the CLI's own stage times"""
import os, subprocess, sys, tempfile, shutil, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
n = int(os.environ.get("LEON_CLI_READS", 10_000_000))
work = tempfile.mkdtemp(prefix="leon_cli_", dir="/dev/shm")
fq = os.path.join(work, "reads.fastq")
bench.write_fastq(fq, n, 150, torch.device("cuda", 0))
leon = os.path.join(ROOT, "leon_amd", "lib", "leon")
try:
    for args in (["-file", fq, "-c", "-lossless", "-verbose", "1"], ["-file", fq + ".leon", "-d", "-test-file", "-verbose", "1"], ["-file", fq, "-c", "-verbose", "1"]):
        t = time.time()
        r = subprocess.run([leon] + args, capture_output=True, text=True, env=dict(os.environ, LEON_TRACE_ALLOC="1"))
        print(" ".join(args[2:]), "->", round(time.time() - t, 2), "s rc", r.returncode)
        print(r.stdout.strip()); print(r.stderr.strip()[-1500:])
finally:
    shutil.rmtree(work, ignore_errors=True)

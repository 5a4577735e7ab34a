"""`leon -d` on BASELINE configuration #3's file with different numbers of pipelined rounds (LEON_DECODE_BLOCKS): one FASTQ, one
`-c -lossless`, then `-d` per setting.  Prints one JSON line."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("LEON_CLI_READS", "100000000")
os.environ["LEON_CLI_KEEP"] = "1"
N = int(os.environ["LEON_CLI_READS"])
work = os.environ.get("LEON_CLI_DIR", "/dev/shm/leon_cli")
import importlib.util
spec = importlib.util.spec_from_file_location("gen", os.path.join(ROOT, "profiles", "scripts", "cli_at_scale.py"))
src = open(os.path.join(ROOT, "profiles", "scripts", "cli_at_scale.py")).read()
gen_part = src[:src.index("leon = os.path.join(ROOT")]
exec(compile(gen_part, "gen", "exec"))                        # writes reads.fastq (fq), defines out / gen_s
leon = os.path.join(ROOT, "leon_amd", "lib", "leon")
res = {"reads": N}
t = time.time(); r = subprocess.run([leon, "-file", fq, "-c", "-lossless"], capture_output=True, text=True); res["compress_s"] = round(time.time() - t, 2); res["compress_rc"] = r.returncode
for blocks in os.environ.get("LEON_ROUNDS_BLOCKS", "2000,1000,667,500,334").split(","):
    for hdr in os.environ.get("LEON_ROUNDS_HEADER", "384").split(","):      # LEON_HEADER_DEVICE_BLOCKS: 384 = the default, a huge number = host threads only
        for dna in os.environ.get("LEON_ROUNDS_DNA", "2").split(","):       # LEON_DECODE_DNA_ROUNDS
            t = time.time()
            r = subprocess.run([leon, "-file", fq + ".leon", "-d", "-verbose", "1"], capture_output=True, text=True,
                               env=dict(os.environ, LEON_DECODE_BLOCKS=blocks, LEON_HEADER_DEVICE_BLOCKS=hdr, LEON_DECODE_DNA_ROUNDS=dna))
            key = "decode_%s_blocks_per_round_hdr%s_dna%s" % (blocks, hdr, dna)
            res[key] = {"s": round(time.time() - t, 2), "rc": r.returncode, "line": r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}
            print(key, res[key], file=sys.stderr, flush=True)
for f in (fq, fq + ".leon", fq + ".d"):
    if os.path.exists(f):
        os.remove(f)
print(json.dumps(res))

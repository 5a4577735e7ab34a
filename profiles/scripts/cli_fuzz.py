"""One-off fuzz of the C++ host mirror: small random FASTQ / FASTA files of awkward shapes through `leon -c` and
`leon -d -test-file` (byte comparison with the original).  Seeded; prints one line per file."""
import gzip, os, random, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
leon = os.path.join(ROOT, "leon_amd", "lib", "leon")
seed = int(os.environ.get("LEON_FUZZ_SEED", 1))
n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(seed)
work = tempfile.mkdtemp(prefix="leon_clifuzz_")
bad = 0
try:
    for f in range(n_files):
        fastq = rnd.random() < 0.75
        n = rnd.choice([0, 1, 2, 7, 100, 999, 2500, 5000])
        genome = "".join(rnd.choice("ACGT") for _ in range(rnd.choice([50, 400, 3000])))
        maxlen = rnd.choice([10, 60, 150, 400])
        plus_mode = rnd.choice(["bare", "header", "mixed"])
        crlf = rnd.random() < 0.15
        nl = "\r\n" if crlf else "\n"
        recs = []
        for i in range(n):
            L = rnd.choice([0, 1, 5, 20, 31, 32, maxlen, rnd.randint(0, maxlen)]) if rnd.random() < 0.3 else maxlen
            p = rnd.randint(0, max(0, len(genome) - 1))
            s = (genome[p:] + genome * (L // len(genome) + 1))[:L]
            s = "".join(("N" if rnd.random() < 0.01 else (rnd.choice("ACGT") if rnd.random() < 0.02 else c)) for c in s)
            if rnd.random() < 0.5:
                s = s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))
            style = rnd.random()
            h = ("SRR%d.%d HWI:%d:%d length=%d" % (seed, i + 1, rnd.randint(1, 9), rnd.randint(1000, 99999), L)) if style < 0.7 else \
                ("" if style < 0.75 else "".join(rnd.choice("abc XYZ_:/.-0123456789") for _ in range(rnd.randint(1, 40))))
            if fastq:
                q = "".join(rnd.choice("#5:?ABCDEFGHIJ") for _ in range(len(s)))
                plus = "" if plus_mode == "bare" else h if plus_mode == "header" else rnd.choice(["", h, h, "other text %d" % i])
                recs.append("@%s%s%s%s+%s%s%s%s" % (h, nl, s, nl, plus, nl, q, nl))
            else:
                if L == 0:
                    recs.append(">%s%s%s" % (h, nl, nl) if rnd.random() < 0.5 else ">%s%s" % (h, nl))
                else:
                    recs.append(">%s%s%s%s" % (h, nl, s, nl))
        name = os.path.join(work, "f%d.%s" % (f, "fastq" if fastq else "fasta"))
        data = "".join(recs).encode()
        open(name, "wb").write(data)
        src = name
        if rnd.random() < 0.25:
            with gzip.open(name + ".gz", "wb") as g:
                g.write(data)
            src = name + ".gz"
        flags = rnd.choice([["-lossless"], ["-lossless"], ["-lossless", "-kmer-size", "21"], ["-lossless", "-kmer-size", "47"]])
        env = dict(os.environ)
        if rnd.random() < 0.5:
            env["LEON_DECODE_BLOCKS"] = "1"
        if rnd.random() < 0.3:
            env["LEON_BATCH_BLOCKS"] = "1"
        r1 = subprocess.run([leon, "-file", src, "-c"] + flags, capture_output=True, text=True, env=env)
        ok, why = True, ""
        if r1.returncode != 0:
            ok, why = (n == 0 or not data.strip()), "compress rc %d: %s" % (r1.returncode, r1.stderr.strip()[-200:])
        else:
            r2 = subprocess.run([leon, "-file", name + ".leon", "-d"], capture_output=True, text=True, env=env)
            if r2.returncode != 0:
                ok, why = False, "decompress rc %d: %s" % (r2.returncode, r2.stderr.strip()[-200:])
            else:
                got = open(name + ".d", "rb").read()
                # what comes back: LF line ends, FASTA sequences on one line, records of empty FASTA sequences as '>h\n\n'
                want = data.replace(b"\r\n", b"\n")
                if got != want:
                    if not fastq and got.replace(b"\n\n", b"\n") == want.replace(b"\n\n", b"\n"):
                        why = "(empty FASTA sequence lines differ: accepted)"
                    else:
                        ok, why = False, "output differs (%d vs %d bytes)" % (len(got), len(want))
        print("file %d: %s n=%d maxlen=%d plus=%s crlf=%s %s %s -> %s %s" % (f, "fastq" if fastq else "fasta", n, maxlen, plus_mode, crlf, os.path.basename(src), " ".join(flags), "ok" if ok else "FAILED", why), flush=True)
        bad += not ok
        for p in (name, name + ".gz", name + ".leon", name + ".d"):
            if os.path.exists(p):
                os.remove(p)
finally:
    shutil.rmtree(work, ignore_errors=True)
print("cli fuzz: %d files, %d failures" % (n_files, bad))
sys.exit(1 if bad else 0)

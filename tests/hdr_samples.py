"""Seeded header sets for the header-stream tests: the shapes FASTA/FASTQ files carry (SRA, Illumina, the toy file's
simulator fields) plus adversarial ones (empty, binary bytes, hundreds of fields, long digit runs, leading zeros)."""
import random


def sra(n, seed=1):
    rnd = random.Random(seed)
    out = []
    for i in range(n):
        x, y = rnd.randint(1000, 20000), rnd.randint(1000, 200000)
        out.append(b"SRR387476.%d HWI-ST1234:%d:%d:%d:%d length=%d" % (i + 1, rnd.randint(1, 8), 1100 + i // 5000, x, y, 100))
    return out


def toy_like(n, seed=2):
    rnd = random.Random(seed)
    return [b"read%d_contig0_position%d_M%d_I0_D0_NG0______er0.01_indel0_rev%d" % (i, rnd.randint(0, 5000), rnd.randint(0, 4), rnd.randint(0, 1))
            for i in range(n)]


def nasty(n, seed=3):
    rnd = random.Random(seed)
    out = []
    for i in range(n):
        t = rnd.random()
        if t < 0.15:
            out.append(b"")
        elif t < 0.3:
            out.append(bytes(rnd.randrange(256) for _ in range(rnd.randint(0, 60))))
        elif t < 0.45:
            out.append(b"000%d__x  y//%d" % (rnd.randint(0, 9999), rnd.randint(0, 10 ** 17)))
        elif t < 0.55:
            out.append(b"0000 00 0 123456789012345678901234 007 0042x 18446744073709551615 999999999999999999")
        elif t < 0.65:
            out.append(b" ".join(b"f%d" % (j + (j == rnd.randint(0, 400))) for j in range(rnd.randint(250, 400))))
        elif t < 0.75:
            out.append(b"a" * rnd.randint(250, 700) + b":" + b"9" * rnd.randint(1, 40))
        elif t < 0.85:
            out.append(out[-1] if out else b"same")
        else:
            out.append(b"x%dy z=%d;%s" % (i, rnd.randint(0, 3), b"\0" * rnd.randint(0, 2)))
    return out


def fastq_quals(n, L, seed=4):
    rnd = random.Random(seed)
    alphabet = b"#,-5:<>?@ABCDEFGHIJ"
    return [bytes(rnd.choice(alphabet) for _ in range(L if L > 0 else rnd.randint(0, 200))) for _ in range(n)]

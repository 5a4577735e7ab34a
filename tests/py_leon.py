"""A second, independent restatement of the DNA encode path in pure Python (small inputs only), written from the
rules of DESIGN.md section 1.1 rather than from oracle/leon_oracle.c.  tests/test_oracle_cpu.py checks the C oracle
against it byte for byte; together with the round trip this is what pins the oracle (reference parity: unpinned)."""

M64 = (1 << 64) - 1
CODE = {"A": 0, "C": 1, "T": 2, "G": 3}
CANO2 = [0, 1, 2, 3, 4, 5, 3, 7, 8, 9, 0, 4, 9, 13, 1, 5]


def hash64(key, seed):
    h = seed
    h ^= ((h << 7) & M64) ^ ((key * (h >> 3)) & M64) ^ (~(((h << 11) + (key ^ (h >> 5))) & M64) & M64)
    h = ((~h & M64) + ((h << 21) & M64)) & M64
    h ^= h >> 24
    h = (h + ((h << 3) & M64) + ((h << 8) & M64)) & M64
    h ^= h >> 14
    h = (h + ((h << 2) & M64) + ((h << 4) & M64)) & M64
    h ^= h >> 28
    return (h + ((h << 31) & M64)) & M64


def random_value(i):
    s = (0x4C454F4E + (i + 1) * 0x9E3779B97F4A7C15) & M64
    s = ((s ^ (s >> 30)) * 0xBF58476D1CE4E5B9) & M64
    s = ((s ^ (s >> 27)) * 0x94D049BB133111EB) & M64
    return s ^ (s >> 31)


RV = [random_value(i) for i in range(256)]
SEED0 = (0xAAAAAAAA55555555 * 0xB5B5B5B54B4B4B4B) & M64


def revcomp(x, k):
    r = 0
    for _ in range(k):
        r = (r << 2) | ((x & 3) ^ 2)
        x >>= 2
    return r


def canonical(x, k):
    return min(x, revcomp(x, k))


class Bloom:
    """BloomNeighborCoherent(tai_bloom, k, 7, 12)"""

    def __init__(self, tai_bloom, k, n_hash=7, block_nbits=12):
        self.k, self.n_hash, self.blk = k, n_hash, 1 << block_nbits
        tai = tai_bloom + 2 * self.blk
        self.nchar = 1 + tai // 8
        if tai & (tai - 1) == 0:
            tai -= 1
        self.reduced = tai - 2 * self.blk
        self.bits = bytearray(self.nchar)
        self.words = 2 if k >= 32 else 1

    def _hash1(self, x):
        h = hash64(x & M64, SEED0)
        if self.words == 2:
            h ^= hash64(x >> 64, SEED0)
        return h

    def _positions(self, item):
        k = self.k
        pv = CANO2[((item >> (2 * (k - 1))) & 3) * 4 + (item & 3)]
        mid = canonical((item >> 2) & ((1 << (2 * (k - 2))) - 1), k - 2)
        h0 = self._hash1(mid) % self.reduced + pv
        low = mid & M64
        pos = [h0]
        for i in range(1, self.n_hash):
            sh = (RV[(low >> i) & 255] ^ RV[(low >> (i + 8)) & 255]) & (self.blk - 1)
            pos.append(h0 + sh)
        return pos

    def insert(self, item):
        for p in self._positions(item):
            self.bits[p >> 3] |= 1 << (p & 7)

    def contains(self, item):
        return all(self.bits[p >> 3] >> (p & 7) & 1 for p in self._positions(item))

    def contains4(self, kmer, right):
        k = self.k
        res = 0
        for nt in range(4):
            nb = (((kmer << 2) | nt) & ((1 << (2 * k)) - 1)) if right else ((kmer >> 2) | (nt << (2 * (k - 1))))
            if self.contains(nb):
                res |= 1 << nt
        return res


class Model:
    def __init__(self, n):
        self.n, self.r = n, list(range(n + 1))

    def update(self, c):
        for i in range(c + 1, self.n + 1):
            self.r[i] += 1


class RangeEncoder:
    TOP, BOTTOM = 1 << 56, 1 << 48

    def __init__(self):
        self.low, self.range, self.out = 0, M64, bytearray()

    def encode(self, m, c):
        self.range //= m.r[m.n]
        self.low = (self.low + m.r[c] * self.range) & M64
        self.range = (self.range * (m.r[c + 1] - m.r[c])) & M64
        while True:
            if (self.low ^ ((self.low + self.range) & M64)) < self.TOP:
                pass
            elif self.range < self.BOTTOM:
                self.range = (-self.low) & (self.BOTTOM - 1)
            else:
                break
            self.out.append(self.low >> 56)
            self.range = (self.range << 8) & M64
            self.low = (self.low << 8) & M64
        m.update(c)

    def flush(self):
        for _ in range(8):
            self.out.append(self.low >> 56)
            self.low = (self.low << 8) & M64


def encode(reads, k, reads_per_block, bloom):
    """reads: list of str.  Returns (block payloads, dictionary stream, anchors) with -nb-cores 1 semantics."""
    anchors, order = {}, []
    drc, dmodel = RangeEncoder(), Model(5)
    blocks = []

    def new_models():
        return {
            "type": Model(2), "noanchor": Model(5), "bif": Model(5), "bin": Model(2), "sizeDT": Model(3),
            "posDT": Model(3), "addrDT": Model(3), "rev": Model(2),
            **{g: [Model(256) for _ in range(9)] for g in ("addr", "pos", "nasize", "size", "npos", "errpos", "num", "nerr")},
        }

    def numeric(rc, models, v):
        bc = 1
        while bc < 8 and (v >> (8 * bc)):
            bc += 1
        rc.encode(models[0], bc)
        for i in range(bc):
            rc.encode(models[i + 1], (v >> (8 * i)) & 255)

    def delta(rc, mt, models, value, prev):
        if value > prev and value - prev < value:
            t, d = 1, value - prev
        elif value <= prev and prev - value < value:
            t, d = 2, prev - value
        else:
            t, d = 0, value
        rc.encode(mt, t)
        numeric(rc, models, d)

    for b0 in range(0, len(reads), reads_per_block):
        rc, M = RangeEncoder(), new_models()
        prev_size = prev_pos = prev_addr = 0
        for read in reads[b0:b0 + reads_per_block]:
            L = len(read)
            npos = [i for i, ch in enumerate(read) if ch not in CODE]
            seq = [CODE.get(ch, 0) for ch in read]                      # N -> 'A'
            kmers = []
            if L >= k:
                km, mask = 0, (1 << (2 * k)) - 1
                for i, c in enumerate(seq):
                    km = ((km << 2) | c) & mask
                    if i + 1 >= k:
                        kmers.append(km)
            apos, addr = -1, 0
            for i, km in enumerate(kmers):                               # findExistingAnchor
                if canonical(km, k) in anchors:
                    apos, addr = i, anchors[canonical(km, k)]
                    break
            if apos < 0 and kmers:                                       # Leon::findAndInsertAnchor
                n = len(kmers)
                for i in list(range(n // 2, min(n // 2 + 10, n))) + list(range(0, n // 2)) + list(range(min(n // 2 + 10, n), n)):
                    cm = canonical(kmers[i], k)
                    if bloom.contains(cm):
                        addr = anchors[cm] = len(order)
                        order.append(cm)
                        for j in range(k):
                            drc.encode(dmodel, (cm >> (2 * (k - 1 - j))) & 3)
                        apos = i
                        break
            if apos < 0:                                                 # encodeNoAnchorRead
                rc.encode(M["type"], 1)
                numeric(rc, M["nasize"], L)
                for ch in read:
                    rc.encode(M["noanchor"], CODE.get(ch, 4))
                continue
            rc.encode(M["type"], 0)
            delta(rc, M["sizeDT"], M["size"], L, prev_size); prev_size = L
            delta(rc, M["posDT"], M["pos"], apos, prev_pos); prev_pos = apos
            delta(rc, M["addrDT"], M["addr"], addr, prev_addr); prev_addr = addr
            anchor = kmers[apos]
            rc.encode(M["rev"], 0 if anchor == canonical(anchor, k) else 1)
            bifs, errs = [], []
            for right in (False, True):
                km = anchor
                positions = range(apos + k, L) if right else range(apos - 1, -1, -1)
                for pos in positions:
                    nt = seq[pos]
                    follow = nt
                    if pos not in npos:
                        res4 = bloom.contains4(km, right)
                        solid = [x for x in range(4) if res4 >> x & 1]
                        if nt in solid:
                            if len(solid) == 2:
                                bifs.append(("bin", 0 if solid[0] == nt else 1))
                            elif len(solid) > 2:
                                bifs.append(("bif", nt))
                        elif solid:                                      # sequencing error: follow the first solid successor
                            errs.append(pos)
                            bifs.append(("bif", nt))
                            follow = solid[0]
                        else:
                            bifs.append(("bif", nt))
                    km = (((km << 2) | follow) & ((1 << (2 * k)) - 1)) if right else ((km >> 2) | (follow << (2 * (k - 1))))
            numeric(rc, M["num"], len(npos))
            p = 0
            for x in npos:
                numeric(rc, M["npos"], x - p); p = x
            numeric(rc, M["nerr"], len(errs))
            p = 0
            for x in sorted(errs):
                numeric(rc, M["errpos"], x - p); p = x
            for name, v in bifs:
                rc.encode(M[name], v)
        rc.flush()
        blocks.append(bytes(rc.out))
    drc.flush()
    return blocks, bytes(drc.out), order

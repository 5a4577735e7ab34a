"""A second, independent implementation of the header stream in pure Python, written from the
rules rather than from oracle/leon_oracle.c: fields by regular expression, records as tuples, then the range coder of
tests/py_leon.py.  tests/test_oracle_cpu.py checks the C oracle against it byte for byte (small inputs only)."""
import re

from py_leon import Model, RangeEncoder

END, END_MATCH, ASCII, NUMERIC, DELTA, DELTA_2, ZERO_ONLY, ZERO_AND_NUMERIC = 1, 2, 3, 4, 5, 6, 7, 8
_FIELD = re.compile(rb"[0-9A-Za-z]*(?:[^0-9A-Za-z]|$)", re.S)


def fields(h):
    """bytes -> list of fields: an alphanumeric run plus the one separator byte after it"""
    out, pos = [], 0
    while pos < len(h):
        m = _FIELD.match(h, pos)
        out.append(m.group(0))
        pos = m.end()
    return out


def kind(f):
    """-> ('num', value, sep) | ('zero', z, sep) | ('zeronum', z, value, sep) | ('ascii',); sep = b'' at the end of a header"""
    tok = f[:-1] if (f and not (48 <= f[-1] <= 57 or 65 <= f[-1] <= 90 or 97 <= f[-1] <= 122)) else f
    sep = f[len(tok):]
    if not tok or not tok.isdigit() or sep == b"\0":
        return ("ascii",)
    z = len(tok) - len(tok.lstrip(b"0"))
    if len(tok) == 1 or z == 0:
        return ("num", int(tok), sep) if len(tok) <= 18 else ("ascii",)
    if z == len(tok):
        return ("zero", z, sep)
    if len(tok) - z <= 18:
        return ("zeronum", z, int(tok[z:]), sep)
    return ("ascii",)


def records(cur, prev):
    """the records of one header against the previous one: list of tuples (type, ...)"""
    fc, fp = fields(cur), fields(prev)
    out = []
    for i, c in enumerate(fc):
        p = fp[i] if i < len(fp) else None
        if p == c:
            continue
        kc = kind(c)
        if kc[0] == "num":
            kp = kind(p) if p is not None else ("ascii",)
            if kp[0] == "num" and kp[2] == kc[2] and kp[1] != kc[1]:
                out.append((DELTA, i, kc[1] - kp[1]) if kc[1] > kp[1] else (DELTA_2, i, kp[1] - kc[1]))
            else:
                out.append((NUMERIC, i, kc[1], kc[2]))
        elif kc[0] == "zero":
            out.append((ZERO_ONLY, i, kc[1], kc[2]))
        elif kc[0] == "zeronum":
            out.append((ZERO_AND_NUMERIC, i, kc[1], kc[2], kc[3]))
        else:
            col = 0
            if p is not None:
                while col < len(c) and col < len(p) and c[col] == p[col]:
                    col += 1
            out.append((ASCII, i, col, c[col:]))
    out.append((END_MATCH,) if len(fc) >= len(fp) else (END, len(fc)))
    return out


def encode_block(headers, first):
    """list of bytes -> payload bytes"""
    rc = RangeEncoder()
    M = {"type": Model(9), "idx": Model(256), "col": Model(256), "size": Model(256), "ascii": Model(256), "zero": Model(256),
         "num": [Model(256) for _ in range(9)]}

    def numeric(v):
        bc = 1
        while bc < 8 and (v >> (8 * bc)):
            bc += 1
        rc.encode(M["num"][0], bc)
        for i in range(bc):
            rc.encode(M["num"][i + 1], (v >> (8 * i)) & 255)

    def count(m, x):
        if x < 255:
            rc.encode(M[m], x)
        else:
            rc.encode(M[m], 255)
            numeric(x - 255)

    def sep(s):
        rc.encode(M["ascii"], s[0] if s else 0)

    prev = first
    for h in headers:
        for r in records(h, prev):
            rc.encode(M["type"], r[0])
            if r[0] == END_MATCH:
                continue
            count("idx", r[1])
            if r[0] in (DELTA, DELTA_2):
                numeric(r[2])
            elif r[0] == NUMERIC:
                numeric(r[2]); sep(r[3])
            elif r[0] == ZERO_ONLY:
                count("zero", r[2]); sep(r[3])
            elif r[0] == ZERO_AND_NUMERIC:
                count("zero", r[2]); numeric(r[3]); sep(r[4])
            elif r[0] == ASCII:
                count("col", r[2]); count("size", len(r[3]))
                for b in r[3]:
                    rc.encode(M["ascii"], b)
        prev = h
    rc.flush()
    return bytes(rc.out)

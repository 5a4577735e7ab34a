"""-m gpu: the header stream through the C-ABI (leon_header_encode_batch: records on the device, one lane per header, then
k_rc_encode with the header model set) against the oracle's HeaderEncoder, block payloads byte for byte; and back
through the product's host decoder."""
import numpy as np
import pytest

import hdr_samples as H
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _ctx(rpb, **kw):
    import leon_amd
    return leon_amd.DnaEncodeContext(kmer_size=31, reads_per_block=rpb, bloom_tai=100000, **kw)


def _oracle_blocks(hs, rpb, first):
    return [O.header_encode_block(hs[b:b + rpb], first) for b in range(0, len(hs), rpb)]


@pytest.mark.parametrize("name,make,n,rpb", [("sra", H.sra, 12000, 5000), ("toy", H.toy_like, 3000, 700), ("nasty", H.nasty, 2000, 150),
                                             ("one_block", H.sra, 900, 50000), ("single", H.toy_like, 1, 10)])
def test_header_blocks_bit_exact(name, make, n, rpb):
    from leon_amd import capi
    hs = make(n)
    ctx = _ctx(rpb)
    blocks = ctx.header_encode_batch(hs)
    ref = _oracle_blocks(hs, rpb, hs[0])
    assert [b[0] for b in blocks] == list(range(len(ref)))
    assert [b[2] for b in blocks] == [len(hs[i:i + rpb]) for i in range(0, n, rpb)]
    for i, (b, r) in enumerate(zip(blocks, ref)):
        assert b[1] == r, "header block %d differs from the oracle" % i
    assert capi.host_header_decode_blocks(blocks, hs[0]) == hs
    ctx.close()


def test_header_stream_batches_shards_and_state():
    import leon_amd
    hs = H.sra(7000) + H.nasty(1300, seed=9)
    rpb, first = 500, b"some other first header 12"
    ref = _oracle_blocks(hs, rpb, first)
    # several batches of whole blocks, the last one partial
    ctx = _ctx(rpb)
    got = []
    for lo, hi in ((0, 2000), (2000, 2500), (2500, len(hs))):
        got += ctx.header_encode_batch(hs[lo:hi], first_header=first)
    assert [b[1] for b in got] == ref and [b[0] for b in got] == list(range(len(ref)))
    with pytest.raises(leon_amd.LeonDnaError) as e:                  # a partial block must be the last batch
        ctx.header_encode_batch(hs[:10], first_header=first)
    assert e.value.code == -4
    ctx.reset_stream()                                               # a new file on the same context
    assert [b[1] for b in ctx.header_encode_batch(hs, first_header=first)] == ref
    ctx.close()
    # ranks of a sharded job produce disjoint block ranges whose union is the stream
    union = []
    for r in range(3):
        c = _ctx(rpb)
        c.set_shard(r, 3)
        union += c.header_encode_batch(hs, first_header=first)
        c.close()
    assert sorted(b[0] for b in union) == list(range(len(ref)))
    assert [b[1] for b in sorted(union)] == ref
    # degenerate batches
    ctx = _ctx(rpb)
    assert ctx.header_encode_batch([]) == []
    assert [b[1] for b in ctx.header_encode_batch([b"", b"", b""])] == _oracle_blocks([b"", b"", b""], rpb, b"")
    ctx.close()

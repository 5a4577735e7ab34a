"""CPU tests of the N>1 path: the block partition and the order in which per-rank block tables merge, single-process
properties plus a world_size-2 gloo run in which every rank codes its block range with the product library (the quality
stream's host entry point; the GPU streams' N-process run is tests/test_gpu_multiprocess.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from leon_amd.shard import block_range, merge_block_tables


@pytest.mark.parametrize("n_blocks", [0, 1, 7, 200, 2000, 2001])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_block_range_partition(world, n_blocks):
    ranges = [block_range(r, world, n_blocks) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == n_blocks
    for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
        assert a1 == b0 and a0 <= a1
    sizes = [b - a for a, b in ranges]
    assert max(sizes) - min(sizes) <= 1


def test_merge_rejects_gaps():
    assert merge_block_tables([[(1, 5, 2)], [(0, 4, 2)]]) == [(0, 4, 2), (1, 5, 2)]
    with pytest.raises(ValueError):
        merge_block_tables([[(0, 1, 1)], [(2, 1, 1)]])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_blocks, rpb, n_reads, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b0, b1 = block_range(rank, world, n_blocks)
    # each rank codes ITS block range of the quality stream through the product library's host entry point (the one stream
    # of the format that needs no GPU: libleon_dna.so IS called here), with the global block ids a real rank's sink sees
    import hdr_samples as H
    import numpy as np
    from leon_amd import capi
    quals = H.fastq_quals(n_reads, 40, seed=12)                 # the same file on every rank
    mine = quals[b0 * rpb:b1 * rpb]
    off = np.zeros(len(mine) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(q) for q in mine])
    blocks = capi.host_qual_encode_blocks(b"".join(mine), off, rpb, n_threads=2, first_block_id=b0)
    table = [(bid, len(pay), nr) for bid, pay, nr in blocks]
    payloads = {bid: pay for bid, pay, nr in blocks}
    # the bloom broadcast of bench.py, on a byte tensor (gloo stands in for RCCL on CPU)
    bits = torch.arange(64, dtype=torch.uint8) if rank == 0 else torch.zeros(64, dtype=torch.uint8)
    dist.broadcast(bits, src=0)
    gathered = [None] * world
    dist.all_gather_object(gathered, table)
    all_payloads = [None] * world
    dist.all_gather_object(all_payloads, payloads)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                # the max-over-ranks timing reduction of bench.py
    if rank == 0:
        merged = merge_block_tables(gathered)
        # the union of the ranks' blocks decodes to the whole file, and equals what ONE process produces
        joined = {}
        for p_ in all_payloads:
            joined.update(p_)
        union = [(bid, joined[bid], nr) for bid, _, nr in merged]
        nbytes = [sum(len(x) for x in quals[b * rpb:(b + 1) * rpb]) for b in range(n_blocks)]
        whole_off = np.zeros(n_reads + 1, dtype=np.uint64)
        whole_off[1:] = np.cumsum([len(x) for x in quals])
        single = capi.host_qual_encode_blocks(b"".join(quals), whole_off, rpb, n_threads=2)
        ok = capi.host_qual_decode_blocks(union, nbytes) == quals and [b[1] for b in single] == [b[1] for b in union]
        q.put((merged, bits.tolist(), float(t.item()), ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_partition_and_merge():
    world, rpb, n_reads = 2, 50, 530
    n_blocks = (n_reads + rpb - 1) // rpb
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_blocks, rpb, n_reads, q)) for r in range(world)]
    for p in procs:
        p.start()
    merged, bits, tmax, union_ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [m[0] for m in merged] == list(range(n_blocks))
    assert sum(m[2] for m in merged) == n_reads
    assert bits == list(range(64)) and tmax == 2.0
    assert union_ok, "the union of the ranks' quality blocks is not the single-process stream"

"""CPU tests of the N>1 path: the block partition and the order in which per-rank block tables merge,
single-process properties plus a world_size-2 gloo run."""
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
    # each rank "encodes" its block range: here only the table a real rank would hand to its sink
    table = [(b, 100 + b, min(rpb, n_reads - b * rpb)) for b in range(b0, b1)]
    # the bloom broadcast of bench.py, on a byte tensor (gloo stands in for RCCL on CPU)
    bits = torch.arange(64, dtype=torch.uint8) if rank == 0 else torch.zeros(64, dtype=torch.uint8)
    dist.broadcast(bits, src=0)
    gathered = [None] * world
    dist.all_gather_object(gathered, table)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                # the max-over-ranks timing reduction of bench.py
    if rank == 0:
        merged = merge_block_tables(gathered)
        q.put((merged, bits.tolist(), float(t.item())))
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
    merged, bits, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [m[0] for m in merged] == list(range(n_blocks))
    assert sum(m[2] for m in merged) == n_reads
    assert bits == list(range(64)) and tmax == 2.0

"""Partition of a read stream over ranks: contiguous ranges of whole read blocks (Leon::READ_PER_BLOCK reads),
the unit whose models and payload are independent (AbstractDnaCoder::startBlock [RECALLED])."""


def block_range(rank, world, n_blocks):
    """blocks [b0, b1) of `rank`: contiguous, disjoint, covering, sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world) or n_blocks < 0:
        raise ValueError("bad partition arguments")
    q, r = divmod(n_blocks, world)
    b0 = rank * q + min(rank, r)
    return b0, b0 + q + (1 if rank < r else 0)


def merge_block_tables(tables):
    """tables: per-rank lists of (block_id, size, n_reads) -> one table in block order; checks it is gap-free."""
    merged = sorted(t for table in tables for t in table)
    for i, t in enumerate(merged):
        if t[0] != i:
            raise ValueError("block table has a gap or a duplicate at block %d" % i)
    return merged

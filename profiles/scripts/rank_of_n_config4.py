"""BASELINE configuration #4 (100 M x 150 bp, k = 31, sharded over N GPUs) from the seat of single ranks, on one GPU: what a rank of
an N-GPU run does under leon_dna_set_shard -- ALL reads resolved (replicated), its contiguous share of the blocks walked and coded --
timed stage by stage for worlds 1, 2, 4, 8 and the first, a middle and the last rank of each.  The 8-GPU node itself is out of this
pipeline's reach; these are its per-rank device times, measured, not projected.
    python profiles/scripts/rank_of_n_config4.py [reads]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench        # noqa: E402
import leon_amd     # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    K, L, RPB = 31, 150, 50_000
    dev = torch.device("cuda", 0)
    G = N * L // 30
    genome = bench.gen_genome(G, dev)
    ctx = leon_amd.DnaEncodeContext(kmer_size=K, reads_per_block=RPB, bloom_tai=(G - K + 1) * 12)
    for lo in range(0, G - K + 1, 1 << 26):
        km = bench.genome_kmers_chunk(genome, lo, min(G - K + 1, lo + (1 << 26)), K)
        torch.cuda.synchronize()
        ctx.bloom_insert_device(km.data_ptr(), km.shape[0])
        del km
    reads = torch.empty((N, L), dtype=torch.uint8, device=dev)
    for c in range(N // bench.CHUNK):
        reads[c * bench.CHUNK:(c + 1) * bench.CHUNK] = bench.gen_reads_chunk(genome, c, bench.CHUNK, 0.01, dev, L=L)
    off = (torch.arange(N + 1, dtype=torch.int64, device=dev) * L).contiguous()
    torch.cuda.synchronize()
    ctx.reserve(N, N * L)
    for world in (1, 2, 4, 8):
        worst = None
        for rank in sorted({0, world // 2, world - 1}):
            ctx.reset_stream()
            ctx.set_shard(rank, world)
            best = None
            for _ in range(2):                               # the second pass is the warm one
                ctx.reset_stream()
                got = ctx.encode_batch_device(reads.data_ptr(), off.data_ptr(), N)
                ctx.finish()
                st = ctx.stats()
                if best is None or st["ms_total"] < best["ms_total"]:
                    best = st
            row = {"world": world, "rank": rank, "blocks": len(got), "payload_bytes": sum(len(g[1]) for g in got)}
            row.update({k: round(v, 2) for k, v in best.items() if k.startswith("ms_")})
            print(json.dumps(row), flush=True)
            if worst is None or best["ms_total"] > worst["ms_total"]:
                worst = best
        print(json.dumps({"world": world, "device_ms_slowest_rank_seen": round(worst["ms_total"], 2),
                          "value_device_only_MBps": round(N * L / 1e6 / (worst["ms_total"] * 1e-3), 1),
                          "replicated_ms": round(worst["ms_pack"] + worst["ms_resolve"], 2)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()

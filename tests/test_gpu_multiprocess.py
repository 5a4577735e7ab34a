"""-m gpu: the REAL N-process path of bench.py (BASELINE.json configuration #4's code: leon_dna_set_shard on every rank,
the bloom built by rank 0 and broadcast, every rank resolving all reads and walking / coding its own block range, the
merged block table) launched both ways the driver may launch it -- `python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N`, and plain `python bench.py --gpus N` (which starts the former as a child before it touches the GPU) --
on the one GPU of the test box, the ranks sharing device 0 (at most 4 of them: the box allows 6 processes on its card),
collectives over gloo (LEON_BENCH_BACKEND=gloo: RCCL needs one device per rank, which only the driver's 8-GPU node has).
The 8-way split itself runs in one process in test_gpu_parity.py::test_sharded_contexts_reproduce_the_single_stream.
The ranks are fresh child processes of torch.distributed.run, itself a child that never touches the GPU.
Asserts: union of the ranks' blocks == the single-process stream (checksum of block checksums), same dictionary stream."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READS = 2_000_000          # 40 read blocks


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(world, extra_env=None, extra_args=(), launcher=True):
    env = dict(os.environ)
    env.update({"LEON_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra_env or {})
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(v, None)
    args = ["bench.py", "--gpus", str(world), "--steps", "1", "--warmup", "0", "--reads", str(READS), "--cpu-sample", "0", "--e2e-reads", "0"] + list(extra_args)
    if world == 1 or not launcher:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, "bench.py world %d failed:\n%s\n%s" % (world, p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 must print exactly one JSON line"
    return json.loads(lines[0])


def test_two_and_three_process_runs_reproduce_the_single_process_stream():
    one = _bench(1)
    assert one["n_gpus"] == 1 and one["verify"]["n_blocks"] == READS // 50000
    for world, launcher, extra in ((2, True, ()), (3, False, ()), (4, True, ("--batch-reads", "700000"))):     # 4 ranks, three batches of 14 + 14 + 12 blocks
        many = _bench(world, extra_args=extra, launcher=launcher)
        assert many["n_gpus"] == world and many["scaling"] == "strong"
        assert many["verify"]["blocks_per_rank"] and len(many["verify"]["blocks_per_rank"]) == world
        assert min(many["verify"]["blocks_per_rank"]) > 0
        assert many["verify"]["blocks_sha256"] == one["verify"]["blocks_sha256"]
        assert many["verify"]["dict_sha256"] == one["verify"]["dict_sha256"]
        assert many["verify"]["n_anchors"] == one["verify"]["n_anchors"]
        assert many["config"]["bloom_bytes"] == one["config"]["bloom_bytes"] and many["config"]["bloom_bcast_ms"] > 0
        for key in ("device_ms_max_over_ranks", "host_chain_ms", "value_device_only", "cold_first_step_ms", "bloom_bcast_ms"):
            assert many[key] > 0
        # what the 1 -> 8 curve will be read from: every rank's device stages, gathered
        assert [r["rank"] for r in many["per_rank"]] == list(range(world)) and many["collective_backend"] == "gloo" and many["rccl_ranks"] is None
        assert [r["blocks"] for r in many["per_rank"]] == many["verify"]["blocks_per_rank"]
        assert all(r["stages_ms"]["ms_resolve"] > 0 and r["stages_ms"]["ms_walk"] > 0 and r["device_ms"] > 0 for r in many["per_rank"])
        assert many["per_rank"][0]["chain_ms"] > 0 and all(r["chain_ms"] == 0 for r in many["per_rank"][1:])      # rank 0 alone codes the dictionary


def test_default_bench_line_fills_every_key():
    """the driver's `python bench.py` line: value beside the H2D-inclusive figure, verify, decode, end_to_end, roofline and
    cpu_baseline -- none of them null at N = 1 (small sizes here; the keys and their checks are what is tested)"""
    env = dict(os.environ)
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(v, None)
    cmd = [sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--reads", "1000000", "--batch-reads", "400000", "--cpu-sample", "100000",
           "--e2e-reads", "200000", "--other-configs", "200000:31:150,100000:63:250"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    for key in ("value", "value_hbm_resident", "value_h2d_inclusive", "pcie_inclusive", "decode", "verify", "streams", "end_to_end", "roofline", "cpu_baseline"):
        assert line[key] is not None, key
    assert line["config"]["batches"] == 3 and line["value"] == line["value_hbm_resident"]
    assert line["decode"]["equals_input"] is True
    assert line["end_to_end"]["identical"] is True and line["end_to_end"]["compress_lossless_rc"] == 0
    assert line["verify"]["n_blocks"] == 20
    assert line["streams"]["header_decode"]["equal_input"] is True and line["streams"]["qual_smooth"]["MBps"] > 0
    assert line["streams"]["qual_deflate"]["inflates_to_input"] is True and 0 < line["streams"]["qual_deflate"]["ratio"] < 0.6
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0
    # the CPU restatement on every CPU the process may use, and as one shared stream on one core; both labelled a port
    assert line["cpu_baseline"]["cores"] == line["host"]["cpus_usable"] and line["cpu_baseline"]["kind"] == "port"
    assert line["cpu_baseline"]["one_stream"]["cores"] == 1 and line["cpu_baseline"]["one_stream"]["value"] > 0
    # a file with real-genome structure in genome-position order, and the other single-GPU configurations: driven by the same run
    st = line["structured"]
    assert st["decode_equals_input"] is True and st["order"] == "sorted" and st["resolve"]["reads_left_to_the_sequential_pass"] > 0
    assert sorted(line["other_configs"]) == ["100000_x_250bp_k63", "200000_x_150bp_k31"]
    assert all(v["value"] > 0 and v["roofline"]["frac"] > 0 and v["steps"] == 5 for v in line["other_configs"].values())
    assert "UNPINNED" in line["parity"]
    assert line["host"]["cpus_allowed"] >= 1 and line["host"]["chain_ns_per_symbol"] > 0
    # the same file as ONE batch: same bytes
    one = _bench(1, extra_args=("--reads", "1000000"))
    assert one["verify"]["blocks_sha256"] == line["verify"]["blocks_sha256"] and one["verify"]["dict_sha256"] == line["verify"]["dict_sha256"]


def test_a_rank_that_never_arrives_ends_the_job_by_name():
    """No collective may sit until the driver's limit: rank 1 never reaches the bloom broadcast (test hook), rank 0's watch names the
    collective it is stuck in after LEON_BENCH_COLL_TIMEOUT seconds and exits non-zero, the launcher tears the job down."""
    env = dict(os.environ, LEON_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", LEON_BENCH_TEST_STALL_RANK="1", LEON_BENCH_COLL_TIMEOUT="8")
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(v, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--reads", "200000", "--cpu-sample", "0", "--quick"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    # (the process group's own timeout raises inside the collective at the same moment the watch would fire: either way the collective is named)
    assert "bench_error" in p.stderr and "barrier (before the bloom broadcast)" in p.stderr and ("did not complete" in p.stderr or "failed" in p.stderr), p.stderr[-3000:]
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]          # and no line pretending to be a result


def test_rccl_path_with_one_rank():
    """RCCL itself (backend nccl) needs one device per rank, so more than one rank cannot run on the test box's one GPU; but the
    N > 1 CODE can: one rank under torch.distributed.run with LEON_BENCH_FORCE_DIST=1 goes through init_process_group("nccl",
    device_id=...), the device-to-device broadcast of the bloom, the reductions, the gather of the block tables and the barriers
    -- every RCCL call the 8-GPU run makes, with a world of one.  Same bytes as the plain run."""
    # (NCCL_DEBUG=VERSION: RCCL then prints its version banner on STDOUT when the communicator starts, as one box of the pool did unasked;
    # the driver takes one JSON line from there, so bench.py keeps everything else off it)
    env = dict(os.environ, LEON_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="VERSION")
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LEON_BENCH_BACKEND"):
        env.pop(v, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--steps", "1", "--warmup", "0", "--reads", str(READS),
           "--cpu-sample", "0", "--quick", "--verify"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "RCCL version" not in p.stdout and "NCCL version" not in p.stdout, p.stdout[:600]
    assert len([l for l in p.stdout.splitlines() if l.startswith("{")]) == 1
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["bloom_bcast_ms"] > 0            # the broadcast ran (through RCCL)
    plain = _bench(1, extra_args=("--quick", "--verify"))
    assert line["verify"]["blocks_sha256"] == plain["verify"]["blocks_sha256"] and line["verify"]["dict_sha256"] == plain["verify"]["dict_sha256"]


def test_exchange_callbacks_over_rccl_with_one_rank():
    """bench.py's two callbacks for leon_dna_set_exchange / leon_dna_set_gather in their RCCL form (all_to_all_single,
    all_gather_into_tensor on device tensors): the library only calls them with world > 1, which RCCL cannot have on one device, and
    the multi-process tests go over gloo -- so the branch the 8-GPU run takes is called here directly, on a process group of ONE:
    what a rank sends itself comes back, a gathered buffer keeps its one part."""
    script = r"""
import ctypes, os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
import bench, leon_amd
from leon_amd import capi
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % sys.argv[1], rank=0, world_size=1, device_id=dev)
leon_amd.load_library()
words = torch.arange(1000, dtype=torch.int64, device=dev) * 7 + 3
W = bench.Watch(0, dev)
xch = bench.make_exchange(dist, dev, "nccl", 0, 1, capi, W)
ptr, total = xch(words.data_ptr(), [1000])
assert total == 1000
back = torch.empty(1000, dtype=torch.int64, device=dev)
capi.device_copy(back.data_ptr(), ptr, 8000)
assert torch.equal(back, words)
ptr, total = xch(0, [0])                                   # a rank with nothing to send
assert total == 0
buf = torch.arange(4096, dtype=torch.int64, device=dev)
want = buf.clone()
bench.make_gather(dist, dev, "nccl", 0, 1, capi, W)(buf.data_ptr(), 4096 * 8, 1)
assert W.summary()["all_to_all_single (walk exchange: the event words)"]["calls"] == 2 and "all_gather_into_tensor (a window's look-ups)" in W.summary()
torch.cuda.synchronize()
assert torch.equal(buf, want)
dist.barrier()
dist.destroy_process_group()
print("callbacks ok")
"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LEON_BENCH_BACKEND"):
        env.pop(v, None)
    p = subprocess.run([sys.executable, "-c", script, str(_free_port())], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "callbacks ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_bench_as_rank_r_of_8_seats_add_up_to_the_single_stream():
    """LEON_BENCH_AS_RANK=r:N: ONE process takes the seat of rank r of an N-rank job (leon_dna_set_shard(r, N), every collective of
    the N-rank code on an RCCL process group of one).  The eight seats of an 8-rank job, one after the other on the one GPU: every seat
    codes exactly its block range, rank 0 alone the dictionary, and the eight seats' blocks are the single-process stream's (each seat's
    checksum over its own blocks == the same checksum over those blocks of the whole stream is what `verify` cannot say from one seat;
    the union over the seats can: block counts add up, ranges are contiguous and disjoint, the dictionary is rank 0's)."""
    import hashlib
    from leon_amd.shard import block_range
    n_blocks = READS // 50000
    one = _bench(1, extra_args=("--quick", "--verify"))
    env0 = {"LEON_BENCH_BACKEND": "nccl"}
    seats = []
    for r in range(8):
        line = _bench(1, extra_env=dict(env0, LEON_BENCH_AS_RANK="%d:8" % r), extra_args=("--quick", "--verify"))
        assert line["as_rank"] == "%d:8" % r and line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["collective_backend"] == "nccl"
        lo, hi = block_range(r, 8, n_blocks)
        v = line["verify"]
        assert (v["n_blocks"], v["first_block"], v["last_block"]) == (hi - lo, lo, hi - 1), (r, v)
        assert line["per_rank"][0]["rank"] == r and line["per_rank"][0]["blocks"] == hi - lo
        assert v["n_anchors"] == one["verify"]["n_anchors"]
        assert (v["dict_sha256"] == one["verify"]["dict_sha256"]) if r == 0 else (v["dict_sha256"] == hashlib.sha256(b"").hexdigest())
        assert (line["host_chain_ms"] > 0) == (r == 0)
        assert line["bloom_bcast_ms"] > 0
        seats.append(v["n_blocks"])
    assert sum(seats) == n_blocks == one["verify"]["n_blocks"]


def test_configuration_5_shape_every_seat_of_8_at_reduced_size():
    """BASELINE configuration #5's shape -- 250 bp reads, k = 63 (two-word k-mers), FIVE batches per file, 8 ranks, the walk divided by anchor and
    the window look-ups divided too -- from EVERY seat of the 8-rank job at a size one GPU runs in seconds (2 M reads: 40 blocks, batches of 8
    blocks, so every rank codes one block of every batch).  At full size only rank 0's seat has ever run (tests/test_gpu_fullsize.py::
    test_configuration_5_as_rank_0_of_8_sees_it); here the eight seats' blocks add up to the one-GPU stream's, rank 0 alone carries the dictionary."""
    import hashlib
    from leon_amd.shard import block_range
    shape = {"LEON_BENCH_K": "63", "LEON_BENCH_L": "250"}
    args = ("--quick", "--verify", "--batch-reads", "400000")
    n_blocks, per_batch = READS // 50000, 8
    one = _bench(1, extra_env=shape, extra_args=args)
    assert one["config"]["batches"] == 5 and one["config"]["kmer_size"] == 63 and one["verify"]["n_blocks"] == n_blocks
    total = 0
    for r in range(8):
        line = _bench(1, extra_env=dict(shape, LEON_BENCH_BACKEND="nccl", LEON_BENCH_AS_RANK="%d:8" % r), extra_args=args)
        lo, hi = block_range(r, 8, per_batch)
        v = line["verify"]
        assert line["as_rank"] == "%d:8" % r and line["config"]["walk_by"] == "anchor" and line["config"]["batches"] == 5
        assert v["n_blocks"] == 5 * (hi - lo) and v["first_block"] == lo and v["last_block"] == 4 * per_batch + hi - 1, (r, v)
        assert v["n_anchors"] == one["verify"]["n_anchors"]
        assert (v["dict_sha256"] == one["verify"]["dict_sha256"]) if r == 0 else (v["dict_sha256"] == hashlib.sha256(b"").hexdigest())
        assert line["per_rank"][0]["stages_ms"]["ms_emulated"] > 0 and line["per_rank"][0]["stages_ms"]["ms_walk"] > 0
        # the seat's blocks, digest by digest, are the one-GPU stream's
        assert v["block_digests"] and all(one["verify"]["block_digests"][b] == d for b, d in v["block_digests"].items()), r
        total += v["n_blocks"]
    assert total == n_blocks

"""-m gpu: the REAL N-process path of bench.py (BASELINE.json configuration #4's code: leon_dna_set_shard on every rank,
the bloom built by rank 0 and broadcast, every rank resolving all reads and walking / coding its own block range, the
merged block table) launched the way the driver launches it -- `python -m torch.distributed.run --nproc-per-node 2
bench.py --gpus 2` -- on the one GPU of the test box, both ranks sharing device 0, collectives over gloo
(LEON_BENCH_BACKEND=gloo: RCCL needs one device per rank, which only the driver's 8-GPU node has).
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


def _bench(world, extra_env=None):
    env = dict(os.environ)
    env.update({"LEON_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra_env or {})
    args = ["bench.py", "--gpus", str(world), "--steps", "1", "--warmup", "0", "--reads", str(READS), "--cpu-sample", "0", "--verify"]
    if world == 1:
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
    for world in (2, 3):
        many = _bench(world)
        assert many["n_gpus"] == world and many["scaling"] == "strong"
        assert many["verify"]["blocks_per_rank"] and len(many["verify"]["blocks_per_rank"]) == world
        assert min(many["verify"]["blocks_per_rank"]) > 0
        assert many["verify"]["blocks_sha256"] == one["verify"]["blocks_sha256"]
        assert many["verify"]["dict_sha256"] == one["verify"]["dict_sha256"]
        assert many["verify"]["n_anchors"] == one["verify"]["n_anchors"]
        assert many["config"]["bloom_bytes"] == one["config"]["bloom_bytes"] and many["config"]["bloom_bcast_ms"] > 0
        for key in ("device_ms_max_over_ranks", "host_chain_ms", "value_device_only", "cold_first_step_ms"):
            assert many[key] > 0

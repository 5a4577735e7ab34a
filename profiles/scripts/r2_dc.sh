#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decoder or crafted or degenerate or roundtrip" > gpurun_out/dc_pytest.log 2>&1 || { tail -20 gpurun_out/dc_pytest.log; exit 1; }
tail -2 gpurun_out/dc_pytest.log
export LEON_TRACE_DECODE=1
timeout -k 10 200 python bench.py --steps 1 --warmup 0 --decode --cpu-sample 0 --reads 10000000 > gpurun_out/dc_10M.json 2> gpurun_out/dc_10M.err
grep "leon decode" gpurun_out/dc_10M.err | grep -E "in its fields|k_decode" | head -3

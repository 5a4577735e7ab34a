#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decoder or crafted or degenerate or roundtrip" > gpurun_out/dc_pytest.log 2>&1 || { tail -20 gpurun_out/dc_pytest.log; exit 1; }
tail -2 gpurun_out/dc_pytest.log
export LEON_TRACE_DECODE=1
for steps in 128 192; do
echo "== k63 prewalk $steps"
LEON_DC_PREWALK=$steps LEON_BENCH_K=63 LEON_BENCH_L=250 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --decode --cpu-sample 0 --reads 20000000 > gpurun_out/dc_k63.json 2> gpurun_out/dc_k63.err
grep "leon decode" gpurun_out/dc_k63.err | sed -n 4,7p
done

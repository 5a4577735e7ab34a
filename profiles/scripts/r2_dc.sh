#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
export LEON_TRACE_DECODE=1
for steps in 64 128; do
echo "== 100M prewalk budget $steps"
LEON_DC_PREWALK=$steps timeout -k 10 400 python bench.py --steps 1 --warmup 0 --decode --cpu-sample 0 > gpurun_out/dc_100M.json 2> gpurun_out/dc_100M.err
grep "leon decode" gpurun_out/dc_100M.err | grep -E "per read|k_decode|walks from" | head -5
done

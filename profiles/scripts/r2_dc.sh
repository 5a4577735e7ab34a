#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
export LEON_TRACE_DECODE=1
timeout -k 10 900 python bench.py --steps 1 --warmup 0 --decode --cpu-sample 0 > gpurun_out/dc_100M.json 2> gpurun_out/dc_100M.err
python -c "import json;d=json.load(open('gpurun_out/dc_100M.json'));print(d.get('decode'))"
grep "leon decode" gpurun_out/dc_100M.err
timeout -k 10 900 python bench.py --steps 1 --warmup 0 --decode --cpu-sample 0 --reads 10000000 > gpurun_out/dc_10M.json 2> gpurun_out/dc_10M.err
python -c "import json;d=json.load(open('gpurun_out/dc_10M.json'));print(d.get('decode'))"
grep "leon decode" gpurun_out/dc_10M.err

#!/bin/bash
# the coder's step without its third product; k_lookup_cand's grid; parity of the build
set -o pipefail
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_streams.py -x -q > gpurun_out/r4_mm3_parity.log 2>&1; rc=$?; echo "parity rc $rc"; tail -3 gpurun_out/r4_mm3_parity.log
[ $rc -eq 0 ] || exit 1
SWEEP_VARIANTS='[{},{"LEON_LOOKUP_BLOCKS_PER_CU":8},{"LEON_LOOKUP_BLOCKS_PER_CU":32},{"LEON_LOOKUP_BLOCKS_PER_CU":64},{"LEON_LOOKUP_BLOCKS_PER_CU":256},{}]' \
  timeout -k 10 900 python profiles/scripts/resolve_sweep.py > gpurun_out/r4_mm3_sweep.txt 2> gpurun_out/r4_mm3_sweep.err; echo "sweep rc $?"
python - <<'PY'
import json
for l in open('gpurun_out/r4_mm3_sweep.txt'):
    j=json.loads(l); print(j['variant'], j['same_bytes'], {k:j['ms'][k] for k in ('ms_resolve','ms_walk','ms_rangecoder','ms_total')})
PY

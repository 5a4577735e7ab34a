#!/bin/bash
# window sizes again, with the cheaper look-ups; the new every-k test
set -o pipefail
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "small_windows or two_word" > gpurun_out/r4_mm4_parity.log 2>&1; rc=$?; echo "parity rc $rc"; tail -3 gpurun_out/r4_mm4_parity.log
[ $rc -eq 0 ] || exit 1
SWEEP_VARIANTS='[{},{"LEON_RESOLVE_WINDOW":1572864},{"LEON_RESOLVE_WINDOW":2097152},{"LEON_RESOLVE_WINDOW":3145728},{"LEON_RESOLVE_WINDOW":4194304}]' \
  timeout -k 10 900 python profiles/scripts/resolve_sweep.py > gpurun_out/r4_mm4_sweep.txt 2> gpurun_out/r4_mm4_sweep.err; echo "sweep rc $?"
python - <<'PY'
import json
for l in open('gpurun_out/r4_mm4_sweep.txt'):
    j=json.loads(l); print(j['variant'], j['same_bytes'], j['rounds'], j['windows'], {k:j['ms'][k] for k in ('ms_resolve','ms_walk','ms_rangecoder','ms_total')})
PY

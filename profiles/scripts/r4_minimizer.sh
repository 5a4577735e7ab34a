#!/bin/bash
# the final keys' filter addressed by minimizer: parity, then the resolution stage under several filter sizes (same bytes checked)
set -o pipefail
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r4_minimizer_parity.log 2>&1; rc=$?; echo "parity rc $rc"; tail -4 gpurun_out/r4_minimizer_parity.log
[ $rc -eq 0 ] || exit 1
SWEEP_VARIANTS='[{"LEON_FBITS_LOG2":25},{"LEON_FBITS_LOG2":26},{"LEON_FBITS_LOG2":27},{"LEON_FBITS_LOG2":28},{"LEON_FBITS_LOG2":29},{"LEON_FBITS_LOG2":30},{"LEON_FBITS_LOG2":31}]' \
  timeout -k 10 900 python profiles/scripts/resolve_sweep.py > gpurun_out/r4_minimizer_sweep.txt 2> gpurun_out/r4_minimizer_sweep.err; echo "sweep rc $?"
cat gpurun_out/r4_minimizer_sweep.txt
LEON_TRACE_RESOLVE=1 timeout -k 10 300 python bench.py --reads 20000000 --quick --steps 1 --warmup 1 --cpu-sample 0 2>&1 >/dev/null | grep "leon resolve" | tail -8

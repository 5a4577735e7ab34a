#!/bin/bash
# the resolution stage with its rounds launched ahead of their counts: parity, then the bench; the chain's take-apart beside it
set -o pipefail
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
CXX=/opt/rocm/lib/llvm/bin/clang++
$CXX -O2 -std=c++17 -mbmi2 -o /tmp/skel profiles/scripts/chain_ab/skel.cpp && /tmp/skel > gpurun_out/r4_chain_take_apart.txt 2>&1
$CXX -O2 -std=c++17 -mbmi2 -o /tmp/ab_spec profiles/scripts/chain_ab/ab_spec.cpp -lpthread
{ for rep in 1 2 3; do /tmp/ab_spec 4000000 0 | grep -v "from DRAM"; done; /tmp/ab_spec 4000000 0 | grep "from DRAM"; } > gpurun_out/r4_chain_spec_ab4.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r4_resolve_parity.log 2>&1; echo "parity rc $?"; tail -4 gpurun_out/r4_resolve_parity.log
timeout -k 10 600 python bench.py --quick --steps 5 --warmup 1 > gpurun_out/r4_bench_quick_resolve.json 2> gpurun_out/r4_bench_quick_resolve.err; echo "bench rc $?"
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r4_bench_quick_resolve.json') if l.startswith('{')][-1])
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'device', j['device_ms_max_over_ranks'], 'chain', j['host_chain_ms'])
print(j['stages_ms_rank0']); print(j['rank0'])
PY
cat gpurun_out/r4_chain_take_apart.txt; grep -v "^PREDICTED FLAG," gpurun_out/r4_chain_spec_ab4.txt

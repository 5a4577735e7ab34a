#!/bin/bash
# the carry-flag variant of the dictionary chain on the box's host; the parity of the build as it stands; configuration #2 again
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
CXX=/opt/rocm/lib/llvm/bin/clang++
$CXX -O2 -std=c++17 -mbmi2 -o /tmp/ab_spec profiles/scripts/chain_ab/ab_spec.cpp -lpthread
{ for rep in 1 2 3; do /tmp/ab_spec 4000000 0 | grep "records in cache"; done; /tmp/ab_spec 4000000 0 | grep "from DRAM" | grep -v "^PREDICTED FLAG,"; } > gpurun_out/r4_chain_carry_ab.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streams.py -x -q > gpurun_out/r4_carry_parity.log 2>&1; echo "parity rc $?"; tail -3 gpurun_out/r4_carry_parity.log
timeout -k 10 300 python bench.py --reads 10000000 --quick --steps 6 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('config2 value', j['value'], 'ms_per_step', j['ms_per_step'], 'rc', round(j['stages_ms_rank0']['ms_rangecoder'],1))"
grep -v "PREDICTED FLAG, records in" gpurun_out/r4_chain_carry_ab.txt

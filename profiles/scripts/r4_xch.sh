#!/bin/bash
# the walk divided by anchor: parity tests, then seats of configuration #4 timed both ways; the chain's variants on the host beside it
set -o pipefail
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
CXX=/opt/rocm/lib/llvm/bin/clang++
g++ -O2 -std=c++17 -mbmi2 -o /tmp/cost profiles/scripts/chain_ab/cost.cpp && /tmp/cost > gpurun_out/r4_host_instruction_costs.txt 2>&1
$CXX -O2 -std=c++17 -mbmi2 -o /tmp/ab_spec profiles/scripts/chain_ab/ab_spec.cpp -lpthread
{ grep -m1 "model name" /proc/cpuinfo; for rep in 1 2 3; do /tmp/ab_spec 4000000 0 | grep -v "from DRAM"; done; /tmp/ab_spec 4000000 0 | grep "from DRAM"; } > gpurun_out/r4_chain_spec_ab3.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -k "sharded or divided" -x -q > gpurun_out/r4_xch_parity.log 2>&1; echo "parity rc $?"; tail -5 gpurun_out/r4_xch_parity.log
for seat in 0 3 7; do for by in anchor; do
  LEON_BENCH_AS_RANK=$seat:8 timeout -k 10 600 python bench.py --quick --steps 3 --warmup 1 --walk-by $by > gpurun_out/r4_seat_${seat}_of_8_${by}.json 2> gpurun_out/r4_seat_${seat}_of_8_${by}.err; echo "seat $seat $by rc $?"
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_seat_*_of_8_*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
        r=j['per_rank'][0]; s=r['stages_ms']
        print(f, 'device_ms', r['device_ms'], {k: round(v,1) for k,v in s.items() if k in ('ms_pack','ms_resolve','ms_sort','ms_walk','ms_exchange','ms_emulated','ms_symbols','ms_rangecoder','ms_total')})
    except Exception as e: print(f, 'ERR', e)
PY
cat gpurun_out/r4_host_instruction_costs.txt; cat gpurun_out/r4_chain_spec_ab3.txt

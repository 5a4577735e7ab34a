#!/bin/bash
# the host chains of configuration #2 under chunk / thread counts; the resolution stage's probes by outcome
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for ch in 4 8 16; do for th in 12 15; do
  echo "chunks $ch threads $th"
  LEON_TRACE_RC_HOST=1 LEON_RC_HOST_CHUNKS=$ch LEON_RC_HOST_THREADS=$th timeout -k 10 300 python bench.py --reads 10000000 --quick --steps 6 --warmup 2 --cpu-sample 0 2> gpurun_out/r4_c2_${ch}_${th}.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'rc', round(j['stages_ms_rank0']['ms_rangecoder'],1))"
  grep "leon rc host" gpurun_out/r4_c2_${ch}_${th}.err | tail -1
done; done > gpurun_out/r4_hostchains_sweep.txt 2>&1
LEON_TRACE_RESOLVE=1 timeout -k 10 300 python bench.py --quick --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2> gpurun_out/r4_resolve_trace.txt
LEON_BENCH_K=63 LEON_BENCH_L=250 LEON_TRACE_RESOLVE=1 timeout -k 10 300 python bench.py --reads 20000000 --quick --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>> gpurun_out/r4_resolve_trace.txt
cat gpurun_out/r4_hostchains_sweep.txt; grep "leon resolve" gpurun_out/r4_resolve_trace.txt

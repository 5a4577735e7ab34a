#!/bin/bash
# small launches' chains on host cores: parity both ways, configuration #2 both ways, the 100 M bench (its coder waves after the refactoring)
set -o pipefail
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_streams.py -x -q > gpurun_out/r4_hostchains_parity.log 2>&1; echo "parity rc $?"; tail -6 gpurun_out/r4_hostchains_parity.log
for mode in 400 0; do
  LEON_RC_HOST_BLOCKS=$mode timeout -k 10 600 python bench.py --reads 10000000 --quick --steps 10 --warmup 2 > gpurun_out/r4_bench_config2_hostblocks_$mode.json 2> gpurun_out/r4_bench_config2_hostblocks_$mode.err; echo "config2 mode $mode rc $?"
done
LEON_BENCH_K=63 LEON_BENCH_L=250 timeout -k 10 600 python bench.py --reads 20000000 --quick --steps 5 --warmup 1 > gpurun_out/r4_bench_k63_20M.json 2> gpurun_out/r4_bench_k63_20M.err; echo "k63 rc $?"
timeout -k 10 600 python bench.py --quick --steps 5 --warmup 1 > gpurun_out/r4_bench_quick_100M_b.json 2> gpurun_out/r4_bench_quick_100M_b.err; echo "100M rc $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_bench_config2_hostblocks_*.json'))+['gpurun_out/r4_bench_k63_20M.json','gpurun_out/r4_bench_quick_100M_b.json']:
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, 'value', j['value'], 'ms_per_step', j['ms_per_step'], 'device', j['device_ms_max_over_ranks'], 'chain', j['host_chain_ms'], {k: round(v,1) for k,v in j['stages_ms_rank0'].items() if k in ('ms_pack','ms_resolve','ms_sort','ms_walk','ms_symbols','ms_rangecoder','ms_d2h')})
    except Exception as e: print(f, 'ERR', e)
PY

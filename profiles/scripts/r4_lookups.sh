#!/bin/bash
# the window look-ups divided among the ranks (leon_dna_set_gather): parity (emulated, threads with a real exchange, processes over
# gloo), then seats of configuration #4 timed with and without
set -o pipefail
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multiprocess.py -x -q > gpurun_out/r4_lookups_parity.log 2>&1; rc=$?; echo "parity rc $rc"; tail -5 gpurun_out/r4_lookups_parity.log
[ $rc -eq 0 ] || exit 1
for seat in 0 3 7; do
  LEON_BENCH_AS_RANK=$seat:8 timeout -k 10 600 python bench.py --quick --steps 3 --warmup 1 > gpurun_out/r4_seat_${seat}_of_8_lookups.json 2> gpurun_out/r4_seat_${seat}_of_8_lookups.err; echo "seat $seat rc $?"
done
LEON_XCH_LOOKUPS=0 LEON_BENCH_AS_RANK=3:8 timeout -k 10 600 python bench.py --quick --steps 3 --warmup 1 > gpurun_out/r4_seat_3_of_8_nolookups.json 2> gpurun_out/r4_seat_3_of_8_nolookups.err; echo "seat 3 (look-ups replicated) rc $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_seat_*_of_8_*lookups.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
        r=j['per_rank'][0]; s=r['stages_ms']
        print(f, 'device_ms', r['device_ms'], {k: round(v,1) for k,v in s.items() if k.startswith('ms_')})
    except Exception as e: print(f, 'ERR', e)
PY

# round 5: seats 0, 3, 7 of an 8-rank job of configuration #4 (100 M x 150 bp), one at a time on one GPU (LEON_BENCH_AS_RANK), last build
cd $GRAFT_REPO_ROOT
for r in 0 3 7; do
  LEON_BENCH_AS_RANK=$r:8 timeout -k 10 400 python bench.py --quick --cpu-sample 0 --steps 3 --warmup 1 > gpurun_out/r5_seat_${r}_of_8.json 2> gpurun_out/r5_seat_${r}_of_8.err || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/r5_seat_${r}_of_8.json')); p=d['per_rank'][0]; print('seat $r of 8: device', p['device_ms'], {k: v for k, v in p['stages_ms'].items() if k in ('ms_pack','ms_resolve','ms_sort','ms_walk','ms_exchange','ms_symbols','ms_rangecoder','ms_emulated')})"
done

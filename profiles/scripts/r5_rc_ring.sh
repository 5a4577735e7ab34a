# round 5: k_rc_encode with the ring kept by counters (depth 3) -- parity tests that reach it under a short limit first, then timing
# (ran against the counter-kept ring, which was dropped: HISTORY.md, Round 5)
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5rc
mkdir -p $O
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "every_group_size or range_coder or host_chains or toy" > $O/tests0.log 2>&1 || { tail -30 $O/tests0.log; exit 1; }
tail -2 $O/tests0.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streams.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for cmp in 1 0 1 0; do
  LEON_RC_CMP=$cmp timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --quick > $O/ab.json 2> $O/ab.err || exit 1
  python3 -c "
import json,sys
d=json.load(open('$O/ab.json')); s=d['per_rank'][0]['stages_ms']
print('default counts_apart=$cmp rangecoder', s['ms_rangecoder'], 'total', s['ms_total'])" | tee -a $O/ring.txt
  LEON_BENCH_K=63 LEON_BENCH_L=250 LEON_RC_CMP=$cmp timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --quick --reads 20000000 --batch-reads 20000000 > $O/ab.json 2> $O/ab.err || exit 1
  python3 -c "
import json,sys
d=json.load(open('$O/ab.json')); s=d['per_rank'][0]['stages_ms']
print('k63 counts_apart=$cmp rangecoder', s['ms_rangecoder'], 'total', s['ms_total'])" | tee -a $O/ring.txt
done

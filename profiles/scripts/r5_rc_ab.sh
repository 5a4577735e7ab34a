# round 5: k_rc_encode A/B on one box -- emitter wave on / off, byte-count models apart on / off (numeric slots per block in LDS: 12 / 13 / 13 / 14)
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5rc
mkdir -p $O
timeout -k 10 150 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "every_group_size or range_coder or host_chains" > $O/tests0.log 2>&1 || { tail -30 $O/tests0.log; exit 1; }
tail -2 $O/tests0.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streams.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for cfg in "1 1" "0 1" "1 0" "0 0" "1 1" "0 0"; do
  set -- $cfg
  LEON_RC_EMIT=$1 LEON_RC_CMP=$2 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --quick > $O/ab_$1_$2.json 2> $O/ab.err || exit 1
  python3 -c "
import json,sys
d=json.load(open('$O/ab_$1_$2.json')); s=d['per_rank'][0]['stages_ms']
print('emit=$1 cmp=$2 rangecoder', s['ms_rangecoder'], 'total', s['ms_total'])" | tee -a $O/ab.txt
done

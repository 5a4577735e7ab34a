# k_rc_encode at the k = 63 / 250 bp shape (400 blocks of 1.9 M symbols, 2 per workgroup) and at the default: ms_rangecoder of bench.py --quick
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5rc
mkdir -p $O
for cfg in "1 1" "0 0"; do
  set -- $cfg
  LEON_BENCH_K=63 LEON_BENCH_L=250 LEON_RC_EMIT=$1 LEON_RC_CMP=$2 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --quick --reads 20000000 --batch-reads 20000000 > $O/k63_$1_$2.json 2> $O/k63.err || { tail -5 $O/k63.err; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('$O/k63_$1_$2.json')); s=d['per_rank'][0]['stages_ms']
print('k63 emit=$1 cmp=$2 rangecoder', s['ms_rangecoder'], 'total', s['ms_total'])" | tee -a $O/k63.txt
  LEON_RC_EMIT=$1 LEON_RC_CMP=$2 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --quick > $O/ab_$1_$2.json 2> $O/ab.err || exit 1
  python3 -c "
import json,sys
d=json.load(open('$O/ab_$1_$2.json')); s=d['per_rank'][0]['stages_ms']
print('default emit=$1 cmp=$2 rangecoder', s['ms_rangecoder'], 'total', s['ms_total'])" | tee -a $O/k63.txt
done

cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_wc_ab.txt; : > $O
for rep in 1 2; do for wcv in 1 0; do
  LEON_WALK_CACHE=$wcv timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 3 --warmup 1 > /tmp/o.json 2>/dev/null || exit 1
  python3 -c "
import json; d=json.load(open('/tmp/o.json')); print('cache $wcv: walk', d['stages_ms_rank0']['ms_walk'], 'device', d['stages_ms_rank0']['ms_total'], 'step', d['ms_per_step'], 'frac', d['roofline']['frac'])" >> $O
done; done
cat $O

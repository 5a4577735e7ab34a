cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_wc_sweep.txt; : > $O
for hop in 2 3 4 5; do for lg in 0 24 27; do
  LEON_WALK_HOP_LOG2=$hop LEON_WALK_CACHE_LOG2=$lg timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 2 --warmup 1 > /tmp/o.json 2>/dev/null || exit 1
  python3 -c "
import json; d=json.load(open('/tmp/o.json')); print('hop 1/2^$hop buckets 2^$lg: walk', d['stages_ms_rank0']['ms_walk'], 'device', d['stages_ms_rank0']['ms_total'])" >> $O
done; done
cat $O

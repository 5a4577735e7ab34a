# round 5: the walk's path cache on / off (LEON_WALK_CACHE) on ONE box: the headline workload, configuration #2, the k = 63 / 250 bp shape
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_wc_ab.txt; : > $O
run() {  # label, env..., then bench args
  label=$1; shift
  for wcv in 1 0; do
    env LEON_WALK_CACHE=$wcv "$@" > /tmp/o.json 2>/dev/null || exit 1
    python3 -c "
import json; d=json.load(open('/tmp/o.json')); print('$label cache $wcv: walk', d['stages_ms_rank0']['ms_walk'], 'device', d['stages_ms_rank0']['ms_total'], 'step', d['ms_per_step'], 'frac', d['roofline']['frac'])" >> $O
  done
}
run "100M x 150 k31" timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 3 --warmup 1
run "100M x 150 k31" timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 3 --warmup 1
run " 10M x 150 k31" timeout -k 10 300 python bench.py --reads 10000000 --quick --cpu-sample 0 --steps 5 --warmup 2
run " 20M x 250 k63" env LEON_BENCH_K=63 LEON_BENCH_L=250 timeout -k 10 300 python bench.py --reads 20000000 --quick --cpu-sample 0 --steps 3 --warmup 1
cat $O

# round 5: configuration #2's range-coder stage: four modeler waves per block (LEON_RC_RECORDS_WAVES=1: the round-4 kernel), two chains side by side per host thread (LEON_RC_HOST_PAIRS=0: off)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_cfg2_pairs.txt; : > $O
for rep in 1 2; do for spec in "4 1" "4 0" "1 1" "1 0"; do
  set -- $spec
  LEON_RC_RECORDS_WAVES=$1 LEON_RC_HOST_PAIRS=$2 LEON_TRACE_RC_HOST=1 timeout -k 10 300 python bench.py --reads 10000000 --quick --cpu-sample 0 --steps 5 --warmup 2 > /tmp/o.json 2> /tmp/o.err || exit 1
  python3 -c "
import json; d=json.load(open('/tmp/o.json')); print('modeler waves $1, pairs $2: rc', d['stages_ms_rank0']['ms_rangecoder'], 'device', d['stages_ms_rank0']['ms_total'], 'step', d['ms_per_step'], 'value', d['value'])" >> $O
  grep "leon rc host" /tmp/o.err | tail -1 | grep -o "chunk 15 in host memory at [0-9.]* ms, coded at [0-9.]*; done at [0-9.]* ms" >> $O
done; done
cat $O

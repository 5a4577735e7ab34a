# HBM traffic of the decoder's kernels at 10 M reads: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), kernel trace only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2dp
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  LEON_BENCH_READS=10000000 timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_decode_blocks|k_pc_prewalk|k_pc_init" --output-format csv -d $O/$c -- python3 $R/bench.py --steps 1 --warmup 0 --decode --cpu-sample 0 > $O/$c.json 2> $O/$c.err || echo "pass $c failed"
done
find $O -name "*counter_collection.csv" | head

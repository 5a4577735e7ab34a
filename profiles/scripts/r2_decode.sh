#!/bin/bash
# the device decoder with its path cache: 100 M / 10 M reads (k = 31), 20 M x 250 bp (k = 63), the cache switched off for
# comparison, and the kernel statistics of the 10 M-read run
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2d; mkdir -p $O
export LEON_TRACE_DECODE=1
run() { name=$1; shift; echo "== $name"; timeout -k 10 500 env "$@" python bench.py --steps 1 --warmup 0 --decode --cpu-sample 0 > $O/$name.json 2> $O/$name.err; python -c "import json;print(json.load(open('$O/$name.json')).get('decode'))"; grep "leon decode" $O/$name.err; }
run decode_100M LEON_BENCH_READS=100000000
run decode_10M LEON_BENCH_READS=10000000
run decode_10M_cache_off LEON_BENCH_READS=10000000 LEON_DC_CACHE_MB=0
run decode_k63_20M LEON_BENCH_READS=20000000 LEON_BENCH_K=63 LEON_BENCH_L=250
cd /tmp && export TMPDIR=/tmp
LEON_BENCH_READS=10000000 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o dec10 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --decode --cpu-sample 0 > $GRAFT_REPO_ROOT/$O/prof.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats*" | head -3

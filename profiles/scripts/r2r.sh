set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r2r
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2r/prof -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/r2r/prof.json 2> $R/gpurun_out/r2r/prof.err
LEON_BENCH_K=63 LEON_BENCH_L=250 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2r/prof63 -- python3 $R/bench.py --reads 20000000 --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/r2r/prof63.json 2> $R/gpurun_out/r2r/prof63.err
echo done

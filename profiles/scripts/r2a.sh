set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py --durations=8 > gpurun_out/r2a/pytest_parity.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest_parity.log
tail -5 gpurun_out/r2a/pytest_parity.log
LEON_TRACE_ALLOC=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/r2a/cold.json 2> gpurun_out/r2a/cold.err
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r2a/warm.json 2> gpurun_out/r2a/warm.err
LEON_BENCH_K=63 LEON_BENCH_L=250 timeout -k 10 300 python bench.py --reads 20000000 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r2a/k63.json 2> gpurun_out/r2a/k63.err
cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/r2a/prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2a/prof.err
echo done

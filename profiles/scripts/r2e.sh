set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r2e/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2e/pytest.log
tail -n 4 gpurun_out/r2e/pytest.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r2e/warm.json 2> gpurun_out/r2e/warm.err
timeout -k 10 300 python bench.py --reads 10000000 --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/r2e/cfg2.json 2> gpurun_out/r2e/cfg2.err
LEON_BENCH_K=63 LEON_BENCH_L=250 timeout -k 10 300 python bench.py --reads 20000000 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r2e/k63.json 2> gpurun_out/r2e/k63.err
cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2e/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/r2e/prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2e/prof.err
echo done

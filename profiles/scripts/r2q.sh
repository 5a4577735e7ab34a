set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2q
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_streams.py -x -q > gpurun_out/r2q/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2q/pytest.log
tail -n 4 gpurun_out/r2q/pytest.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r2q/warm.json 2> gpurun_out/r2q/warm.err
LEON_BENCH_K=63 LEON_BENCH_L=250 timeout -k 10 300 python bench.py --reads 20000000 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r2q/k63.json 2> gpurun_out/r2q/k63.err
timeout -k 10 300 python bench.py --reads 10000000 --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/r2q/cfg2.json 2> gpurun_out/r2q/cfg2.err
echo done

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout -k 10 600 python -m pytest tests/test_gpu_streams.py tests/test_gpu_parity.py -x -q --durations=5 > gpurun_out/r2c/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2c/pytest.log
tail -n 15 gpurun_out/r2c/pytest.log
LEON_TRACE_ALLOC=1 timeout -k 10 300 python bench.py --steps 2 --warmup 0 --cpu-sample 0 > gpurun_out/r2c/cold.json 2> gpurun_out/r2c/cold.err
tail -n 3 gpurun_out/r2c/cold.json | cut -c1-600

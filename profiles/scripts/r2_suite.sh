set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=10 > gpurun_out/r2s/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r2s/pytest_gpu.log
tail -n 20 gpurun_out/r2s/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2s/smoke.log 2>&1; tail -n 2 gpurun_out/r2s/smoke.log

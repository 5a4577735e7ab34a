set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py --durations=8 > gpurun_out/r2d/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2d/pytest.log
tail -n 40 gpurun_out/r2d/pytest.log

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py --durations=5 > gpurun_out/r2m/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2m/pytest.log
tail -n 14 gpurun_out/r2m/pytest.log
timeout -k 10 1000 python profiles/scripts/cli_at_scale.py > gpurun_out/r2m/cli.json 2> gpurun_out/r2m/cli.err
tail -c 1800 gpurun_out/r2m/cli.json

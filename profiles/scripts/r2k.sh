set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streams.py tests/test_host_cli.py -x -q > gpurun_out/r2k/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2k/pytest.log
tail -n 4 gpurun_out/r2k/pytest.log
timeout -k 10 1000 python profiles/scripts/cli_at_scale.py > gpurun_out/r2k/cli.json 2> gpurun_out/r2k/cli.err
tail -c 2500 gpurun_out/r2k/cli.json

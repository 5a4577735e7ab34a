set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2y
timeout -k 10 600 python -m pytest tests/test_host_cli.py -x -q > gpurun_out/r2y/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2y/pytest.log
tail -n 3 gpurun_out/r2y/pytest.log
export LEON_CLI_DIR=/dev/shm/leon_cli
LEON_CLI_READS=100000000 timeout -k 10 1100 python profiles/scripts/cli_at_scale.py > gpurun_out/r2y/cli.json 2> gpurun_out/r2y/cli.err
rm -rf /dev/shm/leon_cli
python -c "
import json;d=json.load(open('gpurun_out/r2y/cli.json'));print(d['compress_lossless_s'],d['decompress_s'],d['decompress_stdout'],d['compress_lossy_s'])"

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2w
export LEON_CLI_DIR=/dev/shm/leon_cli
LEON_CLI_READS=100000000 timeout -k 10 1150 python profiles/scripts/cli_at_scale.py > gpurun_out/r2w/cli.json 2> gpurun_out/r2w/cli.err
rm -rf /dev/shm/leon_cli
tail -c 2500 gpurun_out/r2w/cli.json
tail -n 3 gpurun_out/r2w/cli.err

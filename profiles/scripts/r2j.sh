set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
LEON_TRACE_ALLOC=1 LEON_CLI_READS=5000000 timeout -k 10 1000 python profiles/scripts/cli_at_scale.py > gpurun_out/r2j/cli.json 2> gpurun_out/r2j/cli.err
tail -c 2500 gpurun_out/r2j/cli.json

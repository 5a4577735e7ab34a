set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout -k 10 1000 python profiles/scripts/cli_at_scale.py > gpurun_out/r2i/cli.json 2> gpurun_out/r2i/cli.err
tail -c 3000 gpurun_out/r2i/cli.json
tail -n 5 gpurun_out/r2i/cli.err

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2x
LEON_CLI_READS=30000000 timeout -k 10 1100 python profiles/scripts/cli_at_scale.py > gpurun_out/r2x/cli.json 2> gpurun_out/r2x/cli.err
python -c "
import json;d=json.load(open('gpurun_out/r2x/cli.json'));print(d['decompress_s'],d['decompress_stdout'])"

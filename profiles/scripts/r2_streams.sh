set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2st
timeout -k 10 600 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --streams > gpurun_out/r2st/streams.json 2> gpurun_out/r2st/streams.err
tail -n 5 gpurun_out/r2st/streams.err
python -c "
import json;d=json.loads([l for l in open('gpurun_out/r2st/streams.json') if l.startswith('{')][0]);print(json.dumps(d['streams'],indent=1))"

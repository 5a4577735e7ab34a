#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -m pytest tests/test_host_cli.py -x -q -m gpu > gpurun_out/cli_pytest.log 2>&1 || { tail -30 gpurun_out/cli_pytest.log; exit 1; }
tail -2 gpurun_out/cli_pytest.log
timeout -k 10 500 python profiles/scripts/cli_at_scale.py > gpurun_out/cli10_pipe.json 2> gpurun_out/cli10_pipe.err
python - <<'P'
import json
d=json.load(open('gpurun_out/cli10_pipe.json'))
for k in d:
    if k.endswith('_s') or k.endswith('stdout'): print(k, d[k] if not isinstance(d[k],list) else d[k][-2:])
P

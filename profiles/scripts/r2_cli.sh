#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -m pytest tests/test_host_cli.py -x -q -m gpu > gpurun_out/cli_pytest.log 2>&1 || { tail -30 gpurun_out/cli_pytest.log; exit 1; }
tail -2 gpurun_out/cli_pytest.log

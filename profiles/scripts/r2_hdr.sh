#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_streams.py tests/test_host_cli.py -x -q -m gpu > gpurun_out/hdr_pytest.log 2>&1 || { tail -30 gpurun_out/hdr_pytest.log; exit 1; }
tail -2 gpurun_out/hdr_pytest.log

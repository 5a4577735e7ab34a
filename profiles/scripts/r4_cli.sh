#!/bin/bash
# leon -d after the header symbols moved to one device call per file: the CLI tests, then configuration #3 at full size
set -o pipefail
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_host_cli.py tests/test_gpu_streams.py -m gpu -x -q > gpurun_out/r4_cli_tests.log 2>&1; echo "cli tests rc $?"; tail -5 gpurun_out/r4_cli_tests.log
grep -q "failed\|error" gpurun_out/r4_cli_tests.log && exit 1
mkdir -p /dev/shm/leon_cli
LEON_QUAL_DEFLATE=device LEON_CLI_READS=100000000 LEON_CLI_DIR=/dev/shm/leon_cli timeout -k 10 1000 python profiles/scripts/cli_at_scale.py > gpurun_out/r4_cli_config3_100M.json 2> gpurun_out/r4_cli_config3_100M.err; echo "cli at scale rc $?"
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r4_cli_config3_100M.json') if l.startswith('{')][-1])
for k in ('generate_s','compress_lossless_s','decompress_s','compress_lossy_s','leon_bytes_lossless','leon_bytes_lossy'): print(k, j.get(k))
for k in ('compress_lossless_stdout','decompress_stdout','compress_lossy_stdout'): print(k, j.get(k))
PY

#!/bin/bash
# round 4, first GPU call: the old-layout golden container, the host CPU's instruction latencies, the new tests
set -o pipefail
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
bash tests/golden/make_r2_layout.sh > gpurun_out/r4_make_r2_layout.log 2>&1 && cp gpurun_out/r2_layout_toy.fasta.leon tests/golden/ || echo "golden generation failed"
g++ -O2 -std=c++17 -mbmi2 -o /tmp/lat profiles/scripts/chain_ab/lat.cpp && /tmp/lat > gpurun_out/r4_host_latencies.txt 2>&1
grep -m1 "model name" /proc/cpuinfo >> gpurun_out/r4_host_latencies.txt; cat /sys/fs/cgroup/cpu.max >> gpurun_out/r4_host_latencies.txt
timeout -k 10 900 python -m pytest tests/test_host_cli.py tests/test_gpu_multiprocess.py -m gpu -x -q > gpurun_out/r4_first_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4_first_pytest.log
tail -5 gpurun_out/r4_first_pytest.log

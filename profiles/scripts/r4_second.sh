#!/bin/bash
# round 4, second GPU call: the predicted-flag chain on the box's host, configuration #5 as rank 0 of 8 through bench.py, the 8-way split at 100 M reads
set -o pipefail
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
CXX=/opt/rocm/lib/llvm/bin/clang++
$CXX -O2 -std=c++17 -mbmi2 -o /tmp/ab_spec profiles/scripts/chain_ab/ab_spec.cpp -lpthread
{ grep -m1 "model name" /proc/cpuinfo; for s in 0 1 2; do echo "skew $s"; /tmp/ab_spec 4000000 $s; done; } > gpurun_out/r4_chain_spec_ab.txt 2>&1
LEON_BENCH_AS_RANK=0:8 LEON_BENCH_K=63 LEON_BENCH_L=250 timeout -k 10 1000 python bench.py --reads 500000000 --quick --steps 1 --warmup 0 > gpurun_out/r4_bench_as_rank0_of_8_cfg5.json 2> gpurun_out/r4_bench_as_rank0_of_8_cfg5.err
echo "cfg5 bench rc $?"
tail -c 600 gpurun_out/r4_bench_as_rank0_of_8_cfg5.err
LEON_FULLSIZE_CASES=100000000:31:150 timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -k full_size -x -q -s > gpurun_out/r4_fullsize_8way.log 2>&1
echo "fullsize rc $?"
tail -12 gpurun_out/r4_fullsize_8way.log

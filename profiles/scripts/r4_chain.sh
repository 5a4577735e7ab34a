#!/bin/bash
# the dictionary chain's variants on the GPU box's host (no GPU work): what the core sustains, then every variant on the same records
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
CXX=/opt/rocm/lib/llvm/bin/clang++
g++ -O2 -std=c++17 -mbmi2 -o /tmp/ipc profiles/scripts/chain_ab/ipc.cpp && /tmp/ipc > gpurun_out/r4_host_ipc.txt 2>&1
$CXX -O2 -std=c++17 -mbmi2 -o /tmp/ab_spec profiles/scripts/chain_ab/ab_spec.cpp -lpthread
{ grep -m1 "model name" /proc/cpuinfo; for rep in 1 2; do /tmp/ab_spec 4000000 0; done; } > gpurun_out/r4_chain_spec_ab2.txt 2>&1
cat gpurun_out/r4_host_ipc.txt; grep -v "from DRAM" gpurun_out/r4_chain_spec_ab2.txt

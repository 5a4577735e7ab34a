#!/bin/bash
# How tests/golden/r2_layout_toy.fasta.leon was made (once, on a GPU box): the round-2 build of this repository (git 12ee041, the last
# commit whose `leon -c` wrote 13 parameter words and a 2-word header block table) compresses tests/golden/toy.fasta.
#   mkdir -p _r2build && git archive 12ee041 leon_amd include | tar -x -C _r2build && make -C _r2build/leon_amd/csrc -j6
#   gpurun -- bash tests/golden/make_r2_layout.sh      # then copy gpurun_out/r2_layout_toy.fasta.leon here
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r2
cp tests/golden/toy.fasta gpurun_out/r2/toy.fasta
_r2build/leon_amd/lib/leon -file gpurun_out/r2/toy.fasta -c
cp gpurun_out/r2/toy.fasta.leon gpurun_out/r2_layout_toy.fasta.leon
/opt/conda/bin/h5ls -r gpurun_out/r2_layout_toy.fasta.leon

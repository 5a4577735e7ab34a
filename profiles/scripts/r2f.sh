set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2f
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "k_walk|k_lookup_cand|k_final_pos|k_check" --output-format csv -d $R/gpurun_out/r2f/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $R/gpurun_out/r2f/fetch.json 2> $R/gpurun_out/r2f/fetch.err
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "k_walk|k_lookup_cand|k_final_pos|k_check" --output-format csv -d $R/gpurun_out/r2f/write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $R/gpurun_out/r2f/write.json 2> $R/gpurun_out/r2f/write.err
ls -la $R/gpurun_out/r2f/fetch/*/ | head
echo done

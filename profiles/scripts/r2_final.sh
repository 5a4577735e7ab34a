set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2z
timeout -k 10 400 python bench.py --steps 3 --warmup 1 --host-input > gpurun_out/r2z/bench_default_100M.json 2> gpurun_out/r2z/bench_default.err
timeout -k 10 300 python bench.py --reads 10000000 --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/r2z/bench_config2_10M.json 2> gpurun_out/r2z/cfg2.err
LEON_BENCH_K=63 LEON_BENCH_L=250 timeout -k 10 300 python bench.py --reads 20000000 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r2z/bench_k63_L250_20M.json 2> gpurun_out/r2z/k63.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2z/prof -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/r2z/bench_under_rocprof_100M.json 2> $R/gpurun_out/r2z/prof.err
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "k_walk|k_lookup_cand|k_final_pos|k_check" --output-format csv -d $R/gpurun_out/r2z/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $R/gpurun_out/r2z/fetch.json 2> $R/gpurun_out/r2z/fetch.err
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "k_walk|k_lookup_cand|k_final_pos|k_check" --output-format csv -d $R/gpurun_out/r2z/write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $R/gpurun_out/r2z/write.json 2> $R/gpurun_out/r2z/write.err
echo done

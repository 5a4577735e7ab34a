set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
LEON_GATHER_TABLES_MIB=1,2,4,16,32,64,128,256,752 timeout -k 10 300 python profiles/gather_ceiling.py > gpurun_out/r2b/gather.txt 2>&1
LEON_TRACE_ALLOC=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --kmer-max-keys 1000000000 > gpurun_out/r2b/cold_small_kmer.json 2> gpurun_out/r2b/cold_small_kmer.err
timeout -k 10 900 python -m pytest tests/test_gpu_multiprocess.py -x -q --durations=5 > gpurun_out/r2b/pytest_mp.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/pytest_mp.log
LEON_FULLSIZE_CASES=40000000:63:250 timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s --durations=5 > gpurun_out/r2b/pytest_k63.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/pytest_k63.log
tail -5 gpurun_out/r2b/pytest_mp.log gpurun_out/r2b/pytest_k63.log
echo done

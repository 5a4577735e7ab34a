set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2o
python profiles/scripts/chain_bench.py 14000000 > gpurun_out/r2o/chain_new.txt 2>&1
cat gpurun_out/r2o/chain_new.txt
git stash -q 2>/dev/null || true
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --cpu-sample 0 > gpurun_out/r2o/warm.json 2> gpurun_out/r2o/warm.err
echo done

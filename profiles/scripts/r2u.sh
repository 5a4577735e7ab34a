set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2u
for i in 1 2; do
python profiles/scripts/chain_bench.py 14000000 >> gpurun_out/r2u/chain_model.txt 2>&1
LEON_CHAIN_EXPERIMENT=1 python profiles/scripts/chain_bench.py 14000000 >> gpurun_out/r2u/chain_nomodel.txt 2>&1
done
cat gpurun_out/r2u/chain_model.txt gpurun_out/r2u/chain_nomodel.txt

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2p
for i in 1 2; do
python profiles/scripts/chain_bench.py 14000000 >> gpurun_out/r2p/chain_new.txt 2>&1
LEON_LIB=$GRAFT_REPO_ROOT/profiles/scripts/oldlib/libleon_dna.so python profiles/scripts/chain_bench.py 14000000 >> gpurun_out/r2p/chain_old.txt 2>&1
done
cat gpurun_out/r2p/chain_new.txt gpurun_out/r2p/chain_old.txt
lscpu | grep -E "Model name|MHz" | head -4

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
for cfg in "256 64" "256 32" "64 32" "64 64" "128 32" "64 40" "256 128"; do
set -- $cfg
LEON_LOOKUP_BLOCK=$1 LEON_LOOKUP_WAVES=$2 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r2l/b$1_w$2.json 2> gpurun_out/r2l/b$1_w$2.err
done
echo done

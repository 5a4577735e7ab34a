set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
for fb in 22 24 25 26 27; do
LEON_FBITS_LOG2=$fb timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r2h/fb$fb.json 2> gpurun_out/r2h/fb$fb.err
done
LEON_RESOLVE_WINDOW=4194304 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r2h/win22.json 2> gpurun_out/r2h/win22.err
LEON_RESOLVE_WINDOW=262144 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r2h/win18.json 2> gpurun_out/r2h/win18.err
echo done

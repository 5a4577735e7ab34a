# round 5: k_rc_encode with the emitter wave -- parity tests that reach the coder, then kernel stats of the default workload and of the k = 63 shape
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5rc
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streams.py tests/test_gpu_structured.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --quick > $R/$O/bench_under_rocprof_100M.json 2> $R/$O/prof.err || exit 1
cd $R
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r5rc/prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(" ", r["Name"][:70], r["Calls"], r["AverageNs"])
PY
echo done

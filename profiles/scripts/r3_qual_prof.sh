R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r3q
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3q/prof -- python3 $R/bench.py --quick --streams --cpu-sample 0 --steps 1 --warmup 0 > $R/gpurun_out/r3q/out.json 2> $R/gpurun_out/r3q/err.txt
python3 - <<'PY'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
for f in glob.glob(R + "/gpurun_out/r3q/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(x in r["Name"] for x in ("minimizer", "solid_flags", "qual_rewrite", "qual_smooth", "k_pack", "hdr_symbols")):
            print(r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, "ms")
PY

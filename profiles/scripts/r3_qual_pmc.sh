R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r3qp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "k_solid_flags|k_qual_smooth|k_qual_rewrite|k_read_minimizer" --output-format csv -d $R/gpurun_out/r3qp/fetch -- python3 $R/bench.py --quick --streams --cpu-sample 0 --steps 1 --warmup 0 > $R/gpurun_out/r3qp/out.json 2> $R/gpurun_out/r3qp/err.txt
python3 - <<'PY'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
tot = {}
for f in glob.glob(R + "/gpurun_out/r3qp/fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            n = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("leon::", "")
            tot[n] = tot.get(n, 0.0) + float(r["Counter_Value"])
for n, v in tot.items():
    print(n, "FETCH_SIZE", round(v * 1024 / 1e9, 1), "GB")
PY

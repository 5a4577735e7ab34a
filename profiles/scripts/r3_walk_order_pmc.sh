# FETCH_SIZE of k_walk with the reads in the product's order (first launch) and in true genome order (second launch)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r3wo
WALK_ORDER_ONLY=2 timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "k_walk" --output-format csv -d $R/gpurun_out/r3wo/fetch -- python3 $R/profiles/scripts/walk_order.py > $R/gpurun_out/r3wo/out.txt 2> $R/gpurun_out/r3wo/err.txt
find $R/gpurun_out/r3wo/fetch -name "*counter_collection.csv" | head -3
python3 - <<'PY'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
for f in glob.glob(R + "/gpurun_out/r3wo/fetch/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    per = {}
    for r in rows:
        if "k_walk" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            per.setdefault(r["Dispatch_Id"], 0.0)
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    for d, v in sorted(per.items(), key=lambda x: int(x[0])):
        print("k_walk dispatch", d, "FETCH_SIZE", v, "KB ->", v * 1024 / 1e9, "GB")
PY
cat $R/gpurun_out/r3wo/out.txt | tail -3

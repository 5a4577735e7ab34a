# round 5: the bench line, kernel stats and PMC passes the docs cite (one MI355X box; ~10 min).  Output: gpurun_out/r5f/
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5f
mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_default_100M.json 2> $O/bench_default.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --quick > $R/$O/bench_under_rocprof_100M.json 2> $R/$O/prof.err || exit 1
DECODE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_sorted -- python3 $R/profiles/scripts/structured_case.py 10000000 sorted > $R/$O/structured_sorted_10M_under_rocprof.json 2> $R/$O/prof_sorted.err || exit 1
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "k_walk|k_lookup_cand|k_final_pos|k_check" --output-format csv -d $R/$O/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --quick > $R/$O/fetch.json 2> $R/$O/fetch.err || exit 1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "k_walk|k_lookup_cand|k_final_pos|k_check" --output-format csv -d $R/$O/write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --quick > $R/$O/write.json 2> $R/$O/write.err || exit 1
cd $R
python3 - <<'PY'
import csv, glob, json, os
O = "gpurun_out/r5f"
def per_kernel(d, counter):
    tot = {}
    for f in glob.glob(O + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("leon::", "")
                tot[name] = tot.get(name, 0.0) + float(r["Counter_Value"])
    return tot
fe, wr = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
print("FETCH_SIZE KB per step:", fe)
print("WRITE_SIZE KB per step:", wr)
json.dump({"fetch_size_kb": fe, "write_size_kb": wr}, open(O + "/pmc_summary.json", "w"), indent=1)
for d in ("prof", "prof_sorted"):
    for f in glob.glob(O + "/" + d + "/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        print(d)
        for r in rows[:16]:
            print(" ", r["Name"][:70], r["Calls"], r["AverageNs"])
PY
echo done

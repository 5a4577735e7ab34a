# round 4, after the minimizer-addressed filter: the bench lines, kernel stats, PMC passes and look-up traces the docs cite (one MI355X box; ~10 min)
set -x
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r4y
mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench_default_100M.json 2> $O/bench_default.err || exit 1
timeout -k 10 300 python bench.py --reads 10000000 --steps 5 --warmup 2 --cpu-sample 0 --quick > $O/bench_config2_10M.json 2> $O/cfg2.err || exit 1
LEON_BENCH_K=63 LEON_BENCH_L=250 timeout -k 10 300 python bench.py --reads 20000000 --steps 3 --warmup 1 --cpu-sample 0 --quick > $O/bench_k63_L250_20M.json 2> $O/k63.err || exit 1
LEON_TRACE_RESOLVE=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --quick 2> $O/trace_100M.err > /dev/null; grep "leon resolve" $O/trace_100M.err > $O/resolve_probes_by_outcome.txt
LEON_TRACE_RESOLVE=1 LEON_BENCH_K=63 LEON_BENCH_L=250 timeout -k 10 300 python bench.py --reads 20000000 --steps 1 --warmup 0 --cpu-sample 0 --quick 2> $O/trace_k63.err > /dev/null; grep "leon resolve" $O/trace_k63.err | sed 's/^/(k = 63, 20 M x 250 bp) /' >> $O/resolve_probes_by_outcome.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --quick > $R/$O/bench_under_rocprof_100M.json 2> $R/$O/prof.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof2 -- python3 $R/bench.py --reads 10000000 --steps 3 --warmup 1 --cpu-sample 0 --quick > $R/$O/bench_under_rocprof_10M.json 2> $R/$O/prof2.err || exit 1
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "k_walk|k_lookup_cand|k_final_pos|k_check" --output-format csv -d $R/$O/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --quick > $R/$O/fetch.json 2> $R/$O/fetch.err || exit 1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "k_walk|k_lookup_cand|k_final_pos|k_check" --output-format csv -d $R/$O/write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --quick > $R/$O/write.json 2> $R/$O/write.err || exit 1
cd $R
python3 - <<'PY'
import csv, glob, json, os
O = "gpurun_out/r4y"
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
for d in ("prof", "prof2"):
    for f in glob.glob(O + "/" + d + "/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        print(d)
        for r in rows[:14]:
            print(" ", r["Name"][:70], r["Calls"], r["AverageNs"])
PY
echo done

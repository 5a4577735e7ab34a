import json, glob, shutil, sys, os, hashlib, csv
O='/root/repo/gpurun_out/r5f'; P='/root/repo/profiles'
def newest(pat):
    fs=glob.glob(O+pat, recursive=True); fs.sort(key=os.path.getmtime); return fs[-1]
s=json.load(open(O+'/pmc_summary.json'))
fe, wr = s['fetch_size_kb']['k_walk'], s['write_size_kb']['k_walk']
h=hashlib.sha256()
for f in ("dna_kernels.hip","leon_device.h"):
    h.update(open('/root/repo/leon_amd/csrc/'+f,'rb').read())
src=h.hexdigest()[:16]
tj=json.load(open(P+'/walk_traffic.json'))
tj.update({"fetch_size_kb": fe, "write_size_kb": wr, "traffic_bytes": int((fe+wr)*1024), "k_walk_source": src, "round": 5})
json.dump(tj, open(P+'/walk_traffic.json','w'), indent=1)
print(src, tj['traffic_bytes'])
shutil.copy(newest('/prof/**/*kernel_stats.csv'), P+'/r5_kernel_stats_100M.csv')
shutil.copy(newest('/prof_sorted/**/*kernel_stats.csv'), P+'/r5_kernel_stats_structured_sorted_10M.csv')
shutil.copy(newest('/fetch/**/*counter_collection.csv'), P+'/r5_pmc_fetch_100M.csv')
shutil.copy(newest('/write/**/*counter_collection.csv'), P+'/r5_pmc_write_100M.csv')
shutil.copy(O+'/pmc_summary.json', P+'/r5_pmc_summary.json')
shutil.copy(O+'/bench_default_100M.json', P+'/r5_bench_default_100M.json')
shutil.copy(O+'/bench_under_rocprof_100M.json', P+'/r5_bench_under_rocprof_100M.json')
shutil.copy(O+'/structured_sorted_10M_under_rocprof.json', P+'/r5_structured_sorted_10M_under_rocprof.json')
d=json.load(open(O+'/bench_default_100M.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['one_stream']['value'])
print(d['stages_ms_rank0']['ms_total'], d['stages_ms_rank0']['ms_walk'], d['stages_ms_rank0']['ms_resolve'])
print(d['structured']['step_ms'], d['structured']['stages_ms'])
for k,v in d['other_configs'].items(): print(k, v['value'], v['ms_per_step'], v['roofline']['frac'], v['stages_ms']['ms_rangecoder'], v['stages_ms']['ms_total'])

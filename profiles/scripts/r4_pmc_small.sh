#!/bin/bash
# instruction counters of the resolution's smaller kernels and of k_symbols (20 M reads, one step): how busy the vector units are
set -o pipefail
R="$(cd "$(dirname "$0")/../.." && pwd)"; O=gpurun_out/r4_pmc_small
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --reads 20000000 --steps 1 --warmup 0 --cpu-sample 0 --quick > $R/$O/bench.json 2> $R/$O/prof.err || exit 1
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_check|k_final_pos|k_symbols|k_lookup_cand|k_walk" --output-format csv -d $R/$O/pmc_$n -- python3 $R/bench.py --reads 20000000 --steps 1 --warmup 0 --cpu-sample 0 --quick > $R/$O/pmc_$n.json 2> $R/$O/pmc_$n.err || exit 1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
O='gpurun_out/r4_pmc_small'
f=glob.glob(O+'/prof/**/*kernel_stats.csv', recursive=True)[0]
t={}
for r in csv.DictReader(open(f)):
    for k in ('k_check','k_final_pos','k_symbols<true>','k_symbols<false>','k_lookup_cand','k_walk'):
        if k in r['Name']: t[k]=float(r['TotalDurationNs'])/1e6
acc=collections.defaultdict(collections.Counter)
for f in glob.glob(O+'/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        for k in t:
            if k in r['Kernel_Name']: acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k in t:
    a=acc[k]
    valu=a.get('SQ_INSTS_VALU',0); salu=a.get('SQ_INSTS_SALU',0)
    # 1024 SIMDs, ~2.1-2.4 GHz: cycles available = ms * 2.4e6 * 1024
    busy = valu*4/(t[k]*2.4e6*1024) if t[k] else 0
    print("%-18s %8.2f ms  VALU %.3g  SALU %.3g  VMEM_RD %.3g  LDS %.3g  waves %.3g  VALU issue share (4 cycles each, 2.4 GHz) %.2f" % (k, t[k], valu, salu, a.get('SQ_INSTS_VMEM_RD',0), a.get('SQ_INSTS_LDS',0), a.get('SQ_WAVES',0), busy))
PY

#!/bin/bash
# k_lookup_cand with the filter addressed by minimizer: kernel times, instruction counters, then the parity of the build
set -o pipefail
R="$(cd "$(dirname "$0")/../.." && pwd)"; O=gpurun_out/r4_mm
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --quick > $R/$O/bench.json 2> $R/$O/prof.err || exit 1
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "FETCH_SIZE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_lookup_cand" --output-format csv -d $R/$O/pmc_$n -- python3 $R/bench.py --reads 20000000 --steps 1 --warmup 0 --cpu-sample 0 --quick > $R/$O/pmc_$n.json 2> $R/$O/pmc_$n.err || exit 1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
O='gpurun_out/r4_mm'
f=glob.glob(O+'/prof/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:70], r['Calls'], round(float(r['TotalDurationNs'])/1e6/4,1), 'ms per step')
for d in glob.glob(O+'/pmc_*/'):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.Counter()
        for r in csv.DictReader(open(f)):
            acc[r['Counter_Name']]+=float(r['Counter_Value'])
        print(d, dict(acc))
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r4_mm/parity.log 2>&1; echo "parity rc $?"; tail -2 gpurun_out/r4_mm/parity.log

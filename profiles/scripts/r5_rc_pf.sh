# k_rc_encode: the coder's records fetched one / two / three steps ahead (lib_pf1, lib, lib_pf3), same box: default workload and the k = 63 shape,
# (ran against builds with the fetch distance as a macro, -DRC_PF=1 / 3, in leon_amd/lib_pf1 / lib_pf3, and with the emitter wave still in the kernel:
# both taken out after the measurement; kept as the record of what ran)
# emitter x counts-apart
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5rc
mkdir -p $O
cp leon_amd/lib/libleon_dna.so /tmp/pf2.so
for v in pf2 pf1 pf3 pf2; do
  if [ $v = pf2 ]; then cp /tmp/pf2.so leon_amd/lib/libleon_dna.so; else cp leon_amd/lib_$v/libleon_dna.so leon_amd/lib/libleon_dna.so; fi
  for cfg in "1 1" "0 1" "0 0"; do
    set -- $cfg
    LEON_RC_EMIT=$1 LEON_RC_CMP=$2 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --quick > $O/pf.json 2> $O/pf.err || { tail -5 $O/pf.err; exit 1; }
    python3 -c "
import json,sys
d=json.load(open('$O/pf.json')); s=d['per_rank'][0]['stages_ms']
print('$v default emit=$1 cmp=$2 rangecoder', s['ms_rangecoder'])" | tee -a $O/pf.txt
    LEON_BENCH_K=63 LEON_BENCH_L=250 LEON_RC_EMIT=$1 LEON_RC_CMP=$2 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --quick --reads 20000000 --batch-reads 20000000 > $O/pf.json 2> $O/pf.err || { tail -5 $O/pf.err; exit 1; }
    python3 -c "
import json,sys
d=json.load(open('$O/pf.json')); s=d['per_rank'][0]['stages_ms']
print('$v k63 emit=$1 cmp=$2 rangecoder', s['ms_rangecoder'])" | tee -a $O/pf.txt
  done
done

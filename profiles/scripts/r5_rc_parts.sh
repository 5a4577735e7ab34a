# round 5: what k_rc_encode's phases wait for -- the kernel with one part left out at a time (builds with -DRC_EXP_*: wrong bytes, timing only)
# (ran against builds of rc_kernels.hip with the models' updates / the rank loop / the coder compiled out by -DRC_EXP_NO_UPDATE / _NO_RANK / _NO_CODER,
# linked into leon_amd/lib_<variant>/; the three #ifdefs were taken out again after the measurement and never committed: kept as the record of what ran)
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5rc
mkdir -p $O
cp leon_amd/lib/libleon_dna.so /tmp/full.so
for v in full NO_UPDATE NO_RANK NO_MODEL NO_CODER; do
  if [ $v = full ]; then cp /tmp/full.so leon_amd/lib/libleon_dna.so; else cp leon_amd/lib_$v/libleon_dna.so leon_amd/lib/libleon_dna.so; fi
  for cfg in "1 1" "0 0"; do
    set -- $cfg
    LEON_RC_EMIT=$1 LEON_RC_CMP=$2 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --quick > $O/parts.json 2> $O/parts.err || { tail -5 $O/parts.err; exit 1; }
    python3 -c "
import json,sys
d=json.load(open('$O/parts.json')); s=d['per_rank'][0]['stages_ms']
print('$v emit=$1 cmp=$2 rangecoder', s['ms_rangecoder'])" | tee -a $O/parts.txt
  done
done

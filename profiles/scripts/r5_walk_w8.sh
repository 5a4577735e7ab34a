# k_walk at seven (lib) or eight (lib_w8: 64 registers, 13 spilt) waves per SIMD, one box, alternating
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5rc
mkdir -p $O
cp leon_amd/lib/libleon_dna.so /tmp/w7.so
for v in w7 w8 w7 w8; do
  if [ $v = w7 ]; then cp /tmp/w7.so leon_amd/lib/libleon_dna.so; else cp leon_amd/lib_w8/libleon_dna.so leon_amd/lib/libleon_dna.so; fi
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --quick > $O/w.json 2> $O/w.err || { tail -5 $O/w.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/w.json')); s=d['per_rank'][0]['stages_ms']
print('$v walk', s['ms_walk'], 'total', s['ms_total'])" | tee -a $O/w8.txt
done

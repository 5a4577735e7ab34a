# round 5: k_rc_encode, 8 blocks per workgroup, on reads of ragged length (more numeric models in use than the 150 bp default: READSIZE every read):
# (ran while the emitter wave, LEON_RC_EMIT, was still in the kernel)
# byte-count models apart or not, emitter or not.  200 blocks in groups of 8 take as long as 2 000 (every block's chain runs at once either way).
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5rc
mkdir -p $O
for cfg in "0 0" "0 1" "1 1" "0 0" "0 1"; do
  set -- $cfg
  DECODE=0 LEON_RC_HOST_BLOCKS=0 LEON_RC_GROUP=8 LEON_RC_EMIT=$1 LEON_RC_CMP=$2 timeout -k 10 200 python profiles/scripts/structured_case.py 10000000 random > $O/ragged.json 2> $O/ragged.err || { tail -5 $O/ragged.err; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('$O/ragged.json')); s=d['stages_ms'] if 'stages_ms' in d else d['stages']
print('ragged 10M emit=$1 cmp=$2 rangecoder', s.get('rangecoder', s.get('ms_rangecoder')))" | tee -a $O/ragged.txt
done

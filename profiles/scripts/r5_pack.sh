# round 5: k_pack with the dword -> read map in LDS and four bases converted at a time
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r5rc
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streams.py tests/test_gpu_structured.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/tests_pack.log 2>&1 || { tail -30 $O/tests_pack.log; exit 1; }
tail -2 $O/tests_pack.log
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --quick > $O/pack.json 2> $O/pack.err || exit 1
python3 -c "
import json,sys
d=json.load(open('$O/pack.json')); s=d['per_rank'][0]['stages_ms']
print('default', s)"
done

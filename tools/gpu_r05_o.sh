export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python bench.py --steps 5 --warmup 1 --repeats 1 --no-alone --no-cpu-baseline --no-host-inclusive --only-extras lr_c4_msaedlib_n15,lr_ins_msawfa_n15,lr_c4_msaedlib_n15_3k,lr_ins_msawfa_n15_2k 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,x in d['extras'].items():
    if isinstance(x,dict) and 'junctions_per_s' in x: print(k, round(x['junctions_per_s'],1), round(x['ms_per_step'],2), round(x['msa_stage_ms'],2), round(x['split_stage_ms'],2))"
timeout 1200 python -m pytest tests/test_gpu_lrmsa.py tests/test_gpu_lr.py tests/test_gpu_edlib_dropin.py tests/test_gpu_big_shapes.py -x -q -m gpu 2>&1 | tail -4

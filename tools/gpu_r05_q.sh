# round 5, call Q: the short-read sparse kernel after the branch-free join (sps_row_best) -- parity of every test that drives it
# and the headline rate
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_split.py tests/test_gpu_fuzz.py tests/test_gpu_lowcx.py -x -q 2>&1 | tail -4 ) 2>&1
for k in 1 2; do
python bench.py --no-extras --no-cpu-baseline --no-host-inclusive 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'].get('achieved'))"
done

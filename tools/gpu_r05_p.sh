# round 5, call P: the whole GPU suite after the branch-free letter helpers + rates of the rows they touch
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > $O/pytest_all2.txt 2>&1
cat $O/pytest_all2.txt
python tools/dense_latency.py 2>&1 | head -3
python bench.py --steps 5 --warmup 1 --repeats 1 --no-alone --no-cpu-baseline --no-host-inclusive --only-extras sr_genotype_classifier,lr_genotype_edit_distance_nw,ins_svt4,lr_c4_align_consensus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,x in d['extras'].items():
    if isinstance(x,dict): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in x.items() if a in ('junctions_per_s','jobs_per_s','pairs_per_s','ms_per_step','gcups')})"

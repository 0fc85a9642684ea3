export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
python bench.py --steps 5 --warmup 1 --repeats 1 --no-alone --no-cpu-baseline --no-host-inclusive --only-extras u_full_n20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('only u_full_n20:', d['extras']['u_full_n20'])"
python bench.py --steps 5 --warmup 1 --repeats 1 --no-alone --no-cpu-baseline --no-host-inclusive --only-extras u_c2_40k_junctions,u_full_n20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('after 40k:', d['extras']['u_full_n20'])"
python tools/msa_rate.py 2000 20

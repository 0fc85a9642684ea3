export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for S in 1 0; do
if [ $S = 1 ]; then export DELLYHIP_MSA_NO_ORDER=1; else unset DELLYHIP_MSA_NO_ORDER; fi
python bench.py --steps 5 --warmup 1 --repeats 1 --no-alone --no-cpu-baseline --only-extras sr_stage_mixed_all_svt 2>/dev/null | S=$S python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
x=d['extras']['sr_stage_mixed_all_svt']; print('given order' if os.environ.get('S')=='1' else 'most expensive first', x['junctions_per_s'], x['ms_per_step'], x['msa_stage_ms'], x['split_stage_ms'], x['host_inclusive']['value'])"
done
timeout 1200 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_msa.py tests/test_gpu_stream.py -x -q -m gpu -k "mixed or msa or stream" 2>&1 | tail -4

export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for S in 1 0; do
DELLYHIP_LRI_SERIAL=$S python bench.py --steps 5 --warmup 1 --repeats 1 --no-alone --no-cpu-baseline --only-extras sr_stage_mixed_all_svt 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
x=d['extras']['sr_stage_mixed_all_svt']; print('serial' if os.environ.get('DELLYHIP_LRI_SERIAL')=='1' else 'side stream', x['junctions_per_s'], x['ms_per_step'], x['msa_stage_ms'], x['split_stage_ms'], x['host_inclusive']['value'])"
done
timeout 1200 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_lr.py tests/test_gpu_stream.py tests/test_gpu_lowcx.py -x -q -m gpu -k "mixed or lr or stream or ins or Ins" 2>&1 | tail -4

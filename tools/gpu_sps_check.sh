# quick loop for split_sparse_kernel work: parity suites of the sparse path, then phase timing and the alone / two-in-flight rates
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_lowcx.py tests/test_gpu_split.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -6
[ -f tools/bin/lib_timing.bin ] && DELLYHIP_LIB=tools/bin/lib_timing.bin python tools/sr_phases.py 2>&1 | sed -n 1,12p
python bench.py --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline --no-extras --no-host-inclusive 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('two in flight %.2f M/s | alone %.2f M/s kernel %.3f ms' % (d['value']/1e6, c['one_launch_at_a_time_alignments_per_s']/1e6, c['one_launch_at_a_time_kernel_ms']))"

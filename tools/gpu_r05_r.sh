# round 5, call R: the asynchronous fetch (dellyhip_batch_fetch_begin / _end) -- its tests, the two-process suite, and the N > 1 step
# rates with one rank (--force-comm) and two ranks on the one device
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_fetch_async.py -x -q 2>&1 | tail -15 ) 2>&1
( time timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_robustness.py -x -q 2>&1 | tail -8 ) 2>&1
show='
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d["config"]
print("value", round(d["value"]/1e6,2), "M/s  ms/step", round(d["ms_per_step"],4), {k:(round(v,4) if isinstance(v,float) else v) for k,v in c.items() if k.startswith(("gather_","shm_return_","value_return","launches_in")) and not isinstance(v,str) or k=="value_return_path"})'
echo "--- one rank, --force-comm"
timeout 300 python bench.py --force-comm --no-extras --no-cpu-baseline --no-host-inclusive 2>$O/fc.err | python -c "$show" || tail -5 $O/fc.err
echo "--- two ranks on one device"
timeout 300 python bench.py --gpus 2 --oversubscribe --no-extras --no-cpu-baseline --no-host-inclusive 2>$O/os.err | python -c "$show" || tail -5 $O/os.err

# round 5, call T: the asynchronous fetch on the device's download stream, four resident batches in the N > 1 step
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_fetch_async.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -6
show='
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d["config"]
print("value", round(d["value"]/1e6,2), "M/s  ms/step", round(d["ms_per_step"],4), {k:(round(v,4) if isinstance(v,float) else v) for k,v in c.items() if k in ("gather_alignments_per_s","gather_step_ms","shm_return_gather_ms_per_step","kernels_ms_per_step_rank0")})'
echo "--- one rank, --force-comm, download stream"
timeout 200 python bench.py --force-comm --gather shm --no-extras --no-cpu-baseline --no-host-inclusive 2>$O/fc.err | timeout 20 python -c "$show" || tail -5 $O/fc.err
echo "--- one rank, --force-comm, same stream"
DELLYHIP_FETCH_SAME_STREAM=1 timeout 200 python bench.py --force-comm --gather shm --no-extras --no-cpu-baseline --no-host-inclusive 2>$O/fc2.err | timeout 20 python -c "$show" || tail -5 $O/fc2.err
echo "--- one rank, both paths"
timeout 200 python bench.py --force-comm --no-extras --no-cpu-baseline --no-host-inclusive 2>$O/fc3.err | timeout 20 python -c "$show" || tail -5 $O/fc3.err
echo "--- two ranks on one device"
timeout 200 python bench.py --gpus 2 --oversubscribe --no-extras --no-cpu-baseline --no-host-inclusive 2>$O/os.err | timeout 20 python -c "$show" || tail -5 $O/os.err

# round 5, call V: two returns in flight (two segments per rank), alternating transfer streams -- an experiment whose code is NOT in the tree (no gain: DESIGN.md 5)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_fetch_async.py tests/test_gpu_multirank.py -x -q 2>&1 < /dev/null | tail -6
show='
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d["config"]
print("value", round(d["value"]/1e6,2), "M/s  ms/step", round(d["ms_per_step"],4), {k:(round(v,4) if isinstance(v,float) else v) for k,v in c.items() if k in ("gather_alignments_per_s","shm_return_gather_ms_per_step","kernels_ms_per_step_rank0")})'
for one in 0 1; do
echo "--- one rank, --force-comm, DELLYHIP_FETCH_ONE_STREAM=$one"
DELLYHIP_FETCH_ONE_STREAM=$one timeout 100 python bench.py --force-comm --gather shm --no-extras --no-cpu-baseline --no-host-inclusive 2>$O/fc.err < /dev/null | timeout 20 python -c "$show" || tail -5 $O/fc.err
done
echo "--- one rank, sparse waves 16, two streams"
DELLYHIP_SPS_WAVES=16 timeout 100 python bench.py --force-comm --gather shm --no-extras --no-cpu-baseline --no-host-inclusive 2>$O/fc.err < /dev/null | timeout 20 python -c "$show" || tail -5 $O/fc.err
echo "--- two ranks on one device, both paths"
timeout 100 python bench.py --gpus 2 --oversubscribe --no-extras --no-cpu-baseline --no-host-inclusive 2>$O/os.err < /dev/null | timeout 20 python -c "$show" || tail -5 $O/os.err

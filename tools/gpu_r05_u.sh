# round 5, call U: which stream for the asynchronous return -- the low-priority download stream or the high-priority upload stream
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
show='
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d["config"]
print("value", round(d["value"]/1e6,2), "M/s  ms/step", round(d["ms_per_step"],4), {k:(round(v,4) if isinstance(v,float) else v) for k,v in c.items() if k in ("shm_return_gather_ms_per_step","kernels_ms_per_step_rank0")})'
for hp in 0 1; do
for w in 12 16 8; do
echo "--- one rank, high priority $hp, sparse waves $w"
DELLYHIP_SPS_WAVES=$w DELLYHIP_FETCH_HIGH_PRIORITY=$hp timeout 100 python bench.py --force-comm --gather shm --no-extras --no-cpu-baseline --no-host-inclusive 2>$O/fc.err | timeout 20 python -c "$show" || tail -5 $O/fc.err
done
echo "--- two ranks on one device, high priority $hp"
DELLYHIP_FETCH_HIGH_PRIORITY=$hp timeout 100 python bench.py --gpus 2 --gather shm --oversubscribe --no-extras --no-cpu-baseline --no-host-inclusive 2>$O/os.err | timeout 20 python -c "$show" || tail -5 $O/os.err
done

# the pipelined path under the conditions of bench.py (torch initialised, a torch stream, resident batches)
run() { echo "== $E $*"; env $E python tools/bench_stream.py --batches 300 --depths ${DEPTHS:-6} "$@" 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d.items(): print(k, round(v['junctions_per_s']/1e6,2), round(v['ms_per_batch'],3), {a:round(b,3) for a,b in v['host_ms_per_batch'].items() if a.endswith('_s')})"; }
E="A=1" run --torch 0
E="A=1" run --torch 2
E="A=1" run --torch 3
E="DELLYHIP_STREAM_NO_PROBE=1" run --torch 2
E="GPU_MAX_HW_QUEUES=2" run --torch 2

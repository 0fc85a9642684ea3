# the pipelined path created several times in one process (the HIP stream -> hardware queue mapping must not depend on history)
run() { echo "== $*"; "$@" python tools/bench_stream.py --batches 200 --depths ${DEPTHS:-4,6,4,8,3} 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d.items(): print(k, round(v['junctions_per_s']/1e6,2), round(v['ms_per_batch'],3), {a:round(b,3) for a,b in v['host_ms_per_batch'].items() if a.endswith('_s')})"; }
run env A=1
run env DELLYHIP_STREAM_PACK_PRIO=0
run env DELLYHIP_STREAM_NO_PACK=1
run env DELLYHIP_SPS_WAVES=12

# resident C2 batches: one at a time and two in flight, by wavefronts of the sparse kernel per CU
for w in 16 14 12 10; do
  DELLYHIP_SPS_WAVES=$w python bench.py --no-cpu-baseline --no-host-inclusive --only-extras u_c2_two_batches_in_flight 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
e=d['extras']['u_c2_two_batches_in_flight']
print('waves/CU', $w, 'one at a time: %.2f M/s, kernel %.3f ms' % (d['value']/1e6, d['roofline']['kernel_ms']), '| two in flight: %.2f M/s' % (e['junctions_per_s']/1e6))"
done

# resident C2 batches, one launch at a time and two in flight (bench.py's headline), by wavefronts of the sparse kernel per CU
for w in 16 14 12 10; do
  DELLYHIP_SPS_WAVES=$w python bench.py --no-cpu-baseline --no-host-inclusive --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
a=d['roofline']['one_launch_at_a_time']
print('waves/CU', $w, '| one launch at a time: %.2f M/s, kernel %.3f ms' % (a['alignments_per_s']/1e6, a['kernel_ms']), '| two in flight (value): %.2f M/s, kernel %.3f ms' % (d['value']/1e6, d['roofline']['kernel_ms']))"
done

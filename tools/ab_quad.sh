for N in 8192 10000 12288 16384 40000; do
  for L in libdh_w4.so libdellyhip.so; do
  DELLYHIP_LIB=$PWD/delly_amd/$L timeout 80 python bench.py --junctions $N --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L N=$N', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), 'per10k', round(d['roofline']['kernel_ms']*10000/$N,3))"
  done
  DELLYHIP_QUAD=0 timeout 80 python bench.py --junctions $N --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pair N=$N', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), 'per10k', round(d['roofline']['kernel_ms']*10000/$N,3))"
done

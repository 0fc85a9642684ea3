# HBM traffic of the dominant kernel from PMC counters, separate passes (MI355X_MICROARCH.md, HBM section)
export TMPDIR=/tmp
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_$C -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pmc_$C.log 2>&1 < /dev/null
  f=$(find gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" $C <<'PY'
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") == c:
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "split_" in k: print(c, k, "launches", len(v), "mean", sum(v) / len(v))
PY
  fi
done

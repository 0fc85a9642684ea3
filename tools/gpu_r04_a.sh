# round 4, GPU call A: the new multi-rank tests, the RCCL duplicate-device probe, the start-of-round bench line, I-cache counters
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04a; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_robustness.py -x -q > $O/pytest_multirank.log 2>&1; echo "pytest rc=$?" >> $O/pytest_multirank.log
tail -5 $O/pytest_multirank.log
timeout 200 python tools/rccl_dup_probe.py > $O/rccl_dup_probe.log 2>&1
cat $O/rccl_dup_probe.log
timeout 900 python bench.py > $O/bench_start.json 2> $O/bench_start.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04a/bench_start.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "dtype")})
    print({k: v for k, v in d["config"].items() if isinstance(v, (int, float))})
except Exception as e:
    print("bench parse failed", e)
PY
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL SQC_DCACHE_REQ SQC_DCACHE_MISSES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_ic/s$i -o p -- python bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-extras --no-host-inclusive --no-alone > $O/pmc_ic_$i.log 2>&1 < /dev/null
done
python - <<'PY' | tee gpurun_out/r04a/pmc_icache_split_sparse.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r04a/pmc_ic/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "split_sparse" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c in sorted(acc):
    v = acc[c]
    print("%-28s %16.0f  (launches %d)" % (c, sum(v) / len(v), len(v)))
PY
rm -rf $O/pmc_ic
ls -la $O

# round 5, call Y: the measurement DESIGN.md 6.3 names -- instruction-cache and wait counters of lrwfa_kernel / lrmsa_kernel at the small and
# the chip-filling batch size (separate --pmc passes, kernel trace only).  NOT COMPLETED in round 5: as first written the passes also
# ran the headline's 25 timed regions under the counters and hit their time limits with the round's last GPU minutes; the flags below
# (--repeats 1 --no-alone) are the fix, unmeasured.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
rm -rf /tmp/pmc_lr
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_lr/s$i -o p -- python bench.py --steps 2 --warmup 1 --repeats 1 --no-alone --no-cpu-baseline --no-host-inclusive --only-extras lr_ins_msawfa_n15,lr_ins_msawfa_n15_2k,lr_c4_msaedlib_n15,lr_c4_msaedlib_n15_3k > $O/pmc_lr_$i.log 2>&1 < /dev/null
done
timeout 60 python - <<'PY' | tee $O/pmc_icache_lr_consensus.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_lr/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "lrwfa_kernel" in k or "lrmsa_kernel" in k:
            # the small and the chip-filling launches differ in their grid
            acc[(k.split("(")[0], r.get("Grid_Size", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), d in sorted(acc.items()):
    print(k, "grid", g, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(d.items())}, "launches", max(len(v) for v in d.values()))
PY

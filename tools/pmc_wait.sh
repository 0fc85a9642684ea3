# where the wavefronts of a kernel spend their cycles (separate --pmc passes, kernel trace only); run on the GPU box from
# the repo root: BENCH_ARGS="--no-extras" KERNEL=split_sparse bash tools/pmc_wait.sh
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_wait
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_wait/s$i -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:---no-extras} > gpurun_out/pmc_wait_$i.log 2>&1 < /dev/null
done
python - <<'PY' | tee gpurun_out/pmc_wait_summary.txt
import csv, glob, collections, os
kern = os.environ.get("KERNEL", "split_sparse")
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_wait/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c in sorted(acc):
    v = acc[c]
    print("%-24s %14.0f  (launches %d)" % (c, sum(v) / len(v), len(v)))
PY

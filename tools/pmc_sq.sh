# SQ issue counters of the dominant kernels (separate --pmc pass, kernel trace only): VALU instructions, busy cycles,
# wave-cycles; run on the GPU box from the repo root.  Output: gpurun_out/pmc_sq_summary.txt + the raw csv.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_sq -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:---only-extras sr_genotype_classifier,lr_genotype_edit_distance_nw} > gpurun_out/pmc_sq.log 2>&1 < /dev/null
f=$(find gpurun_out/pmc_sq -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then python - "$f" <<'PY' | tee gpurun_out/pmc_sq_summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in acc.items():
    if not any(x in k for x in ("split_sparse", "split_quad", "split_post", "classify_kernel<1>", "nw_jobs", "msa_kernel", "lrmsa", "lr_kernel", "ins_kernel", "myers_pairs", "lrwfa", "wfa_pairs", "blob_", "lr_dense_team")):
        continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    ns = sum(dur[k]) / len(dur[k])
    print(k, "launches", len(dur[k]) // max(1, len(d)), "avg_ns(with counters)", round(ns), {c: round(v) for c, v in m.items()})
PY
fi

# One parametrised GPU-box script (replaces the per-call tools/gpu_rNN_[a-z].sh of earlier rounds).
#   gpurun --timeout T -- 'ROUND=r06 PARTS="lrpmc bench" bash tools/gpu_run.sh'
# Everything lands in gpurun_out/$ROUND; copy what is to be judged into profiles/$ROUND.
# PMC passes are separate runs with --kernel-trace only (gpurun refuses --pmc beside the hip/hsa trace domains).
# Parts:
#   counters   rocprofv3 -L (names of the counters this box offers)
#   lrpmc      instruction-cache / SQ / L2 counters of lrwfa_kernel and lrmsa_kernel at the small and the chip-filling batch
#   bench      the default bench line (+ the driver's view of it)
#   stats      rocprofv3 --kernel-trace --stats of the headline and of the side rows
#   traffic    (run `sq` BEFORE it in the same call: the summary takes SQ_INSTS_VALU from that pass)
#              FETCH_SIZE / WRITE_SIZE of the headline's kernels -> pmc_traffic.json (stamped with the kernel sources' hash)
#   sq         SQ instruction counters (tools/pmc_sq.sh), headline and side rows
#   wait       SQ wait counters of split_sparse_kernel and msa_kernel (tools/pmc_wait.sh)
#   pytest     the whole GPU suite; PYTEST_ARGS narrows it
#   py         python $PY_SCRIPT $PY_ARGS (one-off measurement scripts under tools/)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ROUND=${ROUND:-r06}
O=gpurun_out/$ROUND
mkdir -p $O
PARTS=${PARTS:-bench}
TAG=${TAG:-}
EXTRAS=${EXTRAS:-u_full_n20,u_full_n20_10k_junctions,u_full_n5,sr_stage_mixed_all_svt,ins_svt4,lr_c4_align_consensus,lr_c4_msaedlib_n15,lr_ins_msawfa_n15,lr_stress_10kb_x_20kb,sr_genotype_classifier,lr_genotype_edit_distance_nw}
LR_ROWS=${LR_ROWS:-lr_ins_msawfa_n15,lr_ins_msawfa_n15_2k,lr_c4_msaedlib_n15,lr_c4_msaedlib_n15_3k}
QUIET="--repeats 1 --no-alone --no-cpu-baseline --no-host-inclusive"
for P in $PARTS; do
case $P in
counters)
  timeout 120 rocprofv3 -L > $O/counters_available.txt 2>&1 < /dev/null
  grep -c . $O/counters_available.txt
  ;;
lrpmc)
  rm -rf /tmp/pmc_lr
  i=0
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
             "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    timeout ${LRPMC_TIMEOUT:-420} rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_lr/s$i -o p -- python bench.py --steps 2 --warmup 1 $QUIET --only-extras $LR_ROWS > $O/pmc_lr_$i.log 2>&1 < /dev/null
    echo "pass $i rc $?"
  done
  timeout 60 python - <<'PY' | tee $O/pmc_lr_consensus$TAG.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_lr/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(s in k for s in ("lrwfa_kernel", "lrmsa_kernel", "lr_kernel", "lrins_kernel", "wfa_pairs_kernel", "lr_dense_team")):
            acc[(k.split("(")[0][:48], r.get("Grid_Size", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), d in sorted(acc.items()):
    print(k, "grid", g, "launches", max(len(v) for v in d.values()))
    for c, v in sorted(d.items()):
        print("    %-30s %.5g" % (c, sum(v) / len(v)))
PY
  ;;
bench)
  ( time timeout ${BENCH_TIMEOUT:-400} python bench.py $BENCH_ARGS > $O/bench$TAG.json 2> $O/bench$TAG.err ) 2> $O/bench$TAG.time < /dev/null
  tail -3 $O/bench$TAG.time
  timeout 60 python - "$O/bench$TAG.json" <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "valu_frac", "kernel_ms", "traffic", "traffic_stale")})
    for k, v in bench.driver_view_of_config(d["config"]).items():
        print("  ", k, v if not isinstance(v, str) else v[:50])
    print(d["cpu_baseline"])
except Exception as e:
    print("bench line unreadable:", repr(e))
PY
  ;;
stats)
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_u -o u -- python $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-alone --no-cpu-baseline --no-extras --no-host-inclusive > $R/$O/stats_u.log 2>&1 < /dev/null
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_x -o x -- python $R/bench.py --steps 3 --warmup 1 $QUIET --only-extras $EXTRAS > $R/$O/stats_x.log 2>&1 < /dev/null
  cd $R
  cp $(find $O/stats_u -name "*kernel_stats.csv" | head -1) $O/split_u_c2_kernel_stats$TAG.csv 2>/dev/null
  cp $(find $O/stats_x -name "*kernel_stats.csv" | head -1) $O/extras_kernel_stats$TAG.csv 2>/dev/null
  rm -rf $O/stats_u $O/stats_x
  head -6 $O/split_u_c2_kernel_stats$TAG.csv | cut -c1-160
  head -14 $O/extras_kernel_stats$TAG.csv | cut -c1-160
  ;;
traffic)
  rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_$C -o p -- python bench.py --steps 3 --warmup 1 --repeats 1 --no-alone --no-cpu-baseline --no-extras --no-host-inclusive > gpurun_out/pmc_$C.log 2>&1 < /dev/null
  done
  timeout 60 python tools/pmc_traffic_summary.py $O | tee $O/pmc_traffic_raw.txt
  rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
  ;;
sq)
  BENCH_ARGS="--no-extras --no-host-inclusive --no-alone --repeats 1" bash tools/pmc_sq.sh > /dev/null 2>&1; cp gpurun_out/pmc_sq_summary.txt $O/pmc_sq_summary$TAG.txt
  if [ -z "$SQ_HEADLINE_ONLY" ]; then
    BENCH_ARGS="--no-host-inclusive --repeats 1 --no-alone --only-extras $EXTRAS" bash tools/pmc_sq.sh > /dev/null 2>&1; cp gpurun_out/pmc_sq_summary.txt $O/pmc_sq_summary_extras$TAG.txt
  fi
  ;;
wait)
  BENCH_ARGS="--no-extras --no-host-inclusive --no-alone --repeats 1" KERNEL=split_sparse bash tools/pmc_wait.sh > /dev/null 2>&1; cp gpurun_out/pmc_wait_summary.txt $O/pmc_wait_split_sparse$TAG.txt
  if [ -z "$WAIT_HEADLINE_ONLY" ]; then
    BENCH_ARGS="--no-host-inclusive --repeats 1 --no-alone --only-extras u_full_n20_10k_junctions" KERNEL=msa_kernel bash tools/pmc_wait.sh > /dev/null 2>&1; cp gpurun_out/pmc_wait_summary.txt $O/pmc_wait_msa_kernel$TAG.txt
  fi
  ;;
pytest)
  ( time timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -x -q -m gpu $PYTEST_ARGS 2>&1 | tail -${PYTEST_TAIL:-12} ) > $O/pytest$TAG.txt 2>&1 < /dev/null
  cat $O/pytest$TAG.txt
  ;;
py)
  ( time timeout ${PY_TIMEOUT:-300} python $PY_SCRIPT $PY_ARGS ) > $O/py$TAG.txt 2>&1 < /dev/null
  tail -${PY_TAIL:-60} $O/py$TAG.txt
  ;;
esac
done
ls -la $O | tail -30

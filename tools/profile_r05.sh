# Round-5 profile collection (run on the GPU box from the repo root; everything lands in gpurun_out/r05, copy what is to be
# judged into profiles/r05).  PMC passes are separate runs with --kernel-trace only.   PARTS="stats traffic sq wait" (default: all)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
PARTS=${PARTS:-stats traffic sq wait}
EXTRAS="u_full_n20,u_full_n20_10k_junctions,u_full_n5,sr_stage_mixed_all_svt,ins_svt4,lr_c4_align_consensus,lr_c4_msaedlib_n15,lr_ins_msawfa_n15,lr_stress_10kb_x_20kb,sr_genotype_classifier,lr_genotype_edit_distance_nw"
for P in $PARTS; do
case $P in
stats)
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_u -o u -- python $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-alone --no-cpu-baseline --no-extras --no-host-inclusive > $R/$O/stats_u.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_x -o x -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --only-extras $EXTRAS > $R/$O/stats_x.log 2>&1
  cd $R
  cp $(find $O/stats_u -name "*kernel_stats.csv" | head -1) $O/split_u_c2_kernel_stats.csv 2>/dev/null
  cp $(find $O/stats_x -name "*kernel_stats.csv" | head -1) $O/extras_kernel_stats.csv 2>/dev/null
  ;;
traffic)
  rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_$C -o p -- python bench.py --steps 3 --warmup 1 --repeats 1 --no-alone --no-cpu-baseline --no-extras --no-host-inclusive > gpurun_out/pmc_$C.log 2>&1 < /dev/null
  done
  python - <<'PY' | tee $O/pmc_traffic_raw.txt
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(c, k, "launches", len(v), "mean_KB", sum(v) / len(v))
        out.setdefault(k, {})[c] = sum(v) / len(v)
sp = [k for k in out if "split_sparse" in k]
if sp:
    d = out[sp[0]]
    others = sum(v.get("WRITE_SIZE", 0) for k, v in out.items() if k.startswith("void dh::split_") or k.startswith("dh::split_") and "sparse" not in k)
    j = {"kernel": "split_sparse_kernel", "FETCH_SIZE_KB_raw": d.get("FETCH_SIZE"), "WRITE_SIZE_KB_raw": d.get("WRITE_SIZE"),
         "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md HBM section: FETCH_SIZE reads half of a wide coalesced stream on gfx950; WRITE_SIZE uncalibrated); separate --pmc passes, tools/profile_r05.sh",
         "hbm_bytes_per_launch": (2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024,
         "launch": "10000 junctions, BASELINE config 2, one junction per wavefront",
         "all_kernels_KB": out}
    json.dump(j, open("gpurun_out/r05/pmc_traffic.json", "w"), indent=1)
    print(json.dumps({k: j[k] for k in ("FETCH_SIZE_KB_raw", "WRITE_SIZE_KB_raw", "hbm_bytes_per_launch")}))
PY
  ;;
sq)
  BENCH_ARGS="--no-extras --no-host-inclusive --no-alone --repeats 1" bash tools/pmc_sq.sh > /dev/null 2>&1; cp gpurun_out/pmc_sq_summary.txt $O/pmc_sq_summary.txt
  BENCH_ARGS="--no-host-inclusive --only-extras $EXTRAS" bash tools/pmc_sq.sh > /dev/null 2>&1; cp gpurun_out/pmc_sq_summary.txt $O/pmc_sq_summary_extras.txt
  ;;
wait)
  BENCH_ARGS="--no-extras --no-host-inclusive --no-alone --repeats 1" KERNEL=split_sparse bash tools/pmc_wait.sh > /dev/null 2>&1; cp gpurun_out/pmc_wait_summary.txt $O/pmc_wait_split_sparse.txt
  BENCH_ARGS="--no-host-inclusive --only-extras u_full_n20_10k_junctions" KERNEL=msa_kernel bash tools/pmc_wait.sh > /dev/null 2>&1; cp gpurun_out/pmc_wait_summary.txt $O/pmc_wait_msa_kernel.txt
  ;;
micro)
  # LDS access patterns (tools/lds_rate.hip) and the VALU clock (tools/valu_clock.hip): built here, on the GPU box
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/lds_rate.hip -o /tmp/lds_rate.bin 2>/dev/null && /tmp/lds_rate.bin > $O/lds_rate.txt 2>&1
  ;;
stream)
  # the pipelined host-buffer path by depth (CHANGELOG.md 1b)
  python tools/bench_stream.py > $O/bench_stream_u_c2.json 2> $O/bench_stream.err
  bash tools/stream_matrix.sh > $O/stream_matrix.txt 2>&1
  ;;
resources_on_cpu_only)
  # registers / LDS / scratch of every kernel of the shipped library, from the code object's notes
  python tools/resource_usage.py > $O/resource_usage.txt 2>&1
  ;;
esac
done
ls -la $O

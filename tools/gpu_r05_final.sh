# round 5, last call: the default bench line, the kernel stats of the long-read rows after the branch-free letter tables, the whole GPU suite
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
( time timeout 300 python bench.py > $O/bench_final.json 2> $O/bench_final.err ) 2> $O/bench_final.time < /dev/null
tail -3 $O/bench_final.time
timeout 60 python - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
try:
    d = json.loads(open("gpurun_out/r05/bench_final.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "kernel_ms", "traffic")})
    for k, v in bench.driver_view_of_config(d["config"]).items():
        print("  ", k, v if not isinstance(v, str) else v[:50])
    print(d["cpu_baseline"])
except Exception as e:
    print("bench line unreadable:", repr(e))
PY
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_lr -o x -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --only-extras lr_c4_align_consensus,lr_c4_msaedlib_n15,lr_ins_msawfa_n15 > $R/$O/stats_lr.log 2>&1 < /dev/null
cd $R
f=$(find $O/stats_lr -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $O/lr_rows_kernel_stats.csv; timeout 10 head -8 $O/lr_rows_kernel_stats.csv | cut -c1-150; fi
rm -rf $O/stats_lr
( time timeout 540 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > $O/pytest_final.txt 2>&1 < /dev/null
cat $O/pytest_final.txt

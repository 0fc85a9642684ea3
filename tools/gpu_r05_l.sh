export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
O=$R/gpurun_out/r05
rocprofv3 --kernel-trace --output-format csv -d $O/tr_mixed -o m -- python $R/bench.py --steps 5 --warmup 1 --repeats 1 --no-alone --no-cpu-baseline --no-host-inclusive --only-extras sr_stage_mixed_all_svt > /dev/null 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r05/tr_mixed/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "msa_kernel" in r["Kernel_Name"] and "slow" not in r["Kernel_Name"]]
# the resident leg's launches come before the stream leg: take the 3rd and 4th msa launches
a,b=idx[2],idx[3]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b+1]:
    print("  %8.1f us +%7.1f  queue %s  %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Queue_Id","?"), r["Kernel_Name"][:60]))
PY
rm -rf $O/tr_mixed

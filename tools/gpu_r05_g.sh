export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
O=$R/gpurun_out/r05
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_a -o a -- python $R/bench.py --steps 5 --warmup 1 --repeats 1 --no-alone --no-cpu-baseline --no-host-inclusive --only-extras u_full_n20 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_b -o b -- python $R/bench.py --steps 5 --warmup 1 --repeats 1 --no-alone --no-cpu-baseline --no-host-inclusive --only-extras u_c2_40k_junctions,u_full_n20 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv,glob
for tag in "ab":
    f=glob.glob("gpurun_out/r05/tr_%s/**/*kernel_trace.csv"%tag, recursive=True)[0]
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    # find msa_kernel launches; print the sequence between the last two msa_kernel launches
    idx=[i for i,r in enumerate(rows) if "msa_kernel" in r["Kernel_Name"] and "slow" not in r["Kernel_Name"]]
    print(tag, "msa launches", len(idx))
    a,b=idx[-2],idx[-1]
    t0=int(rows[a]["Start_Timestamp"])
    for r in rows[a:b+1]:
        print("  %8.1f us +%7.1f  grid %s  %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Grid_Size_X", r.get("Grid_Size","?")), r["Kernel_Name"][:70]))
PY
rm -rf gpurun_out/r05/tr_a gpurun_out/r05/tr_b

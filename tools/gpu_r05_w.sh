# round 5, call W: kernel traces of the asynchronous return -- on an idle device (tools/fetch_async_rate.py) and inside bench.py's N > 1 step
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fa -o fa -- python $R/tools/fetch_async_rate.py > $O/fetch_async_rate.txt 2>&1 < /dev/null
f=$(find /tmp/fa -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $O/fetch_async_idle_kernel_stats.csv; fi
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fc -o fc -- python $R/bench.py --force-comm --gather shm --no-extras --no-cpu-baseline --no-host-inclusive > $O/force_comm_profiled.log 2>&1 < /dev/null
f=$(find /tmp/fc -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $O/force_comm_shm_kernel_stats.csv; fi
cd $R
grep -h "ms per return" $O/fetch_async_rate.txt
for g in fetch_async_idle_kernel_stats.csv force_comm_shm_kernel_stats.csv; do echo "== $g"; if [ -f $O/$g ]; then timeout 10 cut -c1-140 $O/$g < /dev/null | sed -n 1,9p; fi; done

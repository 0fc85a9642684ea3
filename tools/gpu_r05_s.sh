export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05
mkdir -p $O
python tools/fetch_async_rate.py
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/fa -o fa -- python $R/tools/fetch_async_rate.py > /dev/null 2>&1
f=$(find /tmp/fa -name "*kernel_stats.csv" | head -1)
cp $f $O/fetch_async_kernel_stats.csv
head -12 $f | cut -c1-160

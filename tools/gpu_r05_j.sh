export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
O=$R/gpurun_out/r05
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_mixed -o m -- python $R/bench.py --steps 5 --warmup 1 --repeats 1 --no-alone --no-cpu-baseline --no-host-inclusive --only-extras sr_stage_mixed_all_svt > /dev/null 2>&1
cd $R
cp $(find $O/tr_mixed -name "*kernel_stats.csv" | head -1) $O/mixed_all_svt_kernel_stats.csv
head -25 $O/mixed_all_svt_kernel_stats.csv
rm -rf $O/tr_mixed

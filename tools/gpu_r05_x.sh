# round 5, call X: blob_offsets_kernel with a round's loads in flight at once -- every suite that fetches in a different way, and its time
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_fetch_async.py tests/test_gpu_robustness.py tests/test_gpu_stream.py tests/test_gpu_multirank.py "tests/test_gpu_bench_shapes.py::test_side_measurement_batches_vs_reference[u_c2_40k_junctions]" tests/test_gpu_bench_shapes.py::test_host_inclusive_stream_results_vs_reference tests/test_gpu_ins.py -x -q 2>&1 < /dev/null | tail -5
cd /tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fa -o fa -- python $R/tools/fetch_async_rate.py > $O/fetch_async_rate2.txt 2>&1 < /dev/null
f=$(find /tmp/fa -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $O/fetch_async_idle_kernel_stats2.csv; timeout 10 grep -h "blob_offsets\|fetch_out\|blob_gather" $O/fetch_async_idle_kernel_stats2.csv < /dev/null | cut -c1-40,100-200; fi
grep -h "ms per return" $O/fetch_async_rate2.txt < /dev/null
cd $R
timeout 100 python bench.py --no-extras --no-cpu-baseline 2>/dev/null < /dev/null | timeout 20 python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'host_inclusive', d['config'].get('host_inclusive_alignments_per_s'))"

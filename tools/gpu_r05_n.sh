export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/dense_latency.py 2>&1 | head -3
echo "---- DELLYHIP_SR_WIDE=0"
DELLYHIP_SR_WIDE=0 python tools/dense_latency.py 2>&1 | head -3
timeout 1500 python -m pytest tests/test_gpu_split.py tests/test_gpu_sparse.py tests/test_gpu_fuzz.py tests/test_gpu_lowcx.py tests/test_gpu_example_reads.py tests/test_gpu_stream.py tests/test_gpu_dropin_cpp.py -x -q -m gpu 2>&1 | tail -6

# round 5, call B: parity of the paired Gotoh pass + rates
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_msa.py tests/test_gpu_msa_big.py tests/test_gpu_lowcx.py tests/test_gpu_bench_shapes.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_msa.txt
cat $O/pytest_msa.txt
python tools/msa_rate.py 10000 20 > $O/msa_rate_pair.txt 2>&1
python tools/msa_rate.py 2000 20 >> $O/msa_rate_pair.txt 2>&1
python tools/msa_rate.py 2000 5 >> $O/msa_rate_pair.txt 2>&1
cat $O/msa_rate_pair.txt

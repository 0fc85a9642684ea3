# round 5, call D: parity after the pair-pass rewrite (d16 loads, rotation, shorter passes) + phases
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_msa.py tests/test_gpu_msa_big.py tests/test_gpu_lowcx.py -x -q -m gpu -k "msa or Msa or full" 2>&1 | tail -8 > $O/pytest_msa2.txt
cat $O/pytest_msa2.txt
DELLYHIP_LIB=$R/tools/bin/lib_msa_timing.bin DELLYHIP_MSA_ONLY=1 python tools/msa_phases.py 10000 20 > $O/msa_phases2.txt 2>&1
python tools/msa_rate.py 10000 20 >> $O/msa_phases2.txt 2>&1
python tools/msa_rate.py 2000 20 >> $O/msa_phases2.txt 2>&1
python tools/msa_rate.py 2000 5 >> $O/msa_phases2.txt 2>&1
cat $O/msa_phases2.txt

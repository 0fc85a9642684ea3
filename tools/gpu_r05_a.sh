# round 5, call A: scalar-store microbenchmark, MSA baseline rates, serial kernel trace of one U_full launch (msa_slow_kernel's real duration)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
timeout 120 tools/sstore_rate.bin > $O/sstore_rate.txt 2>&1
tail -20 $O/sstore_rate.txt
python tools/msa_rate.py 10000 20 > $O/msa_rate_start.txt 2>&1
python tools/msa_rate.py 2000 20 >> $O/msa_rate_start.txt 2>&1
cat $O/msa_rate_start.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/serial_ufull -o s -- python $R/tools/msa_rate.py 10000 20 > $R/$O/serial_ufull.log 2>&1
cd $R
cp $(find $O/serial_ufull -name "*kernel_stats.csv" | head -1) $O/serial_ufull_kernel_stats.csv 2>/dev/null
head -12 $O/serial_ufull_kernel_stats.csv

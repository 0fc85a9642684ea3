# VALU issue rate with the clock measured, not assumed: tools/valu_clock.bin beside a rocm-smi poll of sclk
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
( for i in $(seq 1 40); do /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|fclk|mclk" | head -3 | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/valu_clock_smi.txt 2>&1 &
SMI=$!
tools/valu_clock.bin > gpurun_out/valu_clock.txt 2>&1
kill $SMI 2>/dev/null
cat gpurun_out/valu_clock.txt
echo "--- rocm-smi during the run (every 0.5 s) ---"
sort gpurun_out/valu_clock_smi.txt | uniq -c | sort -rn | head -8

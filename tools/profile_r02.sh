export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/r02
bash tools/pmc_traffic.sh > gpurun_out/r02/pmc_traffic.txt 2>&1
BENCH_ARGS="--no-extras" bash tools/pmc_sq.sh > /dev/null 2>&1; cp gpurun_out/pmc_sq_summary.txt gpurun_out/r02/pmc_sq_summary.txt
BENCH_ARGS="--only-extras u_full_n20,lr_c4_align_consensus,lr_c4_msaedlib_n15,ins_svt4,sr_genotype_classifier,lr_genotype_edit_distance_nw" bash tools/pmc_sq.sh > /dev/null 2>&1; cp gpurun_out/pmc_sq_summary.txt gpurun_out/r02/pmc_sq_summary_extras.txt
bash tools/pmc_wait.sh > /dev/null 2>&1; cp gpurun_out/pmc_wait_summary.txt gpurun_out/r02/pmc_wait_split_sparse.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r02/stats_u -o u -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > /root/repo/gpurun_out/r02/stats_u.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r02/stats_x -o x -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --only-extras u_full_n20,u_full_n5,ins_svt4,lr_c4_align_consensus,lr_c4_msaedlib_n15,lr_ins_msawfa_n15,sr_genotype_classifier,lr_genotype_edit_distance_nw > /root/repo/gpurun_out/r02/stats_x.log 2>&1
cd /root/repo
find gpurun_out/r02 -name "*kernel_stats.csv" | head; cat gpurun_out/r02/pmc_traffic.txt | tail -8

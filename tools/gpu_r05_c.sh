# round 5, call C: phase breakdown of msa_kernel with the paired pass; team sizes at small batches
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
rm -f $O/msa_phases.txt
for PAIR in 1; do
  echo "== DELLYHIP_MSA_PAIR=$PAIR" >> $O/msa_phases.txt
  DELLYHIP_MSA_PAIR=$PAIR DELLYHIP_LIB=$R/tools/bin/lib_msa_timing.bin DELLYHIP_MSA_ONLY=1 python tools/msa_phases.py 10000 20 >> $O/msa_phases.txt 2>&1
done
for N in 500 1000 2000 3000; do for T in 1 2 4; do
DELLYHIP_MSA_TEAM=$T python tools/msa_rate.py $N 20 >> $O/msa_phases.txt 2>&1
done; done
cat $O/msa_phases.txt

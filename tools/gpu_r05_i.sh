export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
python bench.py --steps 5 --warmup 1 --repeats 1 --no-alone --no-cpu-baseline --no-host-inclusive --only-extras u_c2_40k_junctions,u_full_n20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
x=d['extras']['u_full_n20']; print('after 40k:', x['junctions_per_s'], x['ms_per_step'], x['msa_stage_ms'], x['split_stage_ms'])"
timeout 900 python -m pytest tests/test_gpu_msa.py tests/test_gpu_lrmsa.py tests/test_gpu_stream.py tests/test_gpu_robustness.py -x -q -m gpu 2>&1 | tail -4
DELLYHIP_LIB=$R/tools/bin/lib_msa_timing.bin DELLYHIP_MSA_ONLY=1 python tools/msa_phases.py 10000 20 | tail -14

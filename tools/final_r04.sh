# round-4 closing run on the GPU box: the whole -m gpu suite, the driver's bench command, kernel statistics of the side rows,
# the time line of the long-read teams.  bash tools/final_r04.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final/pt_full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/final/pt_full.log | tail -3
timeout 600 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; echo "bench rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats_x -o x -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --no-alone --only-extras lr_c4_align_consensus,lr_c4_msaedlib_n15,lr_ins_msawfa_n15,u_full_n20,u_full_n20_10k_junctions,ins_svt4 > gpurun_out/final/stats_x.log 2>&1
cp $(find gpurun_out/final/stats_x -name "*kernel_stats.csv" | head -1) gpurun_out/final/extras_kernel_stats.csv 2>/dev/null
DELLYHIP_LIB=tools/bin/lib_teamdbg.bin timeout 60 python tools/lr_team_check.py 2048 > gpurun_out/final/lr_team_timeline.txt 2>&1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")})
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["config"].items() if isinstance(v, (int, float))})
print("roofline", {k: v for k, v in d["roofline"].items() if not isinstance(v, dict)})
print("cpu", d.get("cpu_baseline"))
PY

# round 5, call E: the new bench rows + their parity tests
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
( time python bench.py --steps 20 --warmup 3 > $O/bench_e.json 2> $O/bench_e.err ) 2> $O/bench_e.time
tail -3 $O/bench_e.time
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05/bench_e.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"])
import sys; sys.path.insert(0, ".")
import bench
for k,v in bench.driver_view_of_config(d["config"]).items(): print("  ", k, v if not isinstance(v,str) else v[:50])
for k,v in d["extras"].items():
    if isinstance(v,dict) and "junctions_per_s" in v: print(k, {x:v[x] for x in v if x in ("junctions","junctions_per_s","ms_per_step","refined_ok","msa_deferred_junctions","msa_stage_ms","split_stage_ms")}, v.get("cpu_reference"))
PY
timeout 1500 python -m pytest tests/test_gpu_bench_shapes.py tests/test_abi.py -x -q -m gpu -k "side_measurement or chip_filling or abi" 2>&1 | tail -8 > $O/pytest_e.txt
cat $O/pytest_e.txt

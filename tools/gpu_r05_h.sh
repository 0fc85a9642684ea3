# round 5, call H: the whole GPU suite + the default bench line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05
mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > $O/pytest_all.txt 2>&1
cat $O/pytest_all.txt
( time python bench.py > $O/bench_h.json 2> $O/bench_h.err ) 2> $O/bench_h.time
tail -3 $O/bench_h.time
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
d=json.loads(open("gpurun_out/r05/bench_h.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", {k:d["roofline"][k] for k in ("achieved","frac","kernel_ms","traffic")})
for k,v in bench.driver_view_of_config(d["config"]).items(): print("  ", k, v if not isinstance(v,str) else v[:50])
print(d["cpu_baseline"])
PY

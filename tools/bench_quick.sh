# full default bench line into gpurun_out/<name>.json + a short summary on stdout:  bash tools/bench_quick.sh name [bench args]
cd ${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-bench}; shift
mkdir -p gpurun_out
python bench.py "$@" > gpurun_out/$N.json 2> gpurun_out/$N.err; echo "bench rc=$?"
python - "$N" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")})
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["config"].items() if isinstance(v, (int, float))})
print("kernel_ms", d["roofline"]["kernel_ms"], "alone", d["roofline"].get("kernel_ms_alone"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY

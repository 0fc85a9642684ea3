"""FETCH_SIZE / WRITE_SIZE (and, when tools/pmc_sq.sh ran before, SQ_INSTS_VALU) of the headline's kernels -> <out>/pmc_traffic.json,
stamped with the hash of the kernel's sources (delly_amd/build.py: headline_kernel_hash) so that bench.py can tell whether the
committed counters belong to the kernel it runs (`roofline.traffic_stale`).  Run on the GPU box by tools/gpu_run.sh (part `traffic`)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from delly_amd import build  # noqa: E402

out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(c, k, "launches", len(v), "mean_KB", sum(v) / len(v))
        out.setdefault(k, {})[c] = sum(v) / len(v)
valu = None
for f in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r.get("Counter_Name") == "SQ_INSTS_VALU" and "split_sparse_kernel" in r["Kernel_Name"]]
    if v:
        valu = sum(v) / len(v)
sp = [k for k in out if "split_sparse_kernel" in k]
if sp:
    d = out[sp[0]]
    j = {"kernel": "split_sparse_kernel", "FETCH_SIZE_KB_raw": d.get("FETCH_SIZE"), "WRITE_SIZE_KB_raw": d.get("WRITE_SIZE"),
         "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md HBM section: FETCH_SIZE reads half of a wide coalesced "
                       "stream on gfx950; WRITE_SIZE uncalibrated); separate --pmc passes, tools/gpu_run.sh part `traffic`",
         "hbm_bytes_per_launch": (2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024,
         "valu_wave_instructions_per_launch": valu,
         "launch": "10000 junctions, BASELINE config 2, one junction per wavefront",
         "kernel_source_sha16": build.headline_kernel_hash(),
         "kernel_sources": list(build.HEADLINE_KERNEL_SOURCES),
         "all_kernels_KB": out}
    json.dump(j, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({k: j[k] for k in ("FETCH_SIZE_KB_raw", "WRITE_SIZE_KB_raw", "hbm_bytes_per_launch", "valu_wave_instructions_per_launch", "kernel_source_sha16")}))

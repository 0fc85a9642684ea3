"""deficit sweep rows (one launch at a time) for a library variant: python tools/sweep_pred.py   (DELLYHIP_LIB selects the build)"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from delly_amd import refine, synth
import bench
ctx = refine.Context()
out = []
for name, kw in bench.SWEEP_PLAN:
    if name not in ("substitutions_0.5pct_baseline", "substitutions_2pct", "substitutions_5pct", "nontemplated_insertion_12bp", "indel_1to3bp_per_consensus"):
        continue
    b = bench.sweep_batch(synth, kw)
    ctx.set_chromosomes(b.chroms)
    rb = ctx.upload(b)
    rb.run(); rb.sync(); rb.kernel_ms()
    t0 = time.perf_counter()
    for _ in range(5): rb.run()
    rb.sync()
    dt = (time.perf_counter() - t0) / 5
    ms_split, _, _ = rb.kernel_ms()
    out.append("%s %.1f M/s (sparse %.3f ms, all %.3f ms, left %d)" % (name.replace("substitutions_", "s").replace("_baseline", ""), b.n / dt / 1e6, rb.dp_kernel_ms(), ms_split, rb.sparse_left()))
    rb.free()
print(" | ".join(out))

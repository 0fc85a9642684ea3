"""PCIe-inclusive rate of the genotyping entry points that take host buffers (never bench.py's `value`):
dellyhip_classify_reads and dellyhip_edit_distance_nw_batch -- upload + kernels + download per call."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from delly_amd import refine, synth  # noqa: E402

ctx = refine.Context()
base_jobs, base_blob = synth.make_align_jobs(160, 40, seed=9)
tiles = 82
jobs = np.tile(base_jobs, tiles)
shift = np.repeat(np.arange(tiles, dtype=np.uint64) * np.uint64(base_blob.size), base_jobs.shape[0])
for f in ("cons_off", "ref_off", "seq_off"):
    jobs[f] += shift
blob = np.tile(base_blob, tiles)
ctx.classify_reads(jobs[:1000], blob)
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    res = ctx.classify_reads(jobs, blob)
dt = (time.perf_counter() - t0) / reps
print("classify_reads host buffers: %d jobs, %.1f MB in / %.1f MB out, %.2f ms per call = %.1f M jobs/s" %
      (jobs.shape[0], (jobs.nbytes + blob.nbytes) / 1e6, res.nbytes / 1e6, dt * 1e3, jobs.shape[0] / dt / 1e6))
nj, nblob = synth.make_nw_jobs(2048, seed=19)
ctx.edit_distance_nw_batch(nj[:16], nblob)
t0 = time.perf_counter()
for _ in range(reps):
    d = ctx.edit_distance_nw_batch(nj, nblob)
dt = (time.perf_counter() - t0) / reps
print("edit_distance_nw_batch host buffers: %d pairs, %.1f MB in, %.2f ms per call = %.2f M pairs/s" %
      (nj.shape[0], (nj.nbytes + nblob.nbytes) / 1e6, dt * 1e3, nj.shape[0] / dt / 1e6))

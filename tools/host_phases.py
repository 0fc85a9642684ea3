"""where the host-buffer entry point spends its time: upload / run + sync / fetch / free of a 10 000-junction C2 batch"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from delly_amd import refine, synth
b = synth.make_batch(10000, mode="c2")
ctx = refine.Context()
ctx.set_chromosomes(b.chroms)
ctx.refine(b)
T = np.zeros(5)
R = 10
for _ in range(R):
    t0 = time.perf_counter(); rb = ctx.upload(b)
    t1 = time.perf_counter(); rb.run(); rb.sync()
    t2 = time.perf_counter(); r, blob = rb.fetch()
    t3 = time.perf_counter(); rb.free()
    t4 = time.perf_counter()
    res, bl = ctx.refine(b)
    t5 = time.perf_counter()
    T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]
print("ms: upload %.2f  run+sync %.2f  fetch %.2f  free %.2f | one-call refine %.2f" % tuple(T / R * 1e3))

"""U_full micro-benchmark: msa() + alignConsensus() per junction (what `delly sr` pays), resident batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from delly_amd import refine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
b = synth.make_batch(n, mode="c2", n_reads=nreads)
ctx = refine.Context()
ctx.set_chromosomes(b.chroms)
rb = ctx.upload(b)
rb.run(); rb.sync(); rb.kernel_ms()
t = time.perf_counter()
for _ in range(steps):
    rb.run()
rb.sync()
dt = (time.perf_counter() - t) / steps
ms_split, ms_msa, _ = rb.kernel_ms()
res, _ = rb.fetch()
print("U_full n=%d reads=%d: %.2f ms/step -> %.0f junctions/s | msa kernel %.2f ms, split %.2f ms | ok %d mean cons %.0f" % (
    n, nreads, dt * 1e3, n / dt, ms_msa, ms_split, int(res["ok"].sum()), res["cons_len"].mean()), flush=True)

"""Side benchmarks on a resident batch: mode c2/mixed/ins/lr, n_reads 0 = given consensus (unit U),
>0 = msa stage + alignConsensus (U_full; lr: msaEdlib).  GPU only; the CPU reference numbers quoted in
DESIGN.md come from bench.py (cpu_baseline and the reference legs of its extras)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from delly_amd import refine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mode = sys.argv[4] if len(sys.argv) > 4 else "c2"
from delly_amd import abi
kw = dict(sub_rate=0.01) if mode == "lr" else {}
params = abi.params_lr(realign=True) if mode == "lr" else None
b = synth.make_batch(n, mode=mode, n_reads=nreads, **kw)
ctx = refine.Context(params=params)
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
print(mode + " n=%d reads=%d: %.2f ms/step -> %.0f junctions/s | msa kernel %.2f ms, split %.2f ms | ok %d mean cons %.0f" % (
    n, nreads, dt * 1e3, n / dt, ms_msa, ms_split, int(res["ok"].sum()), res["cons_len"].mean()), flush=True)

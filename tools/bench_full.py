"""U_full micro-benchmark: msa() + alignConsensus() per junction (what `delly sr` pays), resident batch."""
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

if len(sys.argv) > 5:  # CPU reference beside it (oracle/_ref, all host threads) -- test infrastructure, not the product
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyoracle
    kind = "reference" if pyoracle.have_reference() else "port"
    O = pyoracle.Oracle(kind)
    if params is not None:
        O.params = params
    for th in (1, os.cpu_count()):
        k = min(n, 2048 if mode != "lr" else 256) if th > 1 else min(n, 128 if mode != "lr" else 4)
        s1 = synth.make_batch(k, mode=mode, n_reads=nreads, **kw)
        t = time.perf_counter()
        O.refine_batch(s1, want_alignment=False, n_threads=th)
        dt = time.perf_counter() - t
        print("cpu %s threads=%d: %d junctions in %.2f s -> %.0f junctions/s" % (kind, th, k, dt, k / dt), flush=True)

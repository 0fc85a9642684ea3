"""PCIe-inclusive rate of the host-buffer entry point (dellyhip_align_consensus_batch):
H2D of junction records + consensus bytes, kernels, device-side compaction, D2H of records + used
blob bytes.  The chromosome is resident (uploaded once).  Never reported as bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delly_amd import refine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
b = synth.make_batch(n, mode="c2")
ctx = refine.Context()
ctx.set_chromosomes(b.chroms)
ctx.refine(b)
for want in (False, True):
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        res, blob = ctx.refine(b, want_alignment=want)
    dt = (time.perf_counter() - t) / reps
    print("host-buffer path n=%d want_alignment=%d: %.2f ms -> %.0f junctions/s (blob %d bytes, ok %d)" % (
        n, want, dt * 1e3, n / dt, blob.size, int(res["ok"].sum())), flush=True)

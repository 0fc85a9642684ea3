"""What do the dense fallback kernels cost (a) as the latency of a few leftover junctions behind the sparse kernel, (b) at full
load (DELLYHIP_SR_SPARSE=0)?  A/B of launch bounds: DELLYHIP_LIB=<build with -DDH_PAIR_WAVES=2 -DDH_QUAD_WAVES=2>."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delly_amd import refine, synth
import bench
def rate(ctx, b, label, steps=5):
    ctx.set_chromosomes(b.chroms)
    rb = ctx.upload(b)
    rb.run(); rb.sync(); rb.kernel_ms()
    t0 = time.perf_counter()
    for _ in range(steps):
        rb.run()
    rb.sync()
    dt = (time.perf_counter() - t0) / steps
    ms_split, ms_msa, _ = rb.kernel_ms()
    left = rb.sparse_left()
    print("%-44s %.3f ms per step = %.2f M/s; msa %.3f split %.3f, sparse kernel %.3f, left to dense %d" % (label, dt * 1e3, b.n / dt / 1e6, ms_msa, ms_split, rb.dp_kernel_ms(), left), flush=True)
    rb.free()
ctx = refine.Context()
rate(ctx, bench.sweep_batch(synth, dict(sub_rate=0.05)), "C2 with 5 % substitutions (10 000)")
rate(ctx, synth.make_batch(10000, mode="allsvt", n_reads=(2, 20)), "all SV types, 2..20 reads (10 000)")
rate(ctx, synth.make_batch(5000, mode="mixed"), "mixed svt, consensus given (5 000)")
os.environ["DELLYHIP_SR_SPARSE"] = "0"
c2 = refine.Context()
rate(c2, synth.make_batch(10000, mode="c2"), "C2, dense kernels only (10 000)")

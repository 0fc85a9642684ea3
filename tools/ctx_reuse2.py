"""bench.py's u_full_n20 row is slower after the 40 000-junction row (2.1 vs 1.6 ms per step, same kernels): which ingredient?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--torch" in sys.argv:
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
from delly_amd import refine, synth
import bench
small = synth.make_batch(2000, mode="c2", n_reads=20)
big = synth.make_batch(40000, mode="c2")
def rate(ctx, b, label, steps=3):
    ctx.set_chromosomes(b.chroms)
    rb = ctx.upload(b)
    rb.run(); rb.sync(); rb.kernel_ms()
    t0 = time.perf_counter()
    for _ in range(steps):
        rb.run()
    rb.sync()
    dt = (time.perf_counter() - t0) / steps
    ms_split, ms_msa, _ = rb.kernel_ms()
    res, _ = rb.fetch()
    print("%-50s %.3f ms per step; kernels: msa %.3f split %.3f" % (label, dt * 1e3, ms_msa, ms_split), flush=True)
    rb.free()
ctx = refine.Context()
rate(ctx, small, "fresh: 2k x 20 reads")
rate(ctx, big, "40k C2 resident")
rate(refine.Context(), small, "2k x 20 reads, new context, after 40k resident")
ctx.set_chromosomes(big.chroms)
hi = bench.host_inclusive_rate(ctx, [big], 0, seconds=0.5, depth=5)
print("stream of 40k batches: %.1f M/s" % (hi["value"] / 1e6), flush=True)
rate(refine.Context(), small, "2k x 20 reads, new context, after the stream")
hi = bench.host_inclusive_rate(refine.Context(), [small], 1, seconds=0.5, depth=5) if False else None
time.sleep(1.0)
rate(refine.Context(), small, "... after 1 s of sleep")
ctx.trim_memory()
rate(refine.Context(), small, "... after dellyhip_trim_memory")

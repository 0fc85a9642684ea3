"""headline kernel throughput against the batch size (resident batch, run + sync per step): how the tail of
split_sparse_kernel amortises"""
import sys, time
sys.path.insert(0, '/root/repo')
from delly_amd import refine, synth
ctx = refine.Context()
for n in (1250, 2500, 5000, 10000, 20000, 40000, 80000):
    b = synth.make_batch(n, mode="c2")
    ctx.set_chromosomes(b.chroms)
    rb = ctx.upload(b)
    for _ in range(3):
        rb.run(); rb.sync()
    t0 = time.perf_counter()
    R = 20
    for _ in range(R):
        rb.run(); rb.sync()
    dt = (time.perf_counter() - t0) / R
    rb.kernel_ms()
    print("n %6d: %.3f ms per step, %.1f M alignments/s" % (n, dt * 1e3, n / dt / 1e6), flush=True)
    rb.free()

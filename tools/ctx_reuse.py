"""Does a context that has seen a large batch run a small msa() batch slower?  (bench.py's u_full_n20 row after u_c2_40k)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delly_amd import refine, synth
small = synth.make_batch(2000, mode="c2", n_reads=20)
big = synth.make_batch(40000, mode="c2")
def rate(ctx, b, label):
    ctx.set_chromosomes(b.chroms)
    rb = ctx.upload(b)
    rb.run(); rb.sync(); rb.kernel_ms()
    tr = ts = 0.0
    for _ in range(5):
        t0 = time.perf_counter(); rb.run(); t1 = time.perf_counter(); rb.sync(); t2 = time.perf_counter()
        tr += t1 - t0; ts += t2 - t1
    ms_split, ms_msa, _ = rb.kernel_ms()
    print("%-34s run() %.3f ms + sync() %.3f ms per step; kernels: msa %.3f split %.3f" % (label, tr / 5 * 1e3, ts / 5 * 1e3, ms_msa, ms_split))
    rb.free()
ctx = refine.Context()
rate(ctx, small, "fresh context, 2k x 20 reads")
rate(ctx, big, "40k C2")
rate(ctx, small, "same context again, 2k x 20 reads")
ctx2 = refine.Context()
rate(ctx2, small, "second context, 2k x 20 reads")
# ... and after a dellyhip_stream has run on the context (bench.py's host-inclusive leg of the previous row)?
import bench
ctx2.set_chromosomes(big.chroms)
hi = bench.host_inclusive_rate(ctx2, [big], 0, seconds=0.3, depth=5)
print("stream on the second context: %.1f M junctions/s" % (hi["value"] / 1e6))
rate(ctx2, small, "second context after a stream")
rate(ctx, small, "first context (no stream of its own)")
ctx3 = refine.Context()
rate(ctx3, small, "third context, created after")
# ... and after the arrangement of bench.py's headline leg (two contexts sharing a genome, batches on the two compute streams)?
c4 = refine.Context()
c4.set_chromosomes(big.chroms)
c5 = refine.Context(share_with=c4)
streams = list(c4.compute_streams())[:2]
rbs = [c4.upload(big), c5.upload(big)]
for k in range(6):
    rbs[k % 2].run(streams[k % 2])
for x in rbs:
    x.sync(); x.free()
rate(c4, small, "context after the headline arrangement")
c5.close()
rate(c4, small, "... after closing the second context")

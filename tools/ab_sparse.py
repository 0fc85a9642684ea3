"""A/B of split_sparse_kernel builds: python tools/ab_sparse.py lib1 lib2 ...  (each in its own process: the library is loaded once)
per library: kernel ms alone (one launch at a time), two in flight, 40 000 junctions in one launch"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--one":
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "default":
            env["DELLYHIP_LIB"] = lib
        out = subprocess.run([sys.executable, __file__, "--one"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print("%-32s %s" % (lib, out.stdout.strip() or out.stderr[-400:]))
    sys.exit(0)
sys.path.insert(0, ROOT)
import time
import numpy as np
from delly_amd import refine, synth
import bench
raw = [synth.make_batch(10000, mode="c2", first=k * 10000) for k in range(4)]
chroms, batches = bench.one_genome(synth, raw)
ctx = refine.Context()
ctx.set_chromosomes(chroms)
ctx2 = refine.Context(share_with=ctx)
st = ctx.compute_streams()
rbs = [(ctx, ctx2)[k % 2].upload(b) for k, b in enumerate(batches)]
for k, x in enumerate(rbs):
    x.run(st[k % 2])
for x in rbs:
    x.sync(); x.kernel_ms()
# alone
t0 = time.perf_counter()
for k in range(24):
    rbs[k % 4].run(st[0]); rbs[k % 4].sync()
alone = (time.perf_counter() - t0) / 24
kk = [x.kernel_ms() for x in rbs]; dd = [x.dp_kernel_ms() for x in rbs]
kms = sum(d * k[2] for d, k in zip(dd, kk)) / sum(k[2] for k in kk)
# two in flight
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for k in range(40):
        rbs[k % 4].run(st[k % 2])
    for x in rbs:
        x.sync()
    best = min(best, (time.perf_counter() - t0) / 40)
big = synth.make_batch(40000, mode="c2")
ctx.set_chromosomes(big.chroms)
rb = ctx.upload(big)
rb.run(); rb.sync()
t0 = time.perf_counter()
for k in range(5):
    rb.run()
rb.sync()
t40 = (time.perf_counter() - t0) / 5
print("alone %.3f ms/step (kernel %.3f ms) = %.1f M/s | two in flight %.3f ms = %.1f M/s | 40k: %.3f ms = %.1f M/s" %
      (alone * 1e3, kms, 1e4 / alone / 1e6, best * 1e3, 1e4 / best / 1e6, t40 * 1e3, 4e4 / t40 / 1e6))

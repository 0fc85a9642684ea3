"""U_full (msa of N reads + alignConsensus) at one batch size: python tools/msa_rate.py [junctions] [reads]; DELLYHIP_MSA_TEAM=1|2|4 forces the team size"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delly_amd import refine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 20
b = synth.make_batch(n, mode="c2", n_reads=nr)
ctx = refine.Context()
ctx.set_chromosomes(b.chroms)
rb = ctx.upload(b)
rb.run(); rb.sync(); rb.kernel_ms()
t0 = time.perf_counter()
for _ in range(3):
    rb.run()
rb.sync()
dt = (time.perf_counter() - t0) / 3
ms_split, ms_msa, _ = rb.kernel_ms()
res, _ = rb.fetch()
print("team %s: %d junctions x %d reads: %.3f ms per step = %.3f M junctions/s (msa stage %.3f ms, split %.3f ms), ok %d" %
      (os.environ.get("DELLYHIP_MSA_TEAM", "auto"), n, nr, dt * 1e3, n / dt / 1e6, ms_msa, ms_split, int(res["ok"].sum())))
